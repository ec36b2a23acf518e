#!/bin/bash
# round 6, batch a: GPU tests, A/B of the pre-split weight slabs in conv_x3_kernel<32,4>, the driver line (with the prefetch exposure figures)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6a; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
bash tools/gpu_env_ab.sh r6a fp32 conv_x3_kernel,conv_x3d_kernel "ELD_X3_BSLAB=0" "-" > $O/ab_bslab.txt 2>&1; cat $O/ab_bslab.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json
r = json.load(open('gpurun_out/r6a/bench.json'))
print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['alt_bf16']['ms_per_step'], r['alt_bf16']['roofline']['frac'])
print(r['roofline_sampler']['in_step_ms'], r['roofline_sampler']['exposed_ms'], r['roofline_sampler']['exposed_note'][-80:])
PY
