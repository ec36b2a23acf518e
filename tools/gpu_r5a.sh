#!/bin/bash
# round 5, call A: the whole GPU suite (no -x: every failure in one go), evaluation-kernel A/B, the bench line, both eval sweeps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5a}; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_fuzz_gpu.py ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log
( time timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q ) > $O/pytest_fuzz.log 2>&1; tail -8 $O/pytest_fuzz.log
timeout 300 python tools/qa_microbench.py > $O/qa_microbench.json 2> $O/qa_microbench.err; cat $O/qa_microbench.json | tr -d '\n' | cut -c1-1500; echo
timeout 900 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench_fp32.json 2> $O/bench_fp32.err; cut -c1-400 $O/bench_fp32.json
python - $O/bench_fp32.json <<'PY'
import json, sys
try:
    r = json.load(open(sys.argv[1]))
    print('value', r['value'], 'ms', r['ms_per_step'], 'roofline', r['roofline']['frac'], r['roofline'].get('traffic'), r['roofline'].get('traffic_source'))
    print('alt_bf16', r['alt_bf16']['ms_per_step'], r['alt_bf16']['roofline']['frac'])
    print('eval', json.dumps(r['alt_eval_sweep'])[:1500])
except Exception as e:
    print('bench parse failed', e)
PY
for p in fp32 bf16; do timeout 400 python tools/eval_sweep.py --precision $p --frames 2 > $O/eval_sweep_$p.json 2> $O/eval_sweep_$p.err; cut -c1-300 $O/eval_sweep_$p.json; tail -2 $O/eval_sweep_$p.err; done
