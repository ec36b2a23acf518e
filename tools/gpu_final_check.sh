#!/bin/bash
# the round's last call: the whole GPU suite, smoke() and the driver's bench command on the final tree
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final_check}; mkdir -p $O
python -c "import eld_amd; print(eld_amd.load_library().eld_build_info().decode())" > $O/build_info.txt; cat $O/build_info.txt | tail -c 40
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - $O/bench_default.json <<'PY'
import json, sys
r = json.load(open(sys.argv[1]))
print('value', r['value'], 'ms', r['ms_per_step'], 'frac', r['roofline']['frac'], 'traffic', r['roofline']['traffic'], 'bf16', r['alt_bf16']['ms_per_step'], r['alt_bf16']['roofline']['traffic'], 'eval', r['alt_eval_sweep']['fp32']['value'], r['alt_eval_sweep']['bf16']['value'], 'cpu', r['cpu_baseline']['value'])
PY
