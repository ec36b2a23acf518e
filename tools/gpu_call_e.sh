#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03k}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o fp32 -- python bench.py --no-cpu-baseline --no-alt --steps 5 --warmup 2 > $O/bench_fp32.json 2> $O/prof_fp32.err
cut -c1-200 $O/bench_fp32.json
python - $O <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/prof/**/fp32_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:22]:
    print('%-70s calls %5s  total %9.3f ms  avg %9.1f us  %5.2f%%' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
