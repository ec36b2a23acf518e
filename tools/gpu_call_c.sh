#!/bin/bash
# round-3 call C: bf16 kernel work -- bf16 tests, then the bf16 step under rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03c}; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x -k "bf16" ) > $O/pytest_bf16.log 2>&1
tail -4 $O/pytest_bf16.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bf16 -- python bench.py --precision bf16 --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_bf16.json 2> $O/prof_bf16.err
cut -c1-300 $O/bench_bf16.json
python - $O <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/prof/**/bf16_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel stats (%d kernels, %.3f ms total over all steps)' % (len(rows), tot / 1e6))
for r in rows[:32]:
    print('%-70s calls %5s  total %9.3f ms  avg %9.1f us  %5.2f%%' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
