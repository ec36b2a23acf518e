#!/bin/bash
# dev: per-kernel time of the bf16 step under the ablation switches of the dev library (ELD_CONV_DBG bits: 1 skip epilogue, 2 skip compute, 4 skip staging)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-ablate}; mkdir -p $O
KPAT=${2:-conv_bfs}
for dbg in ${ABL_SET:-0 1 2 4 3}; do
  ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so ELD_CONV_DBG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$dbg -o t -- python bench.py --precision bf16 --no-cpu-baseline --steps 3 --warmup 1 > $O/bench$dbg.json 2> $O/err$dbg.txt
  python - $O/prof$dbg $dbg "$KPAT" <<'PY'
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)
for r in csv.DictReader(open(f[0])):
    if any(p in r['Name'] for p in sys.argv[3].split(',')):
        print('dbg %s  %-60s calls %5s avg %9.1f us' % (sys.argv[2], r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
