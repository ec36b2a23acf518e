#!/bin/bash
# dev: per-kernel averages of the step under different ELD_CONV_TILES modes (same box).  usage: gpu_tilemodes.sh <tag> <precision> mode1 mode2 ...  (mode "-" = default)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; PREC=$2; shift 2
for round in 1 2; do
 for mode in "$@"; do
  tag=${mode}_$round
  if [ "$mode" = "-" ]; then unset ELD_CONV_TILES; else export ELD_CONV_TILES=$mode; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o t -- python bench.py --precision $PREC --no-cpu-baseline --no-alt --steps 3 --warmup 1 > $O/bench_$tag.json 2> $O/err_$tag.txt
  python - $O/prof_$tag $tag "conv_x3d,conv_bfd,wgrad8" $O/bench_$tag.json <<'PY'
import csv, sys, glob, json
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)
try: ms = json.loads(open(sys.argv[4]).read())['ms_per_step']
except Exception: ms = None
print('%-22s step %s ms' % (sys.argv[2], ms))
for r in csv.DictReader(open(f[0])):
    if any(p in r['Name'] for p in sys.argv[3].split(',')):
        print('   %-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
 done
done
