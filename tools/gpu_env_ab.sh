#!/bin/bash
# dev: per-kernel averages of the step under different environment settings (same box, interleaved).  usage: gpu_env_ab.sh <tag> <precision> <kernel-patterns> "VAR=a" "VAR=b" ...  ("-" = no setting)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; PREC=$2; KPAT=$3; shift 3
for round in 1 2; do
 for setting in "$@"; do
  tag=$(echo "$setting" | tr '= ' '__')_$round
  ( if [ "$setting" != "-" ]; then export $setting; fi
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o t -- python bench.py --precision $PREC --no-cpu-baseline --no-alt --steps 3 --warmup 1 > $O/bench_$tag.json 2> $O/err_$tag.txt )
  python - $O/prof_$tag $tag "$KPAT" $O/bench_$tag.json <<'PY'
import csv, sys, glob, json
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)
try: ms = json.loads(open(sys.argv[4]).read())['ms_per_step']
except Exception: ms = None
print('%-22s step %s ms' % (sys.argv[2], ms))
for r in csv.DictReader(open(f[0])):
    if any(p in r['Name'] for p in sys.argv[3].split(',')):
        print('   %-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $O/prof_$tag
 done
done
