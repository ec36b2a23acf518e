#!/bin/bash
# dev: one bench.py run under rocprofv3 --kernel-trace --stats; prints the step time and the average duration of the kernels matching a pattern list.
# usage: [ENV=..] gpu_kstats.sh <out-dir> <tag> <kernel-patterns,comma-separated> <precision> <lib.so> [extra bench args]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$1; tag=$2; KPAT=$3; PREC=$4; lib=$5; shift 5
mkdir -p $O
ELD_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -o t -- python bench.py --precision $PREC --no-cpu-baseline --no-alt --steps 3 --warmup 1 "$@" > $O/bench_$tag.json 2> $O/err_$tag.txt
python - $O/prof_$tag $tag "$KPAT" $O/bench_$tag.json <<'PY'
import csv, sys, glob, json
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)
try: ms = json.loads(open(sys.argv[4]).read())['ms_per_step']
except Exception: ms = None
print('%-22s step %s ms' % (sys.argv[2], ms))
if f:
    for r in csv.DictReader(open(f[0])):
        if any(p in r['Name'] for p in sys.argv[3].split(',')):
            print('   %-70s calls %5s avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY
