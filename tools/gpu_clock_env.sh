#!/bin/bash
# dev: effective shader clock per kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration) of the fp32 step under environment settings (same box, interleaved).
# usage: gpu_clock_env.sh <tag> "VAR=a" "VAR=b" ...
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O; shift
for round in 1 2; do
 for setting in "$@"; do
  tag=$(echo "$setting" | tr '= ' '__')_$round
  ( if [ "$setting" != "-" ]; then export $setting; fi
    timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/c_$tag -o t -- python bench.py --no-cpu-baseline --no-alt --steps 2 --warmup 1 > $O/bench_$tag.json 2> $O/err_$tag.txt )
  python - $O/c_$tag $tag <<'PY'
import csv, sys, glob, collections
cc = glob.glob(sys.argv[1] + '/**/t_counter_collection.csv', recursive=True)
kt = glob.glob(sys.argv[1] + '/**/t_kernel_trace.csv', recursive=True)
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'])
acc = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
    d = dur.get(r['Dispatch_Id'])
    if not d: continue
    n = d[1]
    if not any(p in n for p in ('conv_x3', 'wgrad8')): continue
    a = acc[n.replace('void ', '').replace('(anonymous namespace)::', '')[:52]]; a[0] += float(r['Counter_Value']); a[1] += d[0]; a[2] += 1
print(sys.argv[2])
for n, (cyc, ns, k) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:8]:
    print('   %-52s calls %4d avg %8.1f us  clock %.3f GHz' % (n, k, ns / k / 1e3, cyc / 8 / ns))
PY
  rm -rf $O/c_$tag
 done
done
