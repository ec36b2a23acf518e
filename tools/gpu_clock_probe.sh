#!/bin/bash
# dev: effective shader clock per kernel (GRBM_GUI_ACTIVE / duration, MI355X_MICROARCH.md "DVFS give-back") of the bf16 step under the dev library's
# ablation switches: does a kernel's clock move with what it is asked to do?  usage (GPU box): gpu_clock_probe.sh <out-tag> "<ELD_CONV_DBG values>"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-clock}; mkdir -p $O
for d in ${2:-0 1 2 3}; do
  ELD_CONV_DBG=$d ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/c$d -o t -- python bench.py --precision ${3:-bf16} --no-cpu-baseline --no-alt --steps 2 --warmup 1 > $O/bench_$d.json 2> $O/err_$d.txt
  python - $O/c$d $d <<'PY'
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
    if not any(p in n for p in ('conv_bf', 'wgrad8', 'conv_x3', 'noise_kernel<true, 185')): continue
    a = acc[n[:64]]; a[0] += float(r['Counter_Value']); a[1] += d[0]; a[2] += 1
print('ELD_CONV_DBG=%s' % sys.argv[2])
for n, (cyc, ns, k) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:12]:
    print('   %-64s calls %4d avg %8.1f us  clock %.2f GHz' % (n, k, ns / k / 1e3, cyc / ns))
PY
done
