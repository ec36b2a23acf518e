#!/bin/bash
# dev: socket power (rocm-smi, 5 Hz) while bench.py runs the fp32 and the bf16 step; prints min / mean / max of the samples taken during the timed steps
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-power}; mkdir -p $O
for prec in fp32 bf16; do
  ( while true; do rocm-smi --showpower 2>/dev/null | grep -o "Power (W): [0-9.]*" | grep -o "[0-9.]*$"; sleep 0.2; done ) > $O/power_$prec.txt &
  P=$!
  timeout 300 python bench.py --precision $prec --no-cpu-baseline --no-alt --steps 100 --warmup 5 > $O/bench_$prec.json 2> $O/err_$prec.txt
  kill $P 2>/dev/null; wait $P 2>/dev/null
  python - $O/power_$prec.txt $prec $O/bench_$prec.json <<'PY'
import sys, json
v = [float(x) for x in open(sys.argv[1]).read().split() if x]
busy = [x for x in v if x > 600.0]
ms = json.loads(open(sys.argv[3]).read())['ms_per_step']
print('%s: %.2f ms/step; %d samples, %d above 600 W: min %.0f mean %.0f max %.0f W' % (sys.argv[2], ms, len(v), len(busy), min(busy or [0]), sum(busy) / max(len(busy), 1), max(busy or [0])))
PY
done
