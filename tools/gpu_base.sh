#!/bin/bash
# dev: baseline of a build -- GPU tests, the driver's bench line, per-launch trace of one fp32 / bf16 step.  usage: gpu_base.sh <tag> [skip-tests]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; O=gpurun_out/$T; mkdir -p $O
if [ -z "$2" ]; then ( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -3; fi
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
for p in fp32 bf16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$p -o t -- python bench.py --precision $p --no-cpu-baseline --no-alt --steps 5 --warmup 1 > $O/bench_$p.json 2> $O/err_$p.txt
  f=$(find $O/prof_$p -name 't_kernel_trace.csv' | head -1)
  python tools/step_trace.py $f 3 $O/step_trace_$p.md > /dev/null
  python tools/step_breakdown.py $f 3 $O/kernel_stats_$p.md | head -3
  rm -rf $O/prof_$p
done
