#!/bin/bash
# The ONE parametrised GPU driver of the dev loop (rounds 4-6 each left a dozen one-off gpu_rNx.sh scripts: they are in git history, this replaces them).
#   gpurun --timeout 3600 -- "bash tools/gpu_round.sh <tag> <step> [args...]"
# steps:
#   tests [pytest args]                      pytest -m gpu -q -x (default: the whole suite)
#   bench                                    the driver's command (python bench.py) -> gpurun_out/<tag>/bench.json, headline printed
#   trace <fp32|bf16>                        per-launch trace + per-kernel table of one step (tools/step_trace.py, tools/step_breakdown.py)
#   abenv <prec> <kernel,patterns> <env>...  same-box A/B of environment settings, interleaved, two rounds (tools/gpu_env_ab.sh); "-" = no setting
#   ablib <prec> <kernel,patterns> <lib>...  same-box A/B of library builds (tools/gpu_ab.sh; tools/build_variant.sh makes them)
#   ablate [dbg,list]                        tools/conv_ablate.py on the dev library (tools/build_dev.sh first)
#   x3wprof                                  tools/x3w_prof.py on the dev library: stage timeline of conv_x3w_kernel
#   ablatet [dbg,list]                       tools/convt_ablate.py on the dev library: the transposed convolutions
#   clock <env>...                           effective shader clock per kernel under environment settings (tools/gpu_clock_env.sh)
#   switches                                 the U-Net / parity tests under every A/B switch (tools/gpu_switch_matrix.sh)
#   evidence                                 the round's evidence set (tools/gpu_round_evidence.sh <tag>; then tools/profile_collect.sh <tag> <round>)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=$1; STEP=$2; shift 2
O=gpurun_out/$T; mkdir -p $O
case $STEP in
  tests) ( time timeout 2400 python -m pytest ${@:-tests} -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -3 ;;
  bench) timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; python -c "
import json; r = json.load(open('$O/bench.json')); print(r['value'], r['ms_per_step'], r['roofline']['frac'], r['alt_bf16']['ms_per_step'], r['alt_bf16']['roofline']['frac'])" ;;
  trace) p=${1:-fp32}; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$p -o t -- python bench.py --precision $p --no-cpu-baseline --no-alt --steps 5 --warmup 1 > $O/bench_$p.json 2> $O/err_$p.txt
         f=$(find $O/prof_$p -name 't_kernel_trace.csv' | head -1); python tools/step_trace.py $f 3 $O/step_trace_$p.md > /dev/null; python tools/step_breakdown.py $f 3 $O/kernel_stats_$p.md | head -30; rm -rf $O/prof_$p ;;
  abenv) bash tools/gpu_env_ab.sh $T "$@" | tee $O/abenv.txt ;;
  ablib) K=$2; P=$1; shift 2; bash tools/gpu_ab.sh $T $K $P "$@" | tee $O/ablib.txt ;;
  ablate) ABLATE_DBGS=${1:-0,64,1,2,3,4,16,8,5,18} ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 1500 python tools/conv_ablate.py 8 2>&1 | tee $O/ablate.txt ;;
  x3wprof) ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 600 python tools/x3w_prof.py 2>&1 | tee $O/x3wprof.txt ;;
  ablatet) ABLATE_DBGS=${1:-0,1,2,3,4,16,18,20,22} ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 900 python tools/convt_ablate.py 8 2>&1 | tee $O/ablatet.txt ;;
  clock) bash tools/gpu_clock_env.sh $T "$@" | tee $O/clock.txt ;;
  switches) bash tools/gpu_switch_matrix.sh | tee $O/switches.txt ;;
  evidence) bash tools/gpu_round_evidence.sh $T ;;
  *) echo "unknown step $STEP"; exit 2 ;;
esac
