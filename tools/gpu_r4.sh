#!/bin/bash
# round 4 dev call: the -m gpu suite, then a same-box A/B of library builds (per-kernel averages, both precisions).
# usage: gpurun --timeout 1500 -- "bash tools/gpu_r4.sh <tag> libA.so libB.so ..."   (libs relative to the repo root; none: tests only)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-r4}; shift
O=gpurun_out/$TAG; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest.log | tail -5
if [ $# -gt 0 ]; then
  bash tools/gpu_ab.sh $TAG/ab_fp32 "conv_x3d,wgrad8,conv_x3_kernel" fp32 "$@" 2>&1 | tee $O/ab_fp32.txt
  bash tools/gpu_ab.sh $TAG/ab_bf16 "conv_bfd,wgrad8" bf16 "$@" 2>&1 | tee $O/ab_bf16.txt
fi
