#!/bin/bash
# dev: per-kernel averages of the bf16 step under a list of ablation-switch values of the dev library (tools/build_dev.sh; ELD_CONV_DBG bits: see the
# kernels' ELD_DBG uses).  usage (on the GPU box): gpu_dbg_sweep.sh <out-tag> <kernel-patterns,comma-separated> "<dbg values>"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03v}; mkdir -p $O
K="${2:-conv_bfd}"
for d in ${3:-0 256 0 256}; do
ELD_CONV_DBG=$d bash tools/gpu_kstats.sh $O dev_dbg$d $K bf16 tools/probe/libeld_dev.so
done
