#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03v}; mkdir -p $O
K="${2:-conv_bfd}"
for d in ${3:-0 256 0 256}; do
ELD_CONV_DBG=$d bash tools/gpu_kstats.sh $O dev_dbg$d $K bf16 tools/probe/libeld_dev.so
done
