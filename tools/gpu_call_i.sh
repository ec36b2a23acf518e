#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03s}; mkdir -p $O
K="conv_bfw,conv_bfs,conv_bfd"
for n in 0 2 5 10 0 5; do
ELD_CONV_DBG=$((n*256)) bash tools/gpu_kstats.sh $O dev_stag${n} $K bf16 tools/probe/libeld_dev.so
done
