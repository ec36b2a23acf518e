#!/bin/bash
# round 6, batch b: ablation of the fp32 3x3 kernels (dev library)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6b; mkdir -p $O
ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 1500 python tools/conv_ablate.py 8 > $O/ablate.txt 2> $O/ablate.err; cat $O/ablate.txt; tail -3 $O/ablate.err
