#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6e; mkdir -p $O
ABLATE_DBGS=0,512,1024,1536,5,517,1029,1541,21,29 ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 1500 python tools/conv_ablate.py 8 > $O/ablate.txt 2> $O/ablate.err; cat $O/ablate.txt; tail -3 $O/ablate.err
