#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c; mkdir -p $O
ABLATE_DBGS=0,32,2,34,3,35 ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 1500 python tools/conv_ablate.py 8 > $O/ablate.txt 2> $O/ablate.err; cat $O/ablate.txt; tail -3 $O/ablate.err
