#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6k; mkdir -p $O
for m in 1 2; do for d in 1 0; do echo "== ELD_X3W=$m ELD_CONV_DBG=$d (1 = no producer priority)"; ELD_CONV_DBG=$d ELD_X3W=$m ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 600 python tools/x3w_prof.py 2>&1 | grep -A14 "^wg 0 consumer"; done; done > $O/prof.txt 2>&1; cat $O/prof.txt
