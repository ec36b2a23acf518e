#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6l; mkdir -p $O
bash tools/gpu_env_ab.sh r6l fp32 conv_x3_kernel,conv_x3w_kernel "ELD_X3W=0" "ELD_X3W=3" "-" > $O/ab_x3w.txt 2>&1; cat $O/ab_x3w.txt
