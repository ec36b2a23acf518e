#!/bin/bash
# round 6, batch j: conv_x3w_kernel<8,2> vs <4,4> vs conv_x3_kernel<32,4>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6j; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
bash tools/gpu_env_ab.sh r6j fp32 conv_x3_kernel,conv_x3w_kernel "ELD_X3W=0" "ELD_X3W=1" "-" > $O/ab_x3w.txt 2>&1; cat $O/ab_x3w.txt
