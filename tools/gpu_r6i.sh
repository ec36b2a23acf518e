#!/bin/bash
# round 6, batch i: conv_x3w_kernel (specialised waves) -- U-Net / parity / fuzz tests, same-box A/B against conv_x3_kernel<32,4> (ELD_X3W=0)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6i; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -3; tail -30 $O/pytest.log | grep -i "assert\|Error\|FAILED" | head -10
bash tools/gpu_env_ab.sh r6i fp32 conv_x3_kernel,conv_x3w_kernel "ELD_X3W=0" "-" > $O/ab_x3w.txt 2>&1; cat $O/ab_x3w.txt
