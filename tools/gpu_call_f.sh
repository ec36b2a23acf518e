#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03l}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -x -k "not bf16" ) > $O/pytest.log 2>&1
grep -n "passed\|failed" $O/pytest.log | tail -2
bash tools/gpu_ab.sh $1/ab "conv_x3,wgrad8_kernel<float, 4" fp32 tools/probe/libeld_A.so tools/probe/libeld_B.so
