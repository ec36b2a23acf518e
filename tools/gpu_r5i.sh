#!/bin/bash
# round 5, call I: 2-bit slope codes for the fp32 backward-data epilogues of levels 0 / 1: the parity suites, then same-box A/B against the library without them
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5i}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_fuzz_gpu.py tests/test_model_gpu.py tests/test_dropin_gpu.py -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/gpu_ab.sh $(basename $O)/ab "conv_x3_kernel,conv_x3d_kernel<64,conv_first_fwd" fp32 tools/probe/lib_base.so tools/probe/lib_codes.so 2>&1 | tee $O/ab.txt
