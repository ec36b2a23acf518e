#!/bin/bash
# round 5, call G: conv_x3_kernel<32,4> as two phase-locked 4-wave halves in one workgroup (ELD_X3_PP=1): small shapes under a short timeout first (a barrier
# mismatch would hang), then the parity suites, then the same-box A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5g}; mkdir -p $O
( export ELD_X3_PP=1; timeout 150 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -k "conv3x3_forward or conv3x3_backward_data" ) > $O/pytest_pp_small.log 2>&1; RC=$?; echo "PP small (rc $RC): $(tail -1 $O/pytest_pp_small.log)"
if [ $RC -ne 0 ]; then tail -30 $O/pytest_pp_small.log; exit 1; fi
( export ELD_X3_PP=1; timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_fuzz_gpu.py tests/test_model_gpu.py -m gpu -q ) > $O/pytest_pp.log 2>&1; echo "PP=1: $(tail -1 $O/pytest_pp.log)"
bash tools/gpu_env_ab.sh $(basename $O)/ab fp32 "conv_x3_kernel" "-" "ELD_X3_PP=1" 2>&1 | tee $O/ab.txt
