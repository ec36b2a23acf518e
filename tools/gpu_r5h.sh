#!/bin/bash
# round 5, call H: conv_x3_kernel<32,2> -- 8-row tiles, three 4-wave workgroups per CU (ELD_X3_TH8=1): parity subset, then the same-box A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5h}; mkdir -p $O
( export ELD_X3_TH8=1; timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_fuzz_gpu.py -m gpu -q ) > $O/pytest_th8.log 2>&1; echo "TH8=1: $(tail -1 $O/pytest_th8.log)"
bash tools/gpu_env_ab.sh $(basename $O)/ab fp32 "conv_x3_kernel" "-" "ELD_X3_TH8=1" 2>&1 | tee $O/ab.txt
