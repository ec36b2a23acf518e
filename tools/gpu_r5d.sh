#!/bin/bash
# round 5, call D: conv_x3_kernel<32,4> with the slab loads issued ahead of the halo loads (ELD_X3_BFIRST=1): parity subset, then same-box A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5d}; mkdir -p $O
( export ELD_X3_BFIRST=1; timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_fuzz_gpu.py -m gpu -q ) > $O/pytest_bfirst.log 2>&1; echo "BFIRST=1: $(tail -1 $O/pytest_bfirst.log)"
bash tools/gpu_env_ab.sh $(basename $O)/ab fp32 "conv_x3_kernel" "-" "ELD_X3_BFIRST=1" 2>&1 | tee $O/ab.txt
