#!/bin/bash
# round 5, call F: transposed-conv GEMM variants (ELD_X3G: 1 = 16x32x128 tile, 2 = that tile + loads two stages ahead, 3 = 8x32x64 tile + loads two stages ahead)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5f}; mkdir -p $O
for v in 1 2 3; do
  ( export ELD_X3G=$v; timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_fuzz_gpu.py -m gpu -q -k "transpose or oracle or golden or fuzz or random" ) > $O/pytest_x3g$v.log 2>&1; echo "ELD_X3G=$v: $(tail -1 $O/pytest_x3g$v.log)"
done
bash tools/gpu_env_ab.sh $(basename $O)/ab fp32 "conv_x3_gemm_kernel" "-" "ELD_X3G=1" "ELD_X3G=2" "ELD_X3G=3" 2>&1 | tee $O/ab.txt
