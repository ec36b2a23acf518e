#!/bin/bash
# round 5, call B: quick regression of what changed since call A, then same-box A/B of the experimental tile variants of the fp32 laggard layers
# NOTE: the experimental switches this script toggles were removed after the measurement (profiles/r05_ab_notes.md names the commits that carried them).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5b}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_model_gpu.py tests/test_fuzz_gpu.py -m gpu -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
for v in "ELD_X3D_32=2" "ELD_X3D_64W4=1"; do
  ( export $v; timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -k "conv3x3 or oracle or golden or fused or strip" ) > $O/pytest_$(echo $v | tr '=' '_').log 2>&1; echo "$v: $(tail -1 $O/pytest_$(echo $v | tr '=' '_').log)"
done
bash tools/gpu_env_ab.sh $(basename $O)/ab fp32 "conv_x3_kernel,conv_x3d_kernel<32,conv_x3d_kernel<64,wgrad_kernel<float" "-" "ELD_X3D_32=2" "ELD_X3D_64W4=1" 2>&1 | tee $O/ab.txt
