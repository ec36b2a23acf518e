#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04j}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_model_gpu.py tests/test_dropin_gpu.py -m gpu -q -x -k "fused or model or dropin or step or head or forward" ) > $O/pytest.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
K="head_,l1_kernel"
for e in 0 1 0 1; do
ELD_FUSED_HEAD=$e bash tools/gpu_kstats.sh $O b16_fused$e $K bf16 eld_amd/libeld_amd.so
done
for e in 0 1 0 1; do
ELD_FUSED_HEAD=$e bash tools/gpu_kstats.sh $O f32_fused$e $K fp32 eld_amd/libeld_amd.so
done
