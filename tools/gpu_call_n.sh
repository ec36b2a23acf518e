#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04e}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
for l in tools/probe/lib_head.so eld_amd/libeld_amd.so tools/probe/lib_head.so eld_amd/libeld_amd.so; do
bash tools/gpu_kstats.sh $O w16_$(basename $l .so) "wgrad8_kernel<unsigned short" bf16 $l
done
for l in tools/probe/lib_head.so eld_amd/libeld_amd.so tools/probe/lib_head.so eld_amd/libeld_amd.so; do
bash tools/gpu_kstats.sh $O w32_$(basename $l .so) "wgrad8_kernel<float" fp32 $l
done
