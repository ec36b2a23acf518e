#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03z}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -x -k "not bf16" ) > $O/pytest_fp32.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest_fp32.log | tail -3
K="conv_x3"
bash tools/gpu_kstats.sh $O base_1 $K fp32 tools/probe/lib_base.so
bash tools/gpu_kstats.sh $O new_1 $K fp32 eld_amd/libeld_amd.so
bash tools/gpu_kstats.sh $O base_2 $K fp32 tools/probe/lib_base.so
bash tools/gpu_kstats.sh $O new_2 $K fp32 eld_amd/libeld_amd.so
