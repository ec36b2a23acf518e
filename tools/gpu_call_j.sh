#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03t}; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -x -k "bf16" ) > $O/pytest_bf16.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest_bf16.log | tail -3
K="conv_bfw,conv_bfs,conv_bfd"
bash tools/gpu_kstats.sh $O base_1 $K bf16 tools/probe/lib_base.so
bash tools/gpu_kstats.sh $O new_1 $K bf16 eld_amd/libeld_amd.so
bash tools/gpu_kstats.sh $O base_2 $K bf16 tools/probe/lib_base.so
bash tools/gpu_kstats.sh $O new_2 $K bf16 eld_amd/libeld_amd.so
