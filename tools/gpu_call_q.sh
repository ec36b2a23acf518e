#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04m}; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_model_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
bash tools/gpu_kstats.sh $O b16 "pack_all" bf16 eld_amd/libeld_amd.so
bash tools/gpu_kstats.sh $O f32 "pack_all" fp32 eld_amd/libeld_amd.so
bash tools/gpu_kstats.sh $O small "pack_all" fp32 eld_amd/libeld_amd.so --batch 1 --height 512 --width 512 --steps 20 --warmup 5
