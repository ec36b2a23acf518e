#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r04o}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -x -k "bf16" ) > $O/pytest.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest.log | tail -3
for e in 0 1 0 1; do
ELD_BFS_TWO_WG=$e bash tools/gpu_kstats.sh $O b16_two$e "conv_bfs" bf16 eld_amd/libeld_amd.so
done
