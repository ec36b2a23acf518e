#!/bin/bash
# round 6, batch h: streamed fragment blocks in wgrad8_kernel -- U-Net / parity tests, same-box A/B
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6h; mkdir -p $O
( time timeout 2400 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
bash tools/gpu_env_ab.sh r6h fp32 wgrad8_kernel "ELD_WG8_STREAM=0" "-" > $O/ab_wg8.txt 2>&1; cat $O/ab_wg8.txt
