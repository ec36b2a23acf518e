#!/bin/bash
# round 5, call J: slope codes for the bf16 conv_bfs backward-data launches + the per-workspace codes state: parity suites, then same-box A/B through
# eld_debug_kernel_mask bit 7 (codes off) of ONE library
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5j}; mkdir -p $O
python -c "import __graft_entry__ as g; from eld_amd import _lib as L; assert L.build_src_hash() == g.source_hash(), 'stale library'" || exit 9
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_model_gpu.py tests/test_dropin_gpu.py -m gpu -q ) > $O/pytest.log 2>&1; tail -3 $O/pytest.log
bash tools/gpu_env_ab.sh $(basename $O)/ab bf16 "conv_bfs_kernel,conv_first_fwd_mma" - ELD_DEBUG_KERNEL_MASK=128 2>&1 | tee $O/ab.txt
