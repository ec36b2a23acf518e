cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t8
bash tools/gpu_env_ab.sh t8/ab_bf16 bf16 "wgrad8" "ELD_WGRAD_DMA=1" "ELD_WGRAD_DMA=2" "ELD_WGRAD_DMA=3" "ELD_WGRAD_DMA=0" 2>&1 | tee gpurun_out/t8/ab_bf16.txt
( timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -k "bf16" ) > gpurun_out/t8/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/t8/pytest.log | tail -2
