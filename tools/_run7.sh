cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t7
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/t7/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/t7/pytest.log | tail -2
bash tools/gpu_env_ab.sh t7/ab_bf16 bf16 "wgrad8" "ELD_WGRAD_DMA=1" "ELD_WGRAD_DMA=0" 2>&1 | tee gpurun_out/t7/ab_bf16.txt
