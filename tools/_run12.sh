cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t12
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/t12/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/t12/pytest.log | tail -2
bash tools/gpu_tilemodes.sh t12/modes_bf16 bf16 - t f 2>&1 | tee gpurun_out/t12/modes_bf16.txt
bash tools/gpu_tilemodes.sh t12/modes_fp32 fp32 - t f 2>&1 | tee gpurun_out/t12/modes_fp32.txt
