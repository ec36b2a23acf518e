cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t9
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/t9/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/t9/pytest.log | tail -2
bash tools/gpu_env_ab.sh t9/ab_fp32 fp32 "conv_x3" "ELD_X3D_32=1" "ELD_X3D_32=0" 2>&1 | tee gpurun_out/t9/ab_fp32.txt
