cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6y
( timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x ) > gpurun_out/r6y/pytest.log 2>&1; tail -3 gpurun_out/r6y/pytest.log
bash tools/gpu_env_ab.sh r6y fp32 "conv_x3_gemm" "ELD_GEMM_PIPE=1" "ELD_GEMM_PIPE=0" 2>&1 | tee gpurun_out/r6y/ab.txt
