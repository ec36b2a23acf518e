cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_noise_gpu.py tests/test_abi.py -q -x ) > gpurun_out/pytest_noise.log 2>&1
tail -15 gpurun_out/pytest_noise.log
timeout 120 python tools/noise_microbench.py 8 2>&1 | tee gpurun_out/noise_microbench.txt
