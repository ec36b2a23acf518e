cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_unet_gpu.py -q -x ) > gpurun_out/pytest_unet.log 2>&1
tail -3 gpurun_out/pytest_unet.log
LAYER_N=8 timeout 300 python tools/layer_bench.py 2>&1 | grep -E "upv|sum" 
