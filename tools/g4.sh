cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -q -x ) > gpurun_out/pytest_unet.log 2>&1
tail -5 gpurun_out/pytest_unet.log
LAYER_N=8 timeout 300 python tools/layer_bench.py > gpurun_out/layer_bench.txt 2>&1
cat gpurun_out/layer_bench.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r02c -- python bench.py --no-cpu-baseline > gpurun_out/bench_c.json 2> gpurun_out/bench_c.err
cat gpurun_out/bench_c.json | cut -c1-300
