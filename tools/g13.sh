cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -q -x ) > gpurun_out/pytest_unet.log 2>&1
tail -3 gpurun_out/pytest_unet.log
LAYER_N=8 timeout 300 python tools/layer_bench.py > gpurun_out/layer_bench2.txt 2>&1; cat gpurun_out/layer_bench2.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r02e -- python bench.py --no-cpu-baseline --no-alt > gpurun_out/bench_e.json 2> gpurun_out/bench_e.err
cut -c1-300 gpurun_out/bench_e.json
