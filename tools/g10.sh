cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_model_gpu.py -q -x -k "bf16" ) > gpurun_out/pytest_bf16.log 2>&1
tail -15 gpurun_out/pytest_bf16.log
timeout 300 python tools/bf16_bench.py 8 2>&1 | tail -3
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r02bf2 -- python bench.py --no-cpu-baseline --precision bf16 > gpurun_out/bench_bf16b.json 2> gpurun_out/bench_bf16b.err
cut -c1-400 gpurun_out/bench_bf16b.json
