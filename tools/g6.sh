cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r02bf -- python bench.py --no-cpu-baseline --precision bf16 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
cat gpurun_out/bench_bf16.json | cut -c1-1200
