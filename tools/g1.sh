cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest.log 2>&1
tail -3 gpurun_out/pytest.log
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r02a -- python bench.py --no-cpu-baseline > gpurun_out/prof_bench.log 2>&1
ls gpurun_out/prof
