cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t11
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/t11/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/t11/pytest.log | tail -2
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/t11/prof -o small -- python bench.py --batch 1 --height 512 --width 512 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > gpurun_out/t11/bench_small.json 2> gpurun_out/t11/prof_small.err
cut -c1-200 gpurun_out/t11/bench_small.json
ELD_AMD_ANY_PHILOX=1 ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_r03.so timeout 300 python bench.py --batch 1 --height 512 --width 512 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > gpurun_out/t11/bench_small_r03.json 2>/dev/null; cut -c1-200 gpurun_out/t11/bench_small_r03.json
timeout 300 python bench.py --batch 1 --height 512 --width 512 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > gpurun_out/t11/bench_small_b.json 2>/dev/null; cut -c1-200 gpurun_out/t11/bench_small_b.json
