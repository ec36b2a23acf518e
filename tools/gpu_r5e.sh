#!/bin/bash
# round 5, call E: more evidence on the final build -- a larger shape fuzz, a longer end-metric training parity, the bench line on one more board
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5e}; mkdir -p $O
python -c "import eld_amd; print(eld_amd.load_library().eld_build_info().decode())" > $O/build_info.txt
( time timeout 900 python tools/fuzz_shapes.py 120 7 3 ) > $O/fuzz_a.log 2>&1; tail -3 $O/fuzz_a.log
( time timeout 600 python tools/fuzz_shapes.py 120 23 1 ) > $O/fuzz_b.log 2>&1; tail -3 $O/fuzz_b.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_fp32.json 2> $O/bench_fp32.err; cut -c1-200 $O/bench_fp32.json
timeout 1500 python tools/psnr_parity.py --iters 800 --out $O/psnr_parity_800 > $O/psnr_parity_800.log 2>&1; tail -4 $O/psnr_parity_800.log
