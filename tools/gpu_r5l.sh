#!/bin/bash
# round 5, call L: the driver's bench command on the final build with the committed traffic.json (roofline.traffic through the hash gate)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5l}; mkdir -p $O
python -c "import __graft_entry__ as g; from eld_amd import _lib as L; assert L.build_src_hash() == g.source_hash(), 'stale library'; print(L.build_src_hash())" > $O/build.txt || exit 9
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_fp32.json 2> $O/bench_fp32.err; cut -c1-300 $O/bench_fp32.json
( time timeout 120 python bench.py --precision bf16 --no-cpu-baseline --no-alt ) > $O/bench_bf16.json 2> $O/bench_bf16.err; cut -c1-300 $O/bench_bf16.json
