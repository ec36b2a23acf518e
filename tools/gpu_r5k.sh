#!/bin/bash
# round 5, call K: shape fuzz of the final build (2 x 120 shapes, two seeds / size classes as tools/gpu_r5e.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5k}; mkdir -p $O
python -c "import __graft_entry__ as g; from eld_amd import _lib as L; assert L.build_src_hash() == g.source_hash(), 'stale library'; print(L.build_src_hash())" > $O/build.txt || exit 9
( time timeout 400 python tools/fuzz_shapes.py 120 7 3 ) > $O/fuzz_a.log 2>&1; tail -4 $O/fuzz_a.log
( time timeout 300 python tools/fuzz_shapes.py 120 23 1 ) > $O/fuzz_b.log 2>&1; tail -4 $O/fuzz_b.log
