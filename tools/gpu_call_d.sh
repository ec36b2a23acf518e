#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03d}; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x -k "bf16" ) > $O/pytest_bf16.log 2>&1
grep -n "passed\|failed" $O/pytest_bf16.log | tail -2
bash tools/gpu_ablate.sh $1/abl "${2:-conv_bfs}"
