#!/bin/bash
# usage: gpurun --timeout 2400 -- "bash tools/gpu_tests.sh"  (the whole -m gpu suite)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/pytest.log 2>&1
tail -8 gpurun_out/pytest.log
