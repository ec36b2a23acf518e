#!/bin/bash
# round 6, batch f: streamed fragment blocks -- GPU tests, same-box A/B of the fp32 step (ELD_X3_STREAM=0 = the round-5 loops)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6f; mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1; grep -n "passed\|failed\|error" $O/pytest.log | tail -3
bash tools/gpu_env_ab.sh r6f fp32 conv_x3_kernel,conv_x3d_kernel "ELD_X3_STREAM=0" "-" > $O/ab_stream.txt 2>&1; cat $O/ab_stream.txt
