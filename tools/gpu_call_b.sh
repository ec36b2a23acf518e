#!/bin/bash
# round-3 call B: full -m gpu suite, the driver's bench command, train_syn.py with reference-length epochs, the PSNR end-metric tool
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
( time timeout 1200 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; cut -c1-600 $O/bench.json; tail -3 $O/bench.err
timeout 600 bash tools/run_train_syn_gpu.sh $O/train_syn_gpu 2 0 1288 > /dev/null 2>&1; head -30 $O/train_syn_gpu.log
( time timeout 1500 python tools/psnr_parity.py --iters 300 --out $O/psnr_parity ) > $O/psnr_parity.log 2>&1; tail -16 $O/psnr_parity.log
