#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e2e
timeout 600 bash tools/run_train_syn_gpu.sh gpurun_out/r03e2e/train_syn 2 0 > gpurun_out/r03e2e/train_syn.out 2>&1; echo "train_syn rc=$?"; tail -5 gpurun_out/r03e2e/train_syn.log | cut -c1-200
timeout 900 python tools/psnr_parity.py --iters 300 --out gpurun_out/r03e2e/psnr_parity > gpurun_out/r03e2e/psnr.out 2>&1; echo "psnr rc=$?"; tail -12 gpurun_out/r03e2e/psnr.out | cut -c1-250
