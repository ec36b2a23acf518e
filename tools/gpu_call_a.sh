#!/bin/bash
# round-3 call A: the whole -m gpu suite, the sampler alone (timing + SQ counters per model string), and the unmodified train_syn.py on the GPU
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 200 python tools/noise_microbench.py 8 > $O/noise_microbench.txt 2>&1; cat $O/noise_microbench.txt
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc -o noise_sq -- python tools/noise_microbench.py 8 > $O/pmc_noise.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc -o noise_sq2 -- python tools/noise_microbench.py 8 > $O/pmc_noise2.log 2>&1
tail -3 $O/pmc_noise2.log
timeout 600 bash tools/run_train_syn_gpu.sh $O/train_syn_gpu 2 0
ls $O $O/pmc
