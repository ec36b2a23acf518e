cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/final; mkdir -p $O/pmc
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc -o noise_sq -- python tools/noise_microbench.py 8 > $O/pmc_noise.log 2>&1
ls $O/pmc | grep noise
