cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc
B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline"
timeout 500 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc -o sq -- $B > gpurun_out/pmc/sq.log 2>&1
timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc -o fetch -- $B > gpurun_out/pmc/fetch.log 2>&1
timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc -o write -- $B > gpurun_out/pmc/write.log 2>&1
ls -la gpurun_out/pmc
