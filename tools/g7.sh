cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ELD_DEV_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so
for d in 0 1 2 4 8 3 7 15; do echo "== ELD_NOISE_DBG=$d"; ELD_NOISE_DBG=$d timeout 120 python tools/noise_microbench.py 8 2>&1 | grep -E "PGRU K=2.29|Pg|model g " ; done > gpurun_out/noise_ablate.txt 2>&1
cat gpurun_out/noise_ablate.txt
