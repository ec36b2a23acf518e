cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in B C; do echo "== variant $v"; ELD_DEV_LIB=$GRAFT_REPO_ROOT/tools/probe/lib_$v.so timeout 120 python tools/noise_microbench.py 8 2>&1 | grep model; done | tee gpurun_out/noise_variants.txt
