cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export ELD_DIST_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --batch 2 > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank_gloo.err
echo rc=$?
cut -c1-1800 gpurun_out/bench_2rank_gloo.json
tail -5 gpurun_out/bench_2rank_gloo.err
