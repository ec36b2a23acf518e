cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6x
for pf in 0 1; do
ELD_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 2961$pf bench.py --gpus 2 --steps 3 --warmup 1 --batch 1 --height 512 --width 512 --prefetch $pf > gpurun_out/r6x/bench_2rank_pf$pf.json 2> gpurun_out/r6x/bench_2rank_pf$pf.err
python - gpurun_out/r6x/bench_2rank_pf$pf.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], d['ms_per_step'], d.get('per_rank_ms_per_step'), d.get('allreduce',{}).get('ms_per_step_without_exchange'), d.get('allreduce',{}).get('exposed_ms'))
PY
done
ELD_AMD_LIB=$GRAFT_REPO_ROOT/tools/probe/libeld_dev.so timeout 600 python tools/convt_ablate.py 8 2>&1 | tee gpurun_out/r6x/convt_ablate.txt
