# evidence run of a round (usage: gpurun --timeout 3600 -- "bash tools/gpu_round_evidence.sh <tag>"): full test suite, profile set, parity tables, end-metric training parity, eval sweep, unmodified train_syn.py
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
T=${1:-r06f}
mkdir -p gpurun_out/$T
( time timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/$T/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/$T/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/$T/smoke.log 2>&1; tail -1 gpurun_out/$T/smoke.log
bash tools/gpu_profile_set.sh $T > gpurun_out/$T/profile_set.log 2>&1; tail -3 gpurun_out/$T/profile_set.log | cut -c1-200
timeout 300 python tools/qa_microbench.py > gpurun_out/$T/qa_microbench.json 2> gpurun_out/$T/qa_microbench.err
timeout 900 python tools/psnr_parity.py --iters 300 --out gpurun_out/$T/psnr_parity > gpurun_out/$T/psnr_parity.log 2>&1; tail -3 gpurun_out/$T/psnr_parity.log
for p in fp32 bf16; do timeout 400 python tools/eval_sweep.py --precision $p --frames 2 > gpurun_out/$T/eval_sweep_$p.json 2> gpurun_out/$T/eval_sweep_$p.err; cut -c1-200 gpurun_out/$T/eval_sweep_$p.json; done
# (the unmodified train_syn.py run needs a staged copy of the reference tree on the GPU box: tools/stage_reference.sh + tools/run_train_syn_gpu.sh; round 5's log is profiles/r05_train_syn_gpu.log)
( time timeout 1500 python tools/fuzz_shapes.py 240 101 2 ) > gpurun_out/$T/fuzz_shapes.log 2>&1; tail -3 gpurun_out/$T/fuzz_shapes.log
