#!/bin/bash
# usage (from the repo root): gpurun --timeout 3000 -- "bash tools/gpu_profile_set.sh [tag]"; then tools/profile_collect.sh copies the summaries from gpurun_out/<tag> into profiles/
# round profile set: bench lines, kernel traces, sampler table, PMC passes for BOTH precisions (run from the repo root on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-final}; mkdir -p $O
python -c "import eld_amd; print(eld_amd.load_library().eld_build_info().decode())" > $O/build_info.txt 2>/dev/null      # src=<hash>: recorded beside the PMC figures (tools/traffic_from_pmc.py)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_fp32.json 2> $O/bench_fp32.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o fp32 -- python bench.py --no-cpu-baseline --no-alt > $O/bench_fp32_prof.json 2> $O/prof_fp32.err
timeout 600 python bench.py --precision bf16 --no-cpu-baseline > $O/bench_bf16.json 2> $O/bench_bf16.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bf16 -- python bench.py --precision bf16 --no-cpu-baseline > $O/bench_bf16_prof.json 2> $O/prof_bf16.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o small -- python bench.py --batch 1 --height 512 --width 512 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_small.json 2> $O/prof_small.err
timeout 200 python tools/noise_microbench.py 8 > $O/noise_microbench.txt 2>&1
SQ="SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES"
for prec in fp32 bf16; do
  B="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --precision $prec"
  timeout 500 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc -o sq_$prec -- $B > $O/pmc_sq_$prec.log 2>&1
  timeout 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc -o fetch_$prec -- $B > $O/pmc_fetch_$prec.log 2>&1
  timeout 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc -o write_$prec -- $B > $O/pmc_write_$prec.log 2>&1
done
# sampler counters (VALU instructions per pixel, issue / wait split, LDS)
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU --output-format csv -d $O/pmc -o noise_sq -- python tools/noise_microbench.py 8 > $O/pmc_noise.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM --output-format csv -d $O/pmc -o noise_sq2 -- python tools/noise_microbench.py 8 > $O/pmc_noise2.log 2>&1
ls $O $O/prof $O/pmc | head -60
cut -c1-300 $O/bench_fp32.json
# the N > 1 bench path end to end on ONE GPU: two ranks share the device over gloo (RCCL itself needs two devices: tests/test_dist_gpu.py)
ELD_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --batch 1 --height 512 --width 512 > $O/bench_2rank_gloo.json 2> $O/bench_2rank_gloo.err
cut -c1-400 $O/bench_2rank_gloo.json
