#!/bin/bash
# gpurun_out/<tag> (tools/gpu_profile_set.sh) -> profiles/<round>_*: usage tools/profile_collect.sh <tag> <round, e.g. r03>
set -e
cd "$(dirname "$0")/.."
G=gpurun_out/$1; R=$2; P=profiles
cp $G/bench_fp32.json $P/${R}_bench_fp32.json
cp $G/bench_bf16.json $P/${R}_bench_bf16.json
cp $G/bench_small.json $P/${R}_bench_small_step.json
cp $G/noise_microbench.txt $P/${R}_sampler.txt
for v in fp32 bf16 small; do
  f=$(ls $G/prof/${v}_kernel_trace.csv)
  ( echo "# $R per-kernel time of one training step ($v): rocprofv3 --kernel-trace --stats of bench.py (see tools/gpu_profile_set.sh for the exact command); trace cut at the sampler launches"; echo
    python tools/step_breakdown.py $f 3 ) > $P/${R}_kernel_stats_$v.md
  cp $G/prof/${v}_kernel_stats.csv $P/${R}_kernel_stats_$v.csv
done
for v in fp32 bf16; do
  ( echo "# $R SQ counters per kernel ($v): rocprofv3 --pmc (one pass, 8 SQ counters) of bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-alt --precision $v (counter mode serialises dispatches: ms are PMC-mode durations)"; echo
    python tools/pmc_summary.py $G/pmc/sq_${v}_counter_collection.csv 24 ) > $P/${R}_pmc_$v.md
done
python tools/traffic_from_pmc.py $G/pmc/fetch_fp32_counter_collection.csv $G/pmc/write_fp32_counter_collection.csv $P/traffic.json 8 --sq $G/pmc/sq_fp32_counter_collection.csv > /dev/null
python tools/traffic_from_pmc.py $G/pmc/fetch_bf16_counter_collection.csv $G/pmc/write_bf16_counter_collection.csv $P/traffic.json 8 bf16 > /dev/null
( echo "# $R sampler SQ counters per model string: rocprofv3 --pmc, two passes, of tools/noise_microbench.py 8 (8 x 4x1424x2128 per launch; 23 launches per row; template flags: 57 = PGRU (two parameter sets: K = 2.29 and K = 0.1), 5 = Pg, 6 = pg, 4 = g, 0 = scale only)"; echo
  python tools/pmc_summary.py $G/pmc/noise_sq_counter_collection.csv 6; echo; python tools/pmc_summary.py $G/pmc/noise_sq2_counter_collection.csv 6 ) > $P/${R}_sampler_pmc_raw.md
ls $P | grep "^$R" | head -40
