#!/bin/bash
# round 5, call C: batched weight-gradient reduction (small steps) + counted halo wait (ELD_X3D_KEEPA): regression, then same-box A/B on the full frame and on the 512 x 512 patch
# NOTE: the experimental switches this script toggles were removed after the measurement (profiles/r05_ab_notes.md names the commits that carried them).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r5c}; mkdir -p $O
( time timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_model_gpu.py tests/test_fuzz_gpu.py tests/test_dropin_gpu.py -m gpu -q ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
( export ELD_X3D_KEEPA=1; timeout 900 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py tests/test_fuzz_gpu.py -m gpu -q ) > $O/pytest_keepa.log 2>&1; echo "KEEPA=1: $(tail -1 $O/pytest_keepa.log)"
bash tools/gpu_env_ab.sh $(basename $O)/ab fp32 "conv_x3d_kernel" "-" "ELD_X3D_KEEPA=1" 2>&1 | tee $O/ab.txt
for v in 0 1; do for r in 1 2; do
  ( export ELD_X3D_KEEPA=$v; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/small_$v$r -o t -- python bench.py --batch 1 --height 512 --width 512 --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/small_$v$r.json 2> $O/small_$v$r.err )
  python - $O/small_$v$r $O/small_$v$r.json $v <<'PY'
import csv, sys, glob, json
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
try: ms = json.loads(open(sys.argv[2]).read())['ms_per_step']
except Exception: ms = None
calls = sum(int(r['Calls']) for r in rows)
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('small step KEEPA=%s: %s ms per step; %d kernel launches in the trace (25 steps + set-up), kernel time %.2f ms' % (sys.argv[3], ms, calls, tot / 1e6))
for r in rows[:8]: print('   %-64s calls %5s avg %8.1f us' % (r['Name'][:64], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done; done
