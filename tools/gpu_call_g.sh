#!/bin/bash
# round-3 call g: bf16 kernel tests of the new epilogues / conv_bfw / wgrad window reads, then same-box A/B against the previous build
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03q}; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -x -k "bf16" ) > $O/pytest_bf16.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest_bf16.log | tail -3
bash tools/gpu_ab.sh $1/ab_bf16 "conv_bfd,conv_bfs,conv_bfw,wgrad8_kernel<unsigned short,maxpool,head" bf16 tools/probe/lib_base.so eld_amd/libeld_amd.so
ELD_DEBUG_KERNEL_MASK=8 bash tools/gpu_ab.sh $1/ab_bf16_nobfw "conv_bfd_kernel<64" bf16 eld_amd/libeld_amd.so | head -8
for lib in tools/probe/lib_base.so eld_amd/libeld_amd.so; do
  ELD_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof32_$(basename $lib .so) -o t -- python bench.py --no-cpu-baseline --no-alt --steps 3 --warmup 1 > $O/bench32_$(basename $lib .so).json 2>/dev/null
  python - $O/prof32_$(basename $lib .so) $O/bench32_$(basename $lib .so).json <<'PY'
import csv, sys, glob, json
f = glob.glob(sys.argv[1] + '/**/t_kernel_stats.csv', recursive=True)
try: print('fp32 step', json.loads(open(sys.argv[2]).read())['ms_per_step'])
except Exception as e: print('bench failed', e)
for r in csv.DictReader(open(f[0])):
    if 'wgrad8_kernel<float' in r['Name']: print('   %-60s calls %5s avg %9.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
( time timeout 900 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -k "not bf16" ) > $O/pytest_fp32.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest_fp32.log | tail -3
