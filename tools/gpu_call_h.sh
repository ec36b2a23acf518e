#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r03r}; mkdir -p $O
( time timeout 600 python -m pytest tests/test_unet_gpu.py -m gpu -q -x -k "bit_for_bit" ) > $O/pytest_bf16.log 2>&1
grep -n "passed\|failed\|Error" $O/pytest_bf16.log | tail -3
K="conv_bfw,conv_bfs,conv_bfd"
bash tools/gpu_kstats.sh $O new_1 $K bf16 eld_amd/libeld_amd.so
bash tools/gpu_kstats.sh $O dev_0 $K bf16 tools/probe/libeld_dev.so
ELD_CONV_DBG=1 bash tools/gpu_kstats.sh $O dev_noepi $K bf16 tools/probe/libeld_dev.so
ELD_CONV_DBG=2 bash tools/gpu_kstats.sh $O dev_nomfma $K bf16 tools/probe/libeld_dev.so
ELD_CONV_DBG=3 bash tools/gpu_kstats.sh $O dev_dmaonly $K bf16 tools/probe/libeld_dev.so
bash tools/gpu_kstats.sh $O new_2 $K bf16 eld_amd/libeld_amd.so
SQ="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
timeout 300 rocprofv3 --pmc $SQ --output-format csv -d $O/pmc -o sq -- python bench.py --precision bf16 --no-cpu-baseline --no-alt --steps 1 --warmup 1 > $O/pmc.log 2>&1
python - $O/pmc <<'PY'
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/sq_counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f[0])):
    n = r['Kernel_Name']
    if any(p in n for p in ('conv_bfw', 'conv_bfs', 'conv_bfd')):
        acc[n[:60]][r['Counter_Name']] += float(r['Counter_Value'])
for n, d in acc.items():
    print(n)
    print('   ', {k: '%.3g' % v for k, v in d.items()})
PY
