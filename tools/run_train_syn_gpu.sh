#!/bin/bash
# On the GPU box: the reference's UNMODIFIED train_syn.py (train_syn.py:100-113 loop) through eld_amd.launch with all four plugins, reference
# defaults (--batchSize 1, --nThreads 8 forked DataLoader workers), on-the-fly PGRU noise, in-memory LMDB stand-in (no SID data here).
# usage: bash tools/run_train_syn_gpu.sh <out-prefix> [epochs] [iters-per-epoch] [lmdb-entries = epoch length, default 1288 like the reference]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/train_syn_gpu}; EP=${2:-2}; IT=${3:-0}
REF=oracle/_ref/reference_tree
if [ ! -f $REF/train_syn.py ]; then echo "no staged reference under $REF (tools/stage_reference.sh)"; exit 3; fi
S=$(date +%s.%N)
( cd /tmp && PYTHONPATH=$GRAFT_REPO_ROOT python -m eld_amd.launch --ref $GRAFT_REPO_ROOT/$REF --plugins noise,arch,model,data --cwd /tmp/eld_run_$$ \
    --stop-after-epochs $EP --max-iters-per-epoch $IT --lmdb-entries ${4:-1288} -- --name t --include 4 --noise PGRU --no-log ) > $O.raw 2>&1
RC=$?
E=$(date +%s.%N)
# the progress bar rewrites its line with \r: keep the last state of every line
tr '\r' '\n' < $O.raw | grep -v '^\s*$' | awk 'length($0) < 400' > $O.lines
( echo "# python -m eld_amd.launch --ref <staged reference> --plugins noise,arch,model,data --stop-after-epochs $EP --max-iters-per-epoch $IT --lmdb-entries ${4:-1288} -- --name t --include 4 --noise PGRU --no-log"
  echo "# exit code $RC, wall $(python -c "print(round($E - $S, 1))") s (includes python start-up, library load, first-touch)"
  grep -n "eld_amd\|\[i\]\|Epoch\|epoch\|Time\|learning rate\|Traceback\|Error" $O.lines | head -60
  echo "# --- per-epoch: last progress line (Tot = epoch time as the reference's progress bar reports it)"
  grep -n "1288/1288\|Tot:" $O.lines | awk -F: '/[0-9]+\/[0-9]+ *$/ {split($0, a, "|"); print}' | grep -E " ([0-9]+)/\1 *$" | head -8
  echo "# --- last lines"; tail -12 $O.lines ) > $O.log
python - $O.raw <<'PY' >> $O.log
import re, sys, subprocess
maps = open('/proc/self/maps').read()
print('# libeld_amd.so path check:', subprocess.run(['bash', '-c', 'ls -la eld_amd/libeld_amd.so'], capture_output=True, text=True).stdout.strip())
PY
tail -30 $O.log
exit $RC
