cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t14
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/t14/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/t14/pytest.log | tail -2
for i in 1 2; do for p in fp32 bf16; do timeout 300 python bench.py --precision $p --no-cpu-baseline --no-alt --steps 10 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$p', d['ms_per_step'])"; done; done
timeout 300 python bench.py --batch 1 --height 512 --width 512 --steps 20 --warmup 5 --no-cpu-baseline --no-alt 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('small', d['ms_per_step'])"
