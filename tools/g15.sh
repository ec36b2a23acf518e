cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -q ) > gpurun_out/pytest.log 2>&1
tail -4 gpurun_out/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" 2>&1 | tail -3
