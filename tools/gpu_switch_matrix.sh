# the U-Net / parity GPU tests under every A/B switch of the fp32 kernels (the opt-in paths must not rot).  usage: gpurun --timeout 1800 -- "bash tools/gpu_switch_matrix.sh"
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/switches
for s in "-" "ELD_X3W=7" "ELD_X3W=0" "ELD_X3W=4" "ELD_WG8_ROWSHARE=1" "ELD_WG8_STREAM=3" "ELD_X3_WREUSE=0" "ELD_X3_STREAM=0" "ELD_XCD=0" "ELD_TILE_BAND=1" "ELD_TILE_BAND=8" "ELD_DEBUG_KERNEL_MASK=384"; do
  tag=$(echo "$s" | tr '= ' '__')
  ( if [ "$s" != "-" ]; then export $s; fi; timeout 600 python -m pytest tests/test_unet_gpu.py tests/test_parity_full_gpu.py -m gpu -q -x ) > gpurun_out/switches/pytest_$tag.log 2>&1
  echo "$s: $(grep -E 'passed|failed|error' gpurun_out/switches/pytest_$tag.log | tail -1)"
done | tee gpurun_out/switches/summary.txt
