cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t13
( timeout 900 python -m pytest tests -m gpu -q ) > gpurun_out/t13/pytest.log 2>&1; grep -n "passed\|failed" gpurun_out/t13/pytest.log | tail -2
export ELD_AMD_ANY_PHILOX=1
bash tools/gpu_ab.sh t13/ab_fp32 "conv_x3d,wgrad8,conv_x3_kernel,noise_kernel<true, 185" fp32 eld_amd/libeld_amd.so tools/probe/lib_r03.so 2>&1 | tee gpurun_out/t13/ab_fp32.txt
bash tools/gpu_ab.sh t13/ab_bf16 "conv_bfd,wgrad8,noise_kernel<true, 185" bf16 eld_amd/libeld_amd.so tools/probe/lib_r03.so 2>&1 | tee gpurun_out/t13/ab_bf16.txt
