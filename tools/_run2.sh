bash tools/gpu_r4.sh t2
bash tools/gpu_tilemodes.sh t2/modes_bf16 bf16 - f p q4 m24 2>&1 | tee gpurun_out/t2/modes_bf16.txt
bash tools/gpu_tilemodes.sh t2/modes_fp32 fp32 - f p m24 2>&1 | tee gpurun_out/t2/modes_fp32.txt
