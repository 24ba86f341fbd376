mkdir -p gpurun_out
timeout 300 python tools/time_strided.py frozen 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/time_strided.py all 2>&1 | grep -v amdgpu.ids
timeout 300 python tools/find_nondeterminism.py --net r101-bf16 --path winograd --batch 2 --size 256x320 > gpurun_out/nd_r101bf16.txt 2>&1; grep -v "^ok" gpurun_out/nd_r101bf16.txt | cut -c1-180 | head -20
