mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_determinism.py -q -s -k "deeper or config3 or determinism or reproducible or im2col or strided" 2>&1 | grep -v "amdgpu.ids" | tail -60
timeout 900 python -m pytest tests/test_gpu_gconv.py -q -k benchmark_shapes 2>&1 | tail -5
