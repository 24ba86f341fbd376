timeout 900 python -m pytest tests/test_gpu_native_ops.py tests/test_gpu_safety.py -x -q 2>&1 | grep -v amdgpu.ids | tail -25
timeout 600 python -m pytest tests/test_gpu_losses.py tests/test_gpu_parity.py tests/test_gpu_e2e.py -x -q -k "focal_op or sigmoid or nms or config3_r101_bf16_full or bf16" 2>&1 | tail -5
