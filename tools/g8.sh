timeout 900 python -m pytest tests/test_gpu_safety.py tests/test_gpu_parity.py -x -q 2>&1 | grep -v amdgpu.ids | tail -25
timeout 900 python -m pytest tests/test_gpu_e2e.py -q -k "config3_r101_bf16_full or x101_64x4d-backbone2-winograd" 2>&1 | tail -5
