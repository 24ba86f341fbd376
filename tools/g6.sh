mkdir -p gpurun_out
timeout 900 python tools/tune_gemm.py > gpurun_out/tune.log 2>&1; tail -2 gpurun_out/tune.log
timeout 600 python -m pytest tests/test_gpu_determinism.py -x -q 2>&1 | tail -5
B="python bench.py --no-train --no-cpu-baseline --no-pipeline --steps 20"
timeout 300 $B > gpurun_out/bench_det.json 2>gpurun_out/bench_det.err; cut -c1-150 gpurun_out/bench_det.json
timeout 300 $B > gpurun_out/bench_det2.json 2>/dev/null; cut -c1-150 gpurun_out/bench_det2.json
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_determinism.py 2>&1 | tail -15
