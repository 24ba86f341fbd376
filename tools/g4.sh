mkdir -p gpurun_out
T="timeout 600"
$T python -m pytest tests/test_gpu_determinism.py -x -q 2>&1 | tail -15
B="python bench.py --no-train --no-cpu-baseline --no-pipeline --steps 20"
timeout 300 $B > gpurun_out/bench_det.json 2>gpurun_out/bench_det.err; cut -c1-150 gpurun_out/bench_det.json; tail -3 gpurun_out/bench_det.err
timeout 300 python tools/find_nondeterminism.py --net r50 --path winograd > gpurun_out/nd_r50_wino2.txt 2>&1; grep -v "^ok" gpurun_out/nd_r50_wino2.txt | cut -c1-160| tail -5
