set -x
mkdir -p gpurun_out
python tools/tune_gemm.py > gpurun_out/tune.log 2>&1; tail -3 gpurun_out/tune.log
python tools/check_determinism.py > gpurun_out/det_a.txt 2>&1
python tools/check_determinism.py > gpurun_out/det_b.txt 2>&1
diff gpurun_out/det_a.txt gpurun_out/det_b.txt && echo SAME_ACROSS_PROCESSES
cat gpurun_out/det_a.txt
python bench.py --no-train --no-cpu-baseline --no-pipeline --steps 20 > gpurun_out/bench_frozen.json 2> gpurun_out/bench_frozen.err; cat gpurun_out/bench_frozen.json | cut -c1-400
python bench.py --no-train --no-cpu-baseline --no-pipeline --steps 20 --miopen-find > gpurun_out/bench_find.json 2> gpurun_out/bench_find.err; cut -c1-300 gpurun_out/bench_find.json
python bench.py --no-train --no-cpu-baseline --no-pipeline --steps 20 --miopen-find --gemm-tune all > gpurun_out/bench_find_all.json 2>/dev/null; cut -c1-300 gpurun_out/bench_find_all.json
