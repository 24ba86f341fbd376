mkdir -p gpurun_out
T="timeout 240"
$T python tools/find_nondeterminism.py --net r50 --path winograd > gpurun_out/nd_r50_wino.txt 2>&1
$T python tools/find_nondeterminism.py --net r50 --path winograd --deterministic > gpurun_out/nd_r50_wino_det.txt 2>&1
$T python tools/find_nondeterminism.py --net r50 --path module --batch 2 --deterministic > gpurun_out/nd_r50_module_det.txt 2>&1
$T python tools/find_nondeterminism.py --net x101-64x4d --path winograd --batch 2 --size 256x320 --deterministic > gpurun_out/nd_x101_wino_det.txt 2>&1
$T python tools/find_nondeterminism.py --net x101-64x4d --path module --batch 2 --size 256x320 --deterministic > gpurun_out/nd_x101_module_det.txt 2>&1
grep -c "NOT REPRO" gpurun_out/nd_*.txt
B="python bench.py --no-train --no-cpu-baseline --no-pipeline --steps 20"
$T $B > gpurun_out/bench_imm_det.json 2>/dev/null; cut -c1-120 gpurun_out/bench_imm_det.json
export MIOPEN_USER_DB_PATH=$PWD/gpurun_out/miopen_db; mkdir -p $MIOPEN_USER_DB_PATH
timeout 600 $B --miopen-find > gpurun_out/bench_find_det.json 2>/dev/null; cut -c1-120 gpurun_out/bench_find_det.json
ls -la $MIOPEN_USER_DB_PATH
$T $B > gpurun_out/bench_imm_det_userdb.json 2>/dev/null; cut -c1-120 gpurun_out/bench_imm_det_userdb.json
$T python tools/check_determinism.py --nets r50 --paths winograd > gpurun_out/det_c.txt 2>&1; cat gpurun_out/det_c.txt
