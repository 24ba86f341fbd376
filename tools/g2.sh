mkdir -p gpurun_out
python tools/find_nondeterminism.py --net r50 --path winograd > gpurun_out/nd_r50_wino.txt 2>&1
python tools/find_nondeterminism.py --net r50 --path winograd --deterministic > gpurun_out/nd_r50_wino_det.txt 2>&1
python tools/find_nondeterminism.py --net r50 --path module --batch 2 > gpurun_out/nd_r50_module.txt 2>&1
python tools/find_nondeterminism.py --net r50 --path module --batch 2 --deterministic > gpurun_out/nd_r50_module_det.txt 2>&1
python tools/find_nondeterminism.py --net x101-64x4d --path winograd --deterministic > gpurun_out/nd_x101_wino_det.txt 2>&1
python tools/find_nondeterminism.py --net r101-bf16 --path winograd --batch 16 --deterministic > gpurun_out/nd_r101bf16_wino_det.txt 2>&1
grep -c "NOT REPRO" gpurun_out/nd_*.txt
