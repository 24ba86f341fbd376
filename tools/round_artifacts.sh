#!/bin/bash
# everything profiles/ keeps for a round, in one gpurun call (about 15 GPU-minutes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/round
mkdir -p $OUT
bash $ROOT/tools/profile_bench.sh > $OUT/profile_bench.log 2>&1
cp $ROOT/gpurun_out/profile/bench.json $OUT/bench_profiled_run.json
cp $ROOT/gpurun_out/profile/step_summary.txt $OUT/bench_step_summary.txt
cp $ROOT/gpurun_out/profile/kernel_stats_ia.csv $OUT/bench_kernel_stats_ia.csv
cp $ROOT/gpurun_out/profile/kernel_stats.csv $OUT/bench_kernel_stats_top60.csv
cp $ROOT/gpurun_out/profile/head_pmc.json $OUT/head_pmc.json
bash $ROOT/tools/collect_mfma_pmc.sh > $OUT/mfma_pmc.log 2>&1
cp $ROOT/gpurun_out/pmc/mfma_pmc.json $OUT/mfma_pmc.json
bash $ROOT/tools/profile_headloss.sh > $OUT/profile_headloss.log 2>&1
cp $ROOT/gpurun_out/profile/train_loss_part_all.txt $OUT/train_loss_part_summary.txt
cp $ROOT/gpurun_out/profile/train_loss_part_per_level.txt $OUT/train_loss_part_per_level_kernels.txt
cp $ROOT/gpurun_out/profile/train_loss_part_nhwc.txt $OUT/train_loss_part_channels_last_summary.txt
bash $ROOT/tools/profile_train.sh > $OUT/profile_train.log 2>&1
cp $ROOT/gpurun_out/profile/train_step_summary.txt $OUT/train_step_summary.txt
bash $ROOT/tools/collect_wino_pmc.sh > $OUT/wino_pmc.log 2>&1
cp $ROOT/gpurun_out/pmc/wino_pmc.json $OUT/wino_pmc.json
bash $ROOT/tools/collect_train_pmc.sh > $OUT/train_pmc.log 2>&1
cp $ROOT/gpurun_out/pmc/train_pmc.json $OUT/train_pmc.json
bash $ROOT/tools/profile_config.sh r101-bf16 > $OUT/profile_config3.log 2>&1
cp $ROOT/gpurun_out/profile/r101-bf16_step_summary.txt $OUT/config3_r101_bf16_b16_half_step_summary.txt 2>/dev/null
bash $ROOT/tools/profile_x101.sh > $OUT/profile_x101.log 2>&1
cp $ROOT/gpurun_out/profile/x101_step_summary.txt $OUT/x101_64x4d_third_step_summary.txt 2>/dev/null
python $ROOT/tools/try_configs.py > $OUT/other_configs.txt 2>&1
cd $ROOT && python bench.py > $OUT/bench_default_run.json 2> $OUT/bench_default_run.err
tail -3 $OUT/other_configs.txt; cut -c1-400 $OUT/bench_default_run.json
