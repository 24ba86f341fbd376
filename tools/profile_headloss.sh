#!/bin/bash
# kernel summary of the loss part of a training iteration (tools/time_headloss.py, B = 4)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
for mode in all per_level nhwc; do
  rm -rf /tmp/phl
  arg=""; marker="k_box_ml<float, true>"
  if [ $mode = per_level ]; then arg="per_level"; marker="k_assign<true>"; fi
  if [ $mode = nhwc ]; then arg="nhwc"; marker="k_box_nhwc<true>"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/phl -- python $ROOT/tools/time_headloss.py 4 $arg > /tmp/phl.log 2>&1
  tail -2 /tmp/phl.log
  { tail -2 /tmp/phl.log | sed 's/^/# /'; python $ROOT/tools/summarize_trace.py /tmp/phl/*/*_kernel_trace.csv --steps 10 --marker "$marker" --top 30; } > $OUT/train_loss_part_$mode.txt
  grep "ia::" /tmp/phl/*/*_kernel_stats.csv | cut -d, -f1-4 > $OUT/train_loss_part_${mode}_stats.csv
  head -30 $OUT/train_loss_part_$mode.txt | cut -c1-160
done
