#!/bin/bash
# per-iteration kernel summary of the training step (BASELINE config 5, batch 4, one GPU).
# MIOpen's find mode re-times its candidates (including naive reference kernels) in every new
# process, so the first iterations are slow under the profiler (~10 GPU-minutes in total); the
# summary only covers the last iterations (marker: the target-assignment kernel, once per iteration).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pt
TRAIN_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- python $ROOT/tools/time_train.py 4 > /tmp/pt.log 2>&1
tail -2 /tmp/pt.log
mkdir -p $ROOT/gpurun_out/profile
python $ROOT/tools/summarize_trace.py /tmp/pt/*/*_kernel_trace.csv --steps 4 --marker "k_assign<true>" --top 40 \
    > $ROOT/gpurun_out/profile/train_step_summary.txt
head -44 $ROOT/gpurun_out/profile/train_step_summary.txt | cut -c1-150
