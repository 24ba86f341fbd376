#!/bin/bash
# per-iteration kernel summary of the training step (BASELINE config 5, batch 4, one GPU) as
# bench.py's `train` sub-record runs it (MIOpen immediate mode: no find-mode trials).  The summary
# covers the last iterations (marker: the target-assignment kernel, once per iteration).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- env TRAIN_ONLY=1 python $ROOT/tools/try_train_find.py ${FIND:-0} > /tmp/pt.log 2>&1
tail -2 /tmp/pt.log
mkdir -p $ROOT/gpurun_out/profile
python $ROOT/tools/summarize_trace.py /tmp/pt/*/*_kernel_trace.csv --steps 4 --marker "k_assign<true>" --top 60 \
    > $ROOT/gpurun_out/profile/train_step_summary.txt
(head -24; echo ...; tail -10) < $ROOT/gpurun_out/profile/train_step_summary.txt | cut -c1-150
