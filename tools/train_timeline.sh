#!/bin/bash
# ordered kernel timeline (start, duration, gap to the previous kernel) of the last training iteration
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pt
rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -- env TRAIN_ONLY=1 python $ROOT/tools/try_train_find.py 0 > /tmp/pt.log 2>&1
tail -1 /tmp/pt.log
mkdir -p $ROOT/gpurun_out
python $ROOT/tools/trace_timeline.py /tmp/pt/*/*_kernel_trace.csv "k_assign<true>" multi_tensor_apply | cut -c1-150 > $ROOT/gpurun_out/train_timeline.txt
wc -l $ROOT/gpurun_out/train_timeline.txt
