#!/bin/bash
# kernel stats of the training step (BASELINE config 5, batch 4, one GPU).  CAUTION: MIOpen's find mode
# tries its naive reference kernels for every backward convolution under the profiler: ~10 GPU-minutes,
# and the --stats table is dominated by those trials (use a kernel trace + summarize_trace.py instead).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
python $ROOT/tools/time_train.py 4 > /dev/null 2>&1
rm -rf /tmp/pt
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $ROOT/tools/time_train.py 4 > /tmp/pt.log 2>&1
tail -2 /tmp/pt.log
head -40 /tmp/pt/*/*_kernel_stats.csv | cut -d, -f1-5 | cut -c1-170
