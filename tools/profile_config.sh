#!/bin/bash
# Kernel summary of `bench.py --config $1` (default r101-bf16) as the driver would run it (frozen GEMM
# table, MIOpen immediate mode): per-step averages over the last 4 steps -> gpurun_out/profile/
CFG=${1:-r101-bf16}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
python $ROOT/bench.py --config $CFG --steps 3 --warmup 2 --no-cpu-baseline > /tmp/warm.log 2>&1
rm -rf /tmp/pcfg
rocprofv3 --kernel-trace --output-format csv -d /tmp/pcfg -- python $ROOT/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline > /tmp/pcfg.log 2>&1
mkdir -p $ROOT/gpurun_out/profile
grep "^{" /tmp/pcfg.log | cut -c1-300 > $ROOT/gpurun_out/profile/${CFG}_bench.json
python $ROOT/tools/summarize_trace.py /tmp/pcfg/*/*_kernel_trace.csv --steps 4 --marker "k_lazy_greedy" --top 45 > $ROOT/gpurun_out/profile/${CFG}_step_summary.txt
head -60 $ROOT/gpurun_out/profile/${CFG}_step_summary.txt | cut -c1-170
