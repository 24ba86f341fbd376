#!/bin/bash
# Round profile of the bench command on an MI355X (run through gpurun).  Writes summaries to
# gpurun_out/profile/ (copy the ones to keep into profiles/):
#   bench.json              the JSON line of the profiled run
#   kernel_stats.csv        rocprofv3 --kernel-trace --stats of `bench.py --steps 10 --warmup 2`
#                           (whole process: includes MIOpen find-mode trial kernels of step 1)
#   step_summary.txt        the same trace reduced to per-step averages over the last 5 steps
#   head_pmc.json           FETCH_SIZE / WRITE_SIZE of the head kernels (separate --pmc passes)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profile
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
python $ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-train --no-other-configs > /tmp/warm.log 2>&1   # warms MIOpen's find db
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- \
    python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-train --no-other-configs --no-pipeline > /tmp/prof_bench.log 2>&1
grep "^{" /tmp/prof_bench.log > $OUT/bench.json
head -60 /tmp/prof_bench/*/*_kernel_stats.csv | cut -c1-260 > $OUT/kernel_stats.csv
grep "ia::" /tmp/prof_bench/*/*_kernel_stats.csv > $OUT/kernel_stats_ia.csv
python $ROOT/tools/summarize_trace.py /tmp/prof_bench/*/*_kernel_trace.csv --steps 5 > $OUT/step_summary.txt
if [ -z "$SKIP_PMC" ]; then
  bash $ROOT/tools/collect_pmc.sh 8 D > /tmp/pmc.log 2>&1
  cp $ROOT/gpurun_out/pmc/head_pmc.json $OUT/head_pmc.json
fi
cat $OUT/bench.json | cut -c1-600; cat $OUT/kernel_stats_ia.csv | cut -d, -f1-4; head -24 $OUT/step_summary.txt | cut -c1-150
