#!/bin/bash
# kernel durations of the decode stage (tools/time_stage.py) under rocprofv3 --kernel-trace --stats;
# environment switches of the library (IA_FUSED_ROWMAX_FILTER=0 ...) are passed through for A/B runs
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pst
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pst -- python $ROOT/tools/time_stage.py > /tmp/pst.log 2>&1
grep "us per pass" /tmp/pst.log
python - <<EOF
import csv,glob
f=glob.glob('/tmp/pst/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'ia::' in r['Name']: print('  %-60s calls %5s avg %8.1f us min %8.1f' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3))
EOF
