cd /tmp; export TMPDIR=/tmp
$GRAFT_REPO_ROOT/tools/ubench/rowmax_bench 10
rm -rf /tmp/rb; rocprofv3 --kernel-trace --output-format csv -d /tmp/rb -- $GRAFT_REPO_ROOT/tools/ubench/rowmax_bench 10 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/rb/*/*kernel_trace.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_rowmax_nhwc' in r['Kernel_Name']:
        agg[int(r['LDS_Block_Size'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k in sorted(agg):
    v = agg[k][2:]
    print('LDS %6d B: %2d launches  avg %.1f us  min %.1f us' % (k, len(v), sum(v) / len(v), min(v)))
PY
