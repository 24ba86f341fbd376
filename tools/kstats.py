"""print calls / average duration (us) of the kernels matching a substring from a rocprofv3
kernel_stats.csv (names contain commas, so no cut/awk)"""
import csv
import glob
import sys
pat = sys.argv[2] if len(sys.argv) > 2 else 'ia::'
for f in glob.glob(sys.argv[1]):
    for r in csv.DictReader(open(f)):
        if pat in r['Name']:
            print('%-70s calls %5s  avg %9.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
