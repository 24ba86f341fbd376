"""durations of every launch of kernels matching a substring in the last N launches of a trace"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if sys.argv[2] in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Grid_Size_Y', '')))
rows.sort()
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
for s, e, k, gx, gy in rows[-n:]:
    print('%8.1f us  grid %s x %s  %s' % ((e - s) / 1e3, gx, gy, k[:60]))
