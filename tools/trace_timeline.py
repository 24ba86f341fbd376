"""timeline (start, duration, gap to the previous kernel) of the kernels between two markers in
the LAST occurrence inside a rocprofv3 kernel trace CSV

    python tools/trace_timeline.py <kernel_trace.csv> k_rowmax k_lazy_greedy"""
import csv
import sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
a, b = sys.argv[2], sys.argv[3]
starts = [i for i, r in enumerate(rows) if a in r[2]]
i0 = starts[-1]
i1 = next(i for i in range(i0, len(rows)) if b in rows[i][2])
t0 = rows[i0][0]
prev_end = rows[i0 - 1][1]
for s, e, n in rows[i0 - 1:i1 + 1]:
    print('%9.1f us  dur %8.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:90]))
    prev_end = e
print('span %.1f us' % ((rows[i1][1] - t0) / 1e3))
