"""Diagnostic: the kernel sequence of ONE training step (between two clip_adam launches) from a rocprofv3 kernel trace.
usage: python scripts/trace_step.py <kernel_trace.csv> [which_step]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'],
                     r.get('Grid_Size_X', r.get('Grid_Size', '')), r.get('Workgroup_Size_X', r.get('Workgroup_Size', ''))))
rows.sort()
marks = [i for i, r in enumerate(rows) if 'clip_adam' in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) - 2
a, b = marks[k] + 1, marks[k + 1] + 1
t0 = rows[a][0]
prev = rows[a - 1][1]
for s, e, n, g, w in rows[a:b]:
    print('%9.1f  +%6.1f  %8.1f us  grid %8s/%-5s %s' % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, g, w,
                                                     n.replace('void ', '').replace('cvae::', '')[:90]))
    prev = e
print('step span %.1f us' % ((rows[b - 1][1] - rows[a - 1][1]) / 1e3))
