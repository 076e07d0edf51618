"""Diagnostic: every launch of the LAST complete training step of a kernel trace (the launches between the last two
`clip_adam_kernel`s), in order, with its duration and the idle gap before it:
python scripts/trace_step.py <kernel_trace.csv>"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'clip_adam_kernel' in r['Kernel_Name']]
a, b = ends[-2] + 1, ends[-1] + 1
prev = int(rows[a - 1]['End_Timestamp'])
busy = gap = 0.0
for i, r in enumerate(rows[a:b]):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('cvae::', '').replace('void ', '')
    print('%3d %8.1f us  gap %6.1f  grid %-8s %s' % (i, (e - s) / 1e3, (s - prev) / 1e3, r.get('Grid_Size_X', r.get('Grid_Size', '')), name[:70]))
    busy += (e - s) / 1e3; gap += max(0, s - prev) / 1e3
    prev = max(prev, e)
print('launches %d  busy %.1f us  gaps %.1f us' % (b - a, busy, gap))
agg = {}
for r in rows[a:b]:
    name = r['Kernel_Name'].replace('cvae::', '').replace('void ', '').split('(')[0][:60]
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n, t = agg.get(name, (0, 0.0))
    agg[name] = (n + 1, t + d)
print('--- by kernel')
for name, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%4d x %9.1f us  %s' % (n, t, name))
