"""Diagnostic: the kernels before every launch of a named kernel in a rocprofv3 kernel trace, with the idle gap in
front of each: python scripts/trace_around.py <kernel_trace.csv> <name substring> [n_before]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for i, r in enumerate(rows):
    if sys.argv[2] in r['Kernel_Name']:
        print('----')
        for j in range(max(0, i - n), min(len(rows), i + 3)):
            gap = (int(rows[j]['Start_Timestamp']) - int(rows[j - 1]['End_Timestamp'])) / 1e3 if j else 0.0
            dur = (int(rows[j]['End_Timestamp']) - int(rows[j]['Start_Timestamp'])) / 1e3
            print('%10.1f us gap  %10.1f us  %s' % (gap, dur, rows[j]['Kernel_Name'][:90]))
