"""Diagnostic: launches of the named kernels in launch order with their durations (last `n` of the trace):
python scripts/trace_list.py <kernel_trace.csv> <n> <name substring> ..."""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
n = int(sys.argv[2])
sel = [r for r in rows if any(k in r['Kernel_Name'] for k in sys.argv[3:])]
for r in sel[-n:]:
    print('%9.1f us  %s' % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:80]))
