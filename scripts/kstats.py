"""Per-kernel totals of a rocprofv3 kernel-stats csv: python scripts/kstats.py <stats.csv> [steps] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('kernel time %.3f ms per step, %d launches per step' % (tot / 1e6 / steps, sum(int(r['Calls']) for r in rows) / steps))
for r in rows[:top]:
    print('%-86s %7.1f calls %9.3f ms  avg %9.1f us  %5.1f%%' % (r['Name'][:86], int(r['Calls']) / steps,
          int(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, 100.0 * int(r['TotalDurationNs']) / tot))
