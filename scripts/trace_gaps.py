"""Diagnostic: idle time between kernels from a rocprofv3 --kernel-trace csv.
usage: python scripts/trace_gaps.py <kernel_trace.csv> [skip_fraction]
Prints busy / idle time over the tail of the trace and the idle time grouped by the kernel that FOLLOWS the gap."""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = rows[int(len(rows) * skip):]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
gaps = collections.defaultdict(lambda: [0, 0])
short = lambda n: n.split('(')[0].replace('void ', '')[:70]
prev_end = rows[0][1]
for s, e, n in rows[1:]:
    g = max(0, s - prev_end)
    gaps[short(n)][0] += g; gaps[short(n)][1] += 1
    prev_end = max(prev_end, e)
idle = sum(v[0] for v in gaps.values())
print('kernels %d  span %.2f ms  busy %.2f ms (%.1f %%)  idle %.2f ms' % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, idle / 1e6))
for k, (g, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print('%9.1f us idle before %5d x %-70s (%.1f us each)' % (g / 1e3, c, k, g / 1e3 / c))
busy_by = collections.defaultdict(lambda: [0, 0])
for s_, e_, n_ in rows:
    busy_by[short(n_)][0] += e_ - s_; busy_by[short(n_)][1] += 1
print('busy time by kernel:')
for k, (b, c) in sorted(busy_by.items(), key=lambda kv: -kv[1][0])[:30]:
    print('%9.1f us %5.1f %%  %5d x %-70s (%.1f us each)' % (b / 1e3, 100.0 * b / busy, c, k, b / 1e3 / c))
