"""Diagnostic: host-side HIP API calls from a rocprofv3 --hip-runtime-trace csv: total time per API name and the
longest individual calls (a launch that blocks the host shows up here).
usage: python scripts/trace_hip_api.py <hip_api_trace.csv> [skip_fraction]"""
import csv, sys, collections
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Function']))
rows.sort()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rows = rows[int(len(rows) * skip):]
span = rows[-1][1] - rows[0][0]
by = collections.defaultdict(lambda: [0, 0, 0])
for s, e, n in rows:
    by[n][0] += e - s; by[n][1] += 1; by[n][2] = max(by[n][2], e - s)
print('calls %d over %.2f ms' % (len(rows), span / 1e6))
for n, (t, c, mx) in sorted(by.items(), key=lambda kv: -kv[1][0])[:15]:
    print('%10.1f us  %6d x %-40s mean %.1f us  max %.1f us' % (t / 1e3, c, n, t / 1e3 / c, mx / 1e3))
print('longest calls:')
for s, e, n in sorted(rows, key=lambda r: r[0] - r[1])[:25]:
    print('%10.1f us  %-36s at +%.1f us' % ((e - s) / 1e3, n, (s - rows[0][0]) / 1e3))
