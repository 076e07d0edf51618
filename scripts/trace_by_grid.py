"""Average duration of a kernel per launch geometry in a rocprofv3 kernel trace:
python scripts/trace_by_grid.py <kernel_trace.csv> <substring of the kernel name> ..."""
import collections, csv, sys
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name']
    if any(k in n for k in sys.argv[2:]):
        key = (n[:70], r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
        agg[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    print('%-70s grid %s x %s x %s wg %s: %5d launches  avg %8.1f us  median %8.1f  min %8.1f  max %8.1f' %
          (k + (len(v), sum(v) / len(v) / 1e3, v[len(v) // 2] / 1e3, v[0] / 1e3, v[-1] / 1e3)))
