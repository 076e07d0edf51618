"""Diagnostic: per-kernel averages of a rocprofv3 --pmc counter_collection.csv.
usage: python scripts/pmc_kernel.py <counter_collection.csv> <kernel-name substring>"""
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r['Kernel_Name']:
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(acc):
    print('%-32s %16.0f  (mean of %d dispatches)' % (k, acc[k] / n[k], n[k]))
