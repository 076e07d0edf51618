# kernel-trace stats of the IW-NLL evaluation alone (Yelp shape, K = 500): where a batch's time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NBATCH=${NBATCH:-4} timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eval -o e -- python3 $R/scripts/eval_only.py lstm 500 > $R/gpurun_out/prof_eval.log 2>&1
tail -n 1 $R/gpurun_out/prof_eval.log
python3 - $R/gpurun_out/prof_eval/e_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:16]:
    print('%6.2f%%  %9.1f us total  %6d calls  avg %8.1f us  %s' % (float(r['Percentage']), float(r['TotalDurationNs']) / 1e3, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:70]))
print('total %.1f ms' % (tot / 1e6))
PY
rm -f $R/gpurun_out/prof_eval/e_kernel_trace.csv
