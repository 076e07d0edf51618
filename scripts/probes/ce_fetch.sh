# FETCH_SIZE of the evaluation's CE launch under the environment given (diagnostic)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
NBATCH=2 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ce_fetch_$1 -o f -- python3 $R/scripts/eval_only.py lstm 500 > $R/gpurun_out/ce_fetch_$1.log 2>&1
python3 - $R/gpurun_out/ce_fetch_$1/f_counter_collection.csv $1 <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE': agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'ce_core' in k or 'gemm_nt_bf3_batched' in k: print(sys.argv[2], k, len(v), 'launches, fetch per launch (GB):', ' '.join('%.2f' % (x * 2048 / 1e9) for x in v))
PY
rm -f $R/gpurun_out/ce_fetch_$1/f_counter_collection.csv
