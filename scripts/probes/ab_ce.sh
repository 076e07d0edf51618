cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in dma regs dma regs; do
  if [ $l = regs ]; then export CVAE_HIP_LIB=$R/coupled-vae_amd/lib/var/libcvae_hip_ce_REGS.so; else unset CVAE_HIP_LIB; fi
  echo "$l: $(NBATCH=8 timeout -k 10 200 python3 $R/scripts/eval_only.py lstm 500 | tail -1)"
done
for l in dma regs; do
  if [ $l = regs ]; then export CVAE_HIP_LIB=$R/coupled-vae_amd/lib/var/libcvae_hip_ce_REGS.so; else unset CVAE_HIP_LIB; fi
  NBATCH=2 timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/ab_ce_$l -o f -- python3 $R/scripts/eval_only.py lstm 500 > $R/gpurun_out/ab_ce_$l.log 2>&1
  python3 - $R/gpurun_out/ab_ce_$l/f_counter_collection.csv $l <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if r['Counter_Name'] == 'FETCH_SIZE': agg[r['Kernel_Name'][:60]].append(float(r['Counter_Value']))
for k, v in agg.items():
    if 'ce_core' in k or 'wreg' in k: print(sys.argv[2], k, len(v), 'launches, avg fetch (KB x 2 x 1024 -> bytes) %.2f GB' % (sum(v) / len(v) * 2048 / 1e9))
PY
  rm -f $R/gpurun_out/ab_ce_$l/f_counter_collection.csv
done
