cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for l in new old; do
  if [ $l = old ]; then export CVAE_HIP_LIB=$R/coupled-vae_amd/lib/var/libcvae_hip_ce_old.so; else unset CVAE_HIP_LIB; fi
  MODEL=cvae-couple SHAPE=switchboard timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_swb_$l -o s -- python3 $R/scripts/train_only.py lstm 30 > $R/gpurun_out/ab_swb_$l.log 2>&1
  echo "== $l"; python3 - $R/gpurun_out/ab_swb_$l/s_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print('%6.2f%%  %9.1f us  %5d calls  avg %7.1f us  %s' % (float(r['Percentage']), float(r['TotalDurationNs']) / 1e3, int(r['Calls']), float(r['AverageNs']) / 1e3, r['Name'][:60]))
PY
  rm -f $R/gpurun_out/ab_swb_$l/s_kernel_trace.csv
done
