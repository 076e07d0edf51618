# Diagnostic: time the weights-in-registers many-row step (scripts/rnn_large_micro.py) under rocprofv3 for the built
# library and for experimental variants (scripts/build_variant.sh wr_NAME rnn_large.hip -DWR_EXP_NAME=1):
#   bash scripts/wr_variants.sh base NOMFMA ...
cd /tmp && export TMPDIR=/tmp
export CVAE_ALLOW_TIMING_BUILD=1
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset CVAE_HIP_LIB; else export CVAE_HIP_LIB=$R/coupled-vae_amd/lib/var/libcvae_hip_wr_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wrv_$v -o m -- python3 $R/scripts/rnn_large_micro.py lstm 500 8 > $R/gpurun_out/wrv_$v.log 2>&1 || exit 1
  echo "$v: $(python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'rnn_large_' in r['Name']: print(r['Name'][:40], r['Calls'], 'avg %.1f us' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3))
" $R/gpurun_out/wrv_$v/m_kernel_stats.csv)"
  rm -f $R/gpurun_out/wrv_$v/m_kernel_trace.csv
done
