# Diagnostic: time the many-row recurrent step (scripts/rnn_large_micro.py) under rocprofv3 for the built library and
# for experimental variants (scripts/build_variant.sh):  bash scripts/rl_variants.sh base NOEPI ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset CVAE_HIP_LIB; else export CVAE_HIP_LIB=$R/coupled-vae_amd/lib/var/libcvae_hip_rl_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rlv_$v -o m -- python3 $R/scripts/rnn_large_micro.py lstm 500 8 > $R/gpurun_out/rlv_$v.log 2>&1 || exit 1
  echo "$v: $(grep rnn_large_step $R/gpurun_out/rlv_$v/m_kernel_stats.csv | cut -d, -f2-4)"
  rm -f $R/gpurun_out/rlv_$v/m_kernel_trace.csv
done
