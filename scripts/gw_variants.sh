# Diagnostic: gemm_wide time per training step (scripts/train_only.py under rocprofv3) for the built library and for
# experimental variants (scripts/build_variant.sh gw_<NAME> gemm_wide.hip ...):  bash scripts/gw_variants.sh base NOSPLIT ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  if [ $v = base ]; then unset CVAE_HIP_LIB; else export CVAE_HIP_LIB=$R/coupled-vae_amd/lib/var/libcvae_hip_gw_$v.so; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/gwv_$v -o m -- python3 $R/scripts/train_only.py lstm 10 > $R/gpurun_out/gwv_$v.log 2>&1 || exit 1
  echo "== $v: $(grep '^lstm' $R/gpurun_out/gwv_$v.log)"
  grep "gemm_wide_bf3" $R/gpurun_out/gwv_$v/m_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
  rm -f $R/gpurun_out/gwv_$v/m_kernel_trace.csv
done
