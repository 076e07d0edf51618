# kernel-trace of the wreg probe; prints the per-kernel stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wreg_prof -o w -- python3 $R/scripts/probes/wreg_check.py ${1:-512} ${2:-500} > $R/gpurun_out/wreg_prof.log 2>&1
f=$(ls $R/gpurun_out/wreg_prof/*/*kernel_stats.csv $R/gpurun_out/wreg_prof/*kernel_stats.csv 2>/dev/null | head -n 1 || true)
head -n 12 $f | cut -c1-160
