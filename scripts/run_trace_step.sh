set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shp in ptb yelp; do
  SHAPE=$shp timeout -k 10 170 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tr_$shp -o t -- python3 $R/scripts/train_only.py lstm 6 > $R/gpurun_out/tr_$shp.log 2>&1
  f=$(ls $R/gpurun_out/tr_$shp/*/*kernel_trace.csv $R/gpurun_out/tr_$shp/*kernel_trace.csv 2>/dev/null | head -n 1 || true)
  python3 $R/scripts/trace_step.py $f > $R/gpurun_out/step_$shp.txt
  tail -n 2 $R/gpurun_out/tr_$shp.log
  rm -rf $R/gpurun_out/tr_$shp
done
