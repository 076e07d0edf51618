# kernel-trace stats of training steps only, per shape: gpurun_out/kst_<shape>.txt (scripts/kstats.py over 25 steps)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for shp in ${SHAPES:-ptb yelp}; do
  SHAPE=$shp timeout -k 10 170 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kst_$shp -o t -- python3 $R/scripts/train_only.py lstm 20 > $R/gpurun_out/kst_$shp.log 2>&1
  f=$(ls $R/gpurun_out/kst_$shp/*/*kernel_stats.csv $R/gpurun_out/kst_$shp/*kernel_stats.csv 2>/dev/null | head -n 1 || true)
  python3 $R/scripts/kstats.py $f 25 40 > $R/gpurun_out/kst_$shp.txt
  tail -n 1 $R/gpurun_out/kst_$shp.log
  rm -rf $R/gpurun_out/kst_$shp
done
