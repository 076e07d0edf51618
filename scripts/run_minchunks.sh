# Diagnostic: training step time against the split-K floor of gemm_wide (chunks per work item).
set -e
R=${GRAFT_REPO_ROOT:-.}
for shp in ptb yelp; do
  for mc in ${MCS:-1 2 3 4 6}; do
    echo -n "$shp min_chunks=$mc: "
    SHAPE=$shp CVAE_WIDE_MIN_CHUNKS=$mc timeout -k 10 120 python3 $R/scripts/train_only.py lstm 40 | tail -n 1
  done
done
