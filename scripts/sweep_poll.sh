# sweep of the in-band hand-off's work split and first-poll delay (training step, Yelp shape)
for sp in ${SPLITS:-2 4}; do
for d in ${DELAYS:-"8 2" "16 4" "24 6" "32 8"}; do
  set -- $d
  echo "split $sp delay $1 cell $2: $(CVAE_PERSIST_FWD_SPLIT=$sp CVAE_PERSIST_POLL_DELAY=$1 CVAE_PERSIST_POLL_DELAY_CELL=$2 timeout -k 10 100 python scripts/train_only.py lstm 20 2>&1 | tail -1)"
done
done
