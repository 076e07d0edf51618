# sweep of the in-band backward recurrence: work split and first-poll delay (training step, Yelp shape)
echo "flag bwd: $(CVAE_PERSIST_TAG_BWD=0 timeout -k 10 100 python scripts/train_only.py lstm 20 2>&1 | tail -1)"
for sp in ${SPLITS:-1 2 4}; do
for d in ${DELAYS:-0 8 16}; do
  echo "bwd split $sp delay $d: $(CVAE_PERSIST_BWD_SPLIT=$sp CVAE_PERSIST_POLL_DELAY_BWD=$d timeout -k 10 100 python scripts/train_only.py lstm 20 2>&1 | tail -1)"
done
done
