# A/B of the whole-sequence forward kernel's hand-off on the launch-bound shapes
for shp in "cvae-couple switchboard" "vae-couple ptb"; do
  set -- $shp
  for v in "CVAE_PERSIST_TAG=0" "CVAE_PERSIST_POLL_DELAY=8 CVAE_PERSIST_POLL_DELAY_CELL=2" "CVAE_PERSIST_POLL_DELAY=16 CVAE_PERSIST_POLL_DELAY_CELL=4" "CVAE_PERSIST_FWD_SPLIT=2" "CVAE_PERSIST_FWD_SPLIT=1"; do
    echo "$2 $v: $(env MODEL=$1 SHAPE=$2 $v timeout -k 10 100 python scripts/train_only.py lstm 30 2>&1 | tail -1)"
  done
done
