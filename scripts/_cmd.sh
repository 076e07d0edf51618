echo -n "default: "; SHAPE=scaleup BATCH=64 timeout -k 10 200 python3 scripts/train_only.py lstm 5 || exit 1
echo -n "no live rows: "; CVAE_LIVE_ROWS=0 SHAPE=scaleup BATCH=64 timeout -k 10 200 python3 scripts/train_only.py lstm 5 || exit 1
echo -n "no wide batch: "; CVAE_WIDE_BATCH=0 SHAPE=scaleup BATCH=64 timeout -k 10 200 python3 scripts/train_only.py lstm 5 || exit 1
