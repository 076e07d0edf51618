#!/bin/bash
# GPU box: the round-3 measurement record.  usage (through gpurun): bash scripts/collect_profiles_r03.sh <tag>
#   bench.json            the default bench line
#   kt_yelp/              rocprofv3 --kernel-trace --stats of the Yelp legs alone (--no-ptb --no-fp32): every launch of an
#                         instantiation in that CSV belongs to one shape, so roofline.avg_launch_ms can be read off it
#   kt_full/              the same for the default command
#   pmc_hbm_summary.json  FETCH_SIZE / WRITE_SIZE passes;  pmc_mfma_summary.json  SQ MFMA-busy pass
set -e -o pipefail
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LEAN="--no-cpu-baseline --no-ptb --no-fp32"
timeout -k 10 400 python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_yelp -o bench -- python3 $ROOT/bench.py $LEAN --steps 20 --warmup 5 > $OUT/kt_yelp.log 2>&1
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt_full -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/kt_full.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py $LEAN --steps 3 --warmup 1 --iw-batches 2 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py $LEAN --steps 3 --warmup 1 --iw-batches 2 > $OUT/pmc_write.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -o s -- python3 $ROOT/bench.py $LEAN --steps 3 --warmup 1 --iw-batches 2 > $OUT/pmc_sq.log 2>&1
F=$(ls $OUT/pmc_fetch/*/f_counter_collection.csv $OUT/pmc_fetch/f_counter_collection.csv 2>/dev/null | head -n 1 || true)
W=$(ls $OUT/pmc_write/*/w_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv 2>/dev/null | head -n 1 || true)
S=$(ls $OUT/pmc_sq/*/s_counter_collection.csv $OUT/pmc_sq/s_counter_collection.csv 2>/dev/null | head -n 1 || true)
python3 $ROOT/scripts/pmc_summary_r03.py hbm $F $W $OUT/pmc_hbm_summary.json
python3 $ROOT/scripts/pmc_summary_r03.py mfma $S $OUT/pmc_mfma_summary.json
rm -f $F $W $S
find $OUT -name '*kernel_trace.csv' -delete
tail -c 600 $OUT/bench.json
