#!/bin/bash
# GPU box: kernel-trace statistics and the two HBM counter passes of the default bench.
# usage (through gpurun): bash scripts/collect_profiles.sh <tag>     -> gpurun_out/prof_<tag>/
set -e -o pipefail
TAG=${1:-run}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o bench -- python3 $ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/kt.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --iw-batches 2 > $OUT/pmc_fetch.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --no-cpu-baseline --steps 3 --warmup 1 --iw-batches 2 > $OUT/pmc_write.log 2>&1
python3 $ROOT/scripts/pmc_summary.py $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/pmc_hbm_summary.json
rm -f $OUT/pmc_fetch/f_counter_collection.csv $OUT/pmc_write/w_counter_collection.csv $OUT/kt/bench_kernel_trace.csv
tail -c 1500 $OUT/bench.json
