#!/bin/bash
# the two PMC passes of profiles/collect.sh alone (after a trace already exists)
set -eo pipefail
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-roofline --no-cpu-baseline > $OUT/pmc_fetch_$TAG.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-roofline --no-cpu-baseline > $OUT/pmc_write_$TAG.log 2>&1
echo "pmc write done"
cd $ROOT
python3 profiles/summarize.py pmc $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/${TAG}_pmc_traffic.json "bench.py --steps 1 --warmup 0 --no-graph (configs[1], bf16: one whole utterance, token loop launched eagerly)"
rm -rf $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
