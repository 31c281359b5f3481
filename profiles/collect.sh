#!/bin/bash
# Collects everything profiles/ holds for one round on a GPU box:  bash profiles/collect.sh r01
# (run through gpurun from the repo root; outputs land in gpurun_out/ and are summarised by profiles/summarize.py afterwards)
set -eo pipefail
TAG=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python3 bench.py > $OUT/bench_$TAG.json 2> $OUT/bench_$TAG.log
tail -1 $OUT/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline > $OUT/prof_$TAG.log 2>&1
echo "trace done"
# PMC passes on the bench workload itself, one utterance of configs[1] with the token loop launched eagerly (--no-graph: rocprofv3 --pmc
# crashes on a captured graph's dispatches; the kernels and their bytes are the same)
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-roofline --no-cpu-baseline > $OUT/pmc_fetch_$TAG.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-graph --no-roofline --no-cpu-baseline > $OUT/pmc_write_$TAG.log 2>&1
echo "pmc write done"
# the kernel trace is tens of MB: summarise on the box, keep only the summaries
cd $ROOT
python3 profiles/summarize.py trace $OUT/prof_$TAG $OUT/${TAG}_bench
python3 profiles/summarize.py pmc $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/${TAG}_pmc_traffic.json "bench.py --steps 1 --warmup 0 --no-graph (configs[1], bf16: one whole utterance, token loop launched eagerly)"
rm -rf $OUT/prof_$TAG/*/*_kernel_trace.csv $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
