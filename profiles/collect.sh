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
# PMC passes.  rocprofv3 --pmc cannot follow the whole bench on this image: it segfaults inside its dispatch interception on a captured
# graph's launches whoever replays the graph (profiles/r03_pmc_graph_probe.log) and, in round 3, also somewhere in the ~40 000 eager launches
# of `bench.py --no-graph` (gpurun_out/pmc_fetch_r03.log: fault inside the HIP launch call of an ordinary kernel).  The passes therefore run
# the SAME decode launches on a bounded token loop -- tests/diag/run_ar.py 24: prefill + 23 KV-cached steps at configs[1]'s size, launched eagerly
# -- which gives the per-launch HBM bytes the roofline line quotes (per-launch traffic of a GEMV does not depend on how many tokens follow).
PMC_WORK="tests/diag/run_ar.py 24"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $ROOT/$PMC_WORK > $OUT/pmc_fetch_$TAG.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $ROOT/$PMC_WORK > $OUT/pmc_write_$TAG.log 2>&1
echo "pmc write done"
# the kernel trace is tens of MB: summarise on the box, keep only the summaries
cd $ROOT
python3 profiles/summarize.py trace $OUT/prof_$TAG $OUT/${TAG}_bench
python3 profiles/summarize.py pmc $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/${TAG}_pmc_traffic.json "tests/diag/run_ar.py 24 (configs[1] size, bf16, B=16: prefill + 23 KV-cached decode steps launched eagerly -- the bench's decode launches on a bounded token loop)"
rm -rf $OUT/prof_$TAG/*/*_kernel_trace.csv $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
