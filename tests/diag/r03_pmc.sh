#!/bin/bash
# the trace summary + the two PMC passes of profiles/collect.sh alone (after its bench line and kernel trace exist under gpurun_out/)
set -eo pipefail
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
PMC_WORK="tests/diag/run_ar.py 24"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_$TAG -- python3 $ROOT/$PMC_WORK > $OUT/pmc_fetch_$TAG.log 2>&1
echo "pmc fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_$TAG -- python3 $ROOT/$PMC_WORK > $OUT/pmc_write_$TAG.log 2>&1
echo "pmc write done"
cd $ROOT
python3 profiles/summarize.py pmc $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG $OUT/${TAG}_pmc_traffic.json "tests/diag/run_ar.py 24 (configs[1] size, bf16, B=16: prefill + 23 KV-cached decode steps launched eagerly -- the bench's decode launches on a bounded token loop)"
rm -rf $OUT/pmc_fetch_$TAG $OUT/pmc_write_$TAG
