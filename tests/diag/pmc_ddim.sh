#!/bin/bash
# MFMA utilisation and LDS bank conflicts of the DDIM loop's kernels from PMC counters (two separate --pmc runs, no tracing):
#   bash tests/diag/pmc_ddim.sh [tag]     -> gpurun_out/<tag>_pmc_mfma.json
set -eo pipefail
TAG=${1:-r01}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_busy_$TAG $OUT/pmc_lds_$TAG
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_busy_$TAG -- python3 $ROOT/tests/diag/run_ddim.py 3 > $OUT/pmc_busy_$TAG.log 2>&1
echo "mfma pass done"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/pmc_lds_$TAG -- python3 $ROOT/tests/diag/run_ddim.py 3 > $OUT/pmc_lds_$TAG.log 2>&1
echo "lds pass done"
cd $ROOT
python3 profiles/summarize.py mfma $OUT/pmc_busy_$TAG $OUT/pmc_lds_$TAG $OUT/${TAG}_pmc_mfma.json
rm -rf $OUT/pmc_busy_$TAG $OUT/pmc_lds_$TAG
