#!/bin/bash
# rocprofv3 kernel trace of a few DDIM steps at configs[1] size, summarised per (kernel, shape):  bash tests/diag/prof_ddim.sh [tag]
set -eo pipefail
TAG=${1:-ddim}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$TAG -- python3 $ROOT/tests/diag/run_ddim.py 10 > $OUT/prof_$TAG.log 2>&1
cd $ROOT
python3 profiles/summarize.py trace $OUT/prof_$TAG $OUT/$TAG
rm -rf $OUT/prof_$TAG
