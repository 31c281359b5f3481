#!/bin/bash
# rocprofv3 kernel trace of the vocoder on a 1088-frame mel, summarised per (kernel, shape):  bash tests/diag/prof_voc.sh
set -eo pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out
python3 tests/diag/voc_time.py bf16 f32
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/prof_voc
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_voc -- python3 $ROOT/tests/diag/voc_time.py bf16 > $OUT/prof_voc.log 2>&1
cd $ROOT
python3 profiles/summarize.py trace $OUT/prof_voc $OUT/voc
rm -rf $OUT/prof_voc
