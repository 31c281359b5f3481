#!/bin/bash
# Diagnostic: does `rocprofv3 --pmc` survive the eager decode step on the specialised GEMV launches (TTK_AR_LEAN=1) and on k_skinny (=0)?  One pass each.
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_pmc_gemv_probe.log
: > $LOG
cd /tmp && export TMPDIR=/tmp PYTHONFAULTHANDLER=1
for lean in 1 0; do
	rm -rf $OUT/pmc_gprobe_$lean
	echo "=== TTK_AR_LEAN=$lean: rocprofv3 --pmc FETCH_SIZE -- python3 tests/diag/run_ar.py 6" >> $LOG
	TTK_AR_LEAN=$lean timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_gprobe_$lean -- python3 $ROOT/tests/diag/run_ar.py 6 > $OUT/pmc_gprobe_$lean.log 2>&1
	echo "exit code $?" >> $LOG
	grep -v "amdgpu.ids" $OUT/pmc_gprobe_$lean.log | grep -v "^    @ .*unknown" | tail -25 >> $LOG
	rm -rf $OUT/pmc_gprobe_$lean
done
cat $LOG
