#!/bin/bash
# Diagnostic: which side faults when `rocprofv3 --pmc` meets the captured token step?  ONE counter pass per form,
# never repeated: (a) the step replayed with hipGraphLaunch on torch's exec handle (ttk_graph_launch, the default), (b) the same graph replayed
# by torch.cuda.CUDAGraph.replay() (TTK_AR_RAW_REPLAY=0).  PYTHONFAULTHANDLER prints the Python frame of a SIGSEGV.
#   bash tests/diag/pmc_graph_probe.sh [tag]   -> gpurun_out/<tag>_pmc_graph_probe.log
TAG=${1:-r03}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
LOG=$OUT/${TAG}_pmc_graph_probe.log
: > $LOG
cd /tmp && export TMPDIR=/tmp PYTHONFAULTHANDLER=1
for raw in 1 0; do
	rm -rf $OUT/pmc_probe_$raw
	echo "=== TTK_AR_RAW_REPLAY=$raw: rocprofv3 --pmc FETCH_SIZE -- python3 bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline (captured token step)" >> $LOG
	TTK_AR_RAW_REPLAY=$raw timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_probe_$raw -- python3 $ROOT/bench.py --steps 1 --warmup 0 --no-roofline --no-cpu-baseline > $OUT/pmc_probe_$raw.log 2>&1
	rc=$?
	echo "exit code $rc" >> $LOG
	grep -v "amdgpu.ids" $OUT/pmc_probe_$raw.log | tail -40 >> $LOG
	ls $OUT/pmc_probe_$raw/*/ 2>/dev/null | head -5 >> $LOG
	rm -rf $OUT/pmc_probe_$raw
done
cat $LOG | tail -100
