#!/bin/bash
# SQ wave-time breakdown (parked / issue-stalled / issuing, VALU and LDS shares) of the DDIM loop's kernels: one --pmc pass, no tracing
#   bash tests/diag/pmc_sq.sh [tag]   -> gpurun_out/<tag>_pmc_sq.txt
set -eo pipefail
TAG=${1:-r02}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_sq_$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_WAIT_INST_LDS --output-format csv -d $OUT/pmc_sq_$TAG -- python3 $ROOT/tests/diag/run_ddim.py 3 > $OUT/pmc_sq_$TAG.log 2>&1
echo "sq pass done"
cd $ROOT
python3 - $OUT/pmc_sq_$TAG > $OUT/${TAG}_pmc_sq.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
	k = (r["Kernel_Name"][:70], r.get("Grid_Size", ""))
	acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
	if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1]["SQ_WAVE_CYCLES"])[:12]:
	w = c["SQ_WAVE_CYCLES"] or 1
	print(f"{k[0]:72s} grid {k[1]:>8s} x{n[k]:4d}  wave_cycles/launch {w / max(n[k], 1):12.0f}  parked {c['SQ_WAIT_ANY'] / w:5.2f}  issue-stall {c['SQ_WAIT_INST_ANY'] / w:5.2f} (lds {c['SQ_WAIT_INST_LDS'] / w:5.2f})  issuing {c['SQ_ACTIVE_INST_ANY'] / w:5.2f} (valu {c['SQ_ACTIVE_INST_VALU'] / w:5.2f} lds {c['SQ_ACTIVE_INST_LDS'] / w:5.2f})  valu insts/launch {c['SQ_INSTS_VALU'] / max(n[k], 1):12.0f}")
PY
cat $OUT/${TAG}_pmc_sq.txt
rm -rf $OUT/pmc_sq_$TAG
