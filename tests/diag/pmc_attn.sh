#!/bin/bash
# SQ counters of the DDIM loop's attention launch (VALU / MFMA instruction counts, busy and co-execution cycles, LDS): one --pmc pass per library, no tracing
#   bash tests/diag/pmc_attn.sh [tag] [lib ...]   -> gpurun_out/<tag>_pmc_attn.txt        (libs: paths for TTK_LIB, default the in-tree libttk.so)
set -eo pipefail
TAG=${1:-r03}; shift || true
LIBS=${@:-tortoise_tts_amd/libttk.so}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
: > $OUT/${TAG}_pmc_attn.txt
for LIB in $LIBS; do
	export TTK_LIB=$ROOT/$LIB
	rm -rf $OUT/pmc_attn_a $OUT/pmc_attn_b
	rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $OUT/pmc_attn_a -- python3 $ROOT/tests/diag/run_ddim.py 2 > $OUT/pmc_attn_a.log 2>&1
	rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_attn_b -- python3 $ROOT/tests/diag/run_ddim.py 2 > $OUT/pmc_attn_b.log 2>&1
	echo "$LIB passes done"
	python3 - $OUT/pmc_attn_a $OUT/pmc_attn_b "$LIB" >> $OUT/${TAG}_pmc_attn.txt <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for d in sys.argv[1:3]:
	f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
	seen = collections.Counter()
	for r in csv.DictReader(open(f)):
		if "k_attn_fwd" not in r["Kernel_Name"]: continue
		k = (r["Kernel_Name"][:60], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
		acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
		seen[(k, r["Counter_Name"])] += 1
	for (k, c), v in seen.items(): n[k] = max(n[k], v)
print(sys.argv[3])
for k, c in acc.items():
	m = max(n[k], 1)
	print(f"  {k[0]} grid {k[1]} wg {k[2]} x{m}")
	for name in sorted(c): print(f"    {name:32s} {c[name] / m:16.0f} per launch")
PY
done
cd $ROOT
cat $OUT/${TAG}_pmc_attn.txt
rm -rf $OUT/pmc_attn_a $OUT/pmc_attn_b
