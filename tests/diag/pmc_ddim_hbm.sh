#!/bin/bash
set -eo pipefail
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DC_EAGER=1
BIN=$ROOT/tests/diag/ddim_chain_plain.bin
for n in FETCH_SIZE WRITE_SIZE; do
	rm -rf $OUT/pmc_hbm_$n
	timeout -k 10 240 rocprofv3 --pmc $n --output-format csv -d $OUT/pmc_hbm_$n -- $BIN 6 > $OUT/pmc_hbm_$n.log 2>&1 && echo "pass $n done" || echo "pass $n FAILED"
done
cd $ROOT
python3 - $OUT > $OUT/r05_pmc_ddim_hbm.txt <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(collections.Counter)
for name in ("FETCH_SIZE","WRITE_SIZE"):
	fs=glob.glob(f"{out}/pmc_hbm_{name}/**/*_counter_collection.csv", recursive=True)
	if not fs: print("no file for", name); continue
	for r in csv.DictReader(open(fs[0])):
		k=(r["Kernel_Name"][:64], r.get("Grid_Size",""))
		acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); n[k][r["Counter_Name"]]+=1
print("HBM traffic per launch of the DDIM layer chain's kernels (tests/diag/ddim_chain_plain.bin, eager; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; the guide's gfx950 correction doubles FETCH_SIZE)")
for k,c in sorted(acc.items(), key=lambda kv:-kv[1].get("FETCH_SIZE",0)):
	f=c.get("FETCH_SIZE",0)/max(n[k]["FETCH_SIZE"],1); w=c.get("WRITE_SIZE",0)/max(n[k]["WRITE_SIZE"],1)
	print(f"{k[0]:66s} grid {k[1]:>8s} x{max(n[k].values()):4d}  FETCH_SIZE {f:10.1f} KB (x2 = {2*f/1024:7.2f} MB)  WRITE_SIZE {w:10.1f} KB ({w/1024:6.2f} MB)")
PY
cat $OUT/r05_pmc_ddim_hbm.txt
rm -rf $OUT/pmc_hbm_FETCH_SIZE $OUT/pmc_hbm_WRITE_SIZE
