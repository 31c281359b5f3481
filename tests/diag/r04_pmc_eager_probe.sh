#!/bin/bash
# ONE counter pass of rocprofv3 over the whole eagerly launched bench (round 3's fault in this configuration left no tracked log).
# Whatever happens, the complete output is kept: gpurun_out/r04_pmc_eager_probe.log -> profiles/.  Not a loop: the pass runs once.
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/pmc_eager_r04
{
	echo "=== rocprofv3 --pmc FETCH_SIZE -- python3 -X faulthandler bench.py --no-graph --steps 1 --warmup 0 --no-roofline --no-cpu-baseline"
	date -u
	timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_eager_r04 -- python3 -X faulthandler $ROOT/bench.py --no-graph --steps 1 --warmup 0 --no-roofline --no-cpu-baseline
	echo "exit code $?"
	date -u
	echo "=== counter records written:"
	find $OUT/pmc_eager_r04 -name "*counter_collection.csv" -exec wc -l {} \;
} > $OUT/r04_pmc_eager_probe.log 2>&1
cd $ROOT
f=$(find $OUT/pmc_eager_r04 -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then
	python3 - "$f" >> $OUT/r04_pmc_eager_probe.log 2>&1 <<'PY'
import csv, sys, collections
n = collections.Counter(); v = collections.Counter()
with open(sys.argv[1]) as fh:
	for row in csv.DictReader(fh):
		k = row.get("Kernel_Name", "?").split("(")[0][:60]
		n[k] += 1; v[k] += float(row.get("Counter_Value", 0) or 0)
print("=== dispatches with a FETCH_SIZE record, by kernel (count, mean raw counter):")
for k, c in n.most_common(12): print(f"{c:8d}  {v[k] / c:14.1f}  {k}")
print("total dispatches recorded:", sum(n.values()))
PY
fi
rm -rf $OUT/pmc_eager_r04
tail -40 $OUT/r04_pmc_eager_probe.log
