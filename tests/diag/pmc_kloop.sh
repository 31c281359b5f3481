#!/bin/bash
# What bounds the DDIM GEMMs' k-loop: SQ wave-time shares, LDS activity / conflicts / queue-full stalls, L2 hit rate and request
# latency, texture-path (TA / TCP) stalls -- per kernel of the DDIM layer chain.  Separate --pmc passes, no tracing, the binary directly after `--`
# (tests/diag/ddim_chain_plain.bin: the chain WITHOUT stamps, launched eagerly: ~10^3 dispatches, inside what the profiler survives on this image).
#   bash tests/diag/pmc_kloop.sh [tag]   -> gpurun_out/<tag>_pmc_kloop.txt
set -eo pipefail
TAG=${1:-r05}
ROOT=$(pwd); OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export DC_EAGER=1
BIN=$ROOT/tests/diag/ddim_chain_plain.bin
pass() { # name counters...
	local n=$1; shift
	rm -rf $OUT/pmc_kloop_$n
	if timeout -k 10 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc_kloop_$n -- $BIN 6 > $OUT/pmc_kloop_$n.log 2>&1; then echo "pass $n done"; else echo "pass $n FAILED (gpurun_out/pmc_kloop_$n.log)"; fi
}
pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE
pass vmem SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU
# (the TCC / TCP / TA blocks take few counters per pass: "exceeds the capabilities of the hardware" otherwise)
pass l2a TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE
pass l2b TCC_REQ_sum TCC_READ_sum
pass l2c TCC_BUSY_sum TCC_TAG_STALL_sum
pass tcpa TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum GRBM_GUI_ACTIVE
pass tcpb TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tcpc TCP_GATE_EN1_sum TCP_GATE_EN2_sum
pass ta TA_TA_BUSY_sum TA_BUFFER_READ_LDS_WAVEFRONTS_sum GRBM_GUI_ACTIVE
cd $ROOT
python3 - $OUT sq lds vmem l2a l2b l2c tcpa tcpb tcpc ta > $OUT/${TAG}_pmc_kloop.txt <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for name in sys.argv[2:]:
	fs = glob.glob(f"{out}/pmc_kloop_{name}/**/*_counter_collection.csv", recursive=True)
	if not fs:
		print(f"pass {name}: no counter file (see gpurun_out/pmc_kloop_{name}.log)"); continue
	for r in csv.DictReader(open(fs[0])):
		k = (r["Kernel_Name"][:64], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""))
		acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
	m = max(max(n[k].values()), 1)
	print(f"{k[0]}  grid {k[1]} wg {k[2]}  x{m} launches; per launch:")
	for name in sorted(c): print(f"    {name:36s} {c[name] / max(n[k][name], 1):16.0f}")
PY
grep -c . $OUT/${TAG}_pmc_kloop.txt
for n in sq lds vmem l2a l2b l2c tcpa tcpb tcpc ta; do rm -rf $OUT/pmc_kloop_$n; done
