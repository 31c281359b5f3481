"""Turn rocprofv3 output directories into the small summaries committed under profiles/.

  kernel trace (rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python3 bench.py ...):
      python profiles/summarize.py trace DIR profiles/rNN_bench
          -> rNN_bench_kernel_stats.csv  (top 40 rows of rocprofv3's own *_kernel_stats.csv)
          -> rNN_bench_per_shape.csv     (the trace grouped by kernel, grid, workgroup: calls, total ms, median/min/p90 us)
  PMC passes (two separate runs, rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --output-format csv -d DIR -- python3 bench.py --steps 1 ...):
      python profiles/summarize.py pmc FETCH_DIR WRITE_DIR profiles/rNN_pmc_traffic.json ["workload text"]
          -> per (kernel, grid) average FETCH_SIZE / WRITE_SIZE per launch, read side corrected as MI355X_MICROARCH.md prescribes for
             gfx950 (FETCH_SIZE counts KB and under-reports 16-byte-per-lane coalesced streams by 2x), plus the launch-weighted
             k_skinny average bench.py reports as roofline.traffic.
  MFMA / LDS passes (bash tests/diag/pmc_ddim.sh):
      python profiles/summarize.py mfma BUSY_DIR LDS_DIR profiles/rNN_pmc_mfma.json
"""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict


def _one(d, pattern):
	files = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
	if not files:
		raise SystemExit(f"no {pattern} under {d}")
	return files[-1]


def short(name):
	return name if len(name) <= 120 else name[:117] + "..."


def trace(d, out_prefix):
	stats = list(csv.reader(open(_one(d, "*_kernel_stats.csv"))))
	with open(out_prefix + "_kernel_stats.csv", "w", newline="") as f:
		csv.writer(f).writerows(stats[:41])
	groups = defaultdict(list)
	vg = {}
	for r in csv.DictReader(open(_one(d, "*_kernel_trace.csv"))):
		key = (r["Kernel_Name"], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
		groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
		vg[key] = r["VGPR_Count"]
	rows = []
	for key, us in groups.items():
		us.sort()
		rows.append([short(key[0]), *key[1:], vg[key], len(us), f"{sum(us) / 1e3:.2f}", f"{statistics.median(us):.2f}", f"{us[0]:.2f}",
					 f"{us[min(len(us) - 1, int(0.9 * len(us)))]:.2f}"])
	rows.sort(key=lambda r: -float(r[7]))
	with open(out_prefix + "_per_shape.csv", "w", newline="") as f:
		w = csv.writer(f)
		w.writerow(["kernel", "grid_x", "grid_y", "grid_z", "wg", "vgpr", "calls", "total_ms", "median_us", "min_us", "p90_us"])
		w.writerows(rows[:60])
	total = sum(float(r[7]) for r in rows)
	print(f"{len(rows)} (kernel, shape) groups, {total:.1f} ms of kernel time; top 5:")
	for r in rows[:5]:
		print("  ", r[0][:70], r[1], "calls", r[6], "total_ms", r[7], "median_us", r[8])


def _counter(d, name):
	acc = defaultdict(list)
	for r in csv.DictReader(open(_one(d, "*_counter_collection.csv"))):
		if r["Counter_Name"] != name:
			continue
		acc[(r["Kernel_Name"], r["Grid_Size"], r["Workgroup_Size"])].append(float(r["Counter_Value"]))
	return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def pmc(fetch_dir, write_dir, out, workload="tests/diag/run_ar.py 12 (bf16, B=16, prefill + 11 decode tokens)"):
	fetch, calls = _counter(fetch_dir, "FETCH_SIZE")
	write, _ = _counter(write_dir, "WRITE_SIZE")
	per = []
	sk_bytes = sk_calls = 0
	for k in sorted(fetch, key=lambda k: -fetch[k] * calls[k]):
		if not k[0].startswith("void ttk::") and "ttk" not in k[0]:
			continue
		rd = fetch[k] * 1024 * 2          # KB -> bytes, x2 gfx950 correction for wide coalesced reads
		wr = write.get(k, 0.0) * 1024
		per.append({"kernel": short(k[0]), "grid": k[1], "wg": k[2], "launches": calls[k], "FETCH_SIZE_KB_raw": round(fetch[k], 1),
					"hbm_read_bytes_corrected": int(rd), "WRITE_SIZE_bytes": int(wr)})
		if "k_skinny" in k[0] or "k_gemv" in k[0]:      # the decode GEMVs: k_gemv at the benchmarked geometry (csrc/gemv.hip), k_skinny elsewhere
			sk_bytes += (rd + wr) * calls[k]
			sk_calls += calls[k]
	res = {"how": f"rocprofv3 --pmc FETCH_SIZE and, in a separate run, --pmc WRITE_SIZE over {workload}; counter units KB; read side doubled per "
				  "the guide's gfx950 correction for 16-byte-per-lane coalesced streams",
		   "workload": workload,
		   # (key name kept from round 1: bench.py reads it; since round 3 the launches behind it are ttk::k_gemv)
		   "k_skinny_avg_hbm_bytes_per_launch": int(sk_bytes / max(sk_calls, 1)), "k_skinny_launches": sk_calls, "per_kernel": per[:24]}
	json.dump(res, open(out, "w"), indent=1)
	print("decode GEMV (k_gemv / k_skinny) avg HBM bytes / launch:", res["k_skinny_avg_hbm_bytes_per_launch"], "over", sk_calls, "launches")


def mfma(busy_dir, lds_dir, out):
	"""MFMA utilisation and LDS bank-conflict share per kernel of the DDIM loop: SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x
	256 CUs x 4 SIMDs) and SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (separate --pmc runs)."""
	busy, calls = _counter(busy_dir, "SQ_VALU_MFMA_BUSY_CYCLES")
	active, _ = _counter(busy_dir, "GRBM_GUI_ACTIVE")
	conf, _ = _counter(lds_dir, "SQ_LDS_BANK_CONFLICT")
	idx, _ = _counter(lds_dir, "SQ_LDS_IDX_ACTIVE")
	per = []
	for k in sorted(busy, key=lambda k: -active.get(k, 0) * calls[k]):
		if "ttk" not in k[0] or not active.get(k):
			continue
		per.append({"kernel": short(k[0]), "grid": k[1], "wg": k[2], "launches": calls[k],
					"mfma_busy_cycles": int(busy[k]), "gui_active_cycles_per_xcd": int(active[k] / 8),
					"mfma_util": round(busy[k] / (active[k] / 8 * 256 * 4), 4),
					"lds_bank_conflict_share": round(conf.get(k, 0.0) / idx[k], 4) if idx.get(k) else None})
	res = {"how": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE and, in a separate run, --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE over "
				  "tests/diag/run_ddim.py 3 (bf16, T=1088, cond-free batch of 2); mfma_util = busy / (GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs)",
		   "per_kernel": per[:16]}
	json.dump(res, open(out, "w"), indent=1)
	for r in per[:8]:
		print(r["kernel"][:60], r["grid"], "launches", r["launches"], "mfma_util", r["mfma_util"], "lds_conflict", r["lds_bank_conflict_share"])


if __name__ == "__main__":
	if len(sys.argv) == 5 and sys.argv[1] == "mfma":
		mfma(sys.argv[2], sys.argv[3], sys.argv[4])
	elif len(sys.argv) == 4 and sys.argv[1] == "trace":
		trace(sys.argv[2], sys.argv[3])
	elif len(sys.argv) in (5, 6) and sys.argv[1] == "pmc":
		pmc(*sys.argv[2:])
	else:
		raise SystemExit(__doc__)
