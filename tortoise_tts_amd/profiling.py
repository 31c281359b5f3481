"""Roofline bookkeeping for bench.py: per-kernel-kind HIP-event timing collected inside libttk (ttk_prof_begin / ttk_prof_end)
over one instrumented pass of the benchmark step, reduced to the `roofline` object of the dominant kernel.

Peaks (MI355X_MICROARCH.md, chip-level parameters): HBM3E 8.0 TB/s spec; dense MFMA 2.5 PFLOP/s bf16 / f16, 5 PFLOP/s fp8, 157.3 TFLOP/s f32.
A diffusion handle in the `fp8` mode runs its block GEMMs on the fp8 MFMA, so its dense-GEMM kind is graded against the fp8 peak (the
non-block convs of the same kind run bf16: the stricter denominator is used for all of them); `fp8w` is fp8 STORAGE on the bf16 MFMA.
`achieved` = algorithmic work of the kind (flop or bytes, summed over its launches) / summed launch duration, i.e. the
duration-weighted average over the launches of that kernel in one benchmark step.
"""
from __future__ import annotations

from . import _lib

KINDS = ["gemm", "skinny_gemm", "attn_fwd", "attn_decode", "groupnorm_stats", "groupnorm_apply", "layernorm"]
# "skinny_gemm" = the weight-streaming decode GEMVs: at the benchmarked geometry all of them are ttk::k_gemv instantiations (csrc/gemv.hip); the one
# ttk::k_skinny launch per utterance left in the tally is the prefill's mel head
KERNEL_NAMES = {"gemm": "ttk::k_gemm", "skinny_gemm": "ttk::k_gemv", "attn_fwd": "ttk::k_attn_fwd", "attn_decode": "ttk::k_attn_decode",
				"groupnorm_stats": "ttk::k_gn_stats", "groupnorm_apply": "ttk::k_gn_apply", "layernorm": "ttk::k_layernorm"}
MFMA_BOUND = {"gemm", "attn_fwd"}
PEAK_TFLOPS = {_lib.TTK_BF16: 2500.0, _lib.TTK_F16: 2500.0, _lib.TTK_F32: 157.3, _lib.TTK_FP8W: 2500.0, _lib.TTK_FP8: 5000.0}   # fp8w: fp8 weight storage, bf16 MFMA; fp8: fp8 MFMA
PEAK_HBM_GBS = 8000.0


def collect(step_fn, ar=None):
	"""Run step_fn once with per-launch timing on; returns {kind: dict(ms, launches, work)}."""
	import torch
	lib = _lib.load()
	saved = None
	if ar is not None:
		saved, ar.use_graph = ar.use_graph, False      # events cannot be recorded inside a captured graph
	torch.cuda.synchronize()
	_lib.check(lib.ttk_prof_begin(), "ttk_prof_begin")
	try:
		step_fn()
	finally:
		res = (_lib.ProfResult * len(KINDS))()
		rc = lib.ttk_prof_end(res, len(KINDS))
		if ar is not None:
			ar.use_graph = saved
	_lib.check(rc, "ttk_prof_end")
	return {k: dict(ms=res[i].ms, launches=int(res[i].launches), work=res[i].work) for i, k in enumerate(KINDS)}


def _pmc_traffic(kind):
	"""HBM bytes per launch of the dominant kernel from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_traffic.json,
	written by profiles/collect.sh + summarize.py).  PMC counters cannot be read from inside a running benchmark, so this is NOT a
	measurement of this run: it is returned together with the file and the workload it was taken on, and the bench line labels it
	`traffic_from_profiles`."""
	import glob
	import json
	import os
	files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r*_pmc_traffic.json")))
	if not files or kind != "skinny_gemm":
		return None, None
	d = json.load(open(files[-1]))
	return d.get("k_skinny_avg_hbm_bytes_per_launch"), {"file": "profiles/" + os.path.basename(files[-1]), "workload": d.get("workload", "tests/diag/run_ar.py 12 (prefill + 11 decode steps, bf16, B=16)")}


def dominant_kernel_roofline(step_fn, ar, df):
	table = collect(step_fn, ar)
	kind = max(table, key=lambda k: table[k]["ms"])
	r = table[kind]
	sec = r["ms"] * 1e-3
	breakdown = {k: {"ms": round(v["ms"], 3), "launches": v["launches"]} for k, v in table.items()}
	if kind in MFMA_BOUND:
		# attention (QK^T, PV) runs the bf16 MFMA in every 16-bit / fp8 mode; only the dense-GEMM kind of an fp8 handle has fp8 launches
		peak_t = PEAK_TFLOPS[df.dtype] if kind == "gemm" else PEAK_TFLOPS[_lib.TTK_BF16 if df.dtype == _lib.TTK_FP8 else df.dtype]
		achieved, peak, unit, bound = r["work"] / sec / 1e12, peak_t, "TFLOP/s", "mfma"
	else:
		achieved, peak, unit, bound = r["work"] / sec / 1e9, PEAK_HBM_GBS, "GB/s", "hbm"
	traffic, source = _pmc_traffic(kind)
	# `traffic`: HBM bytes per launch from PMC counters.  They cannot be collected inside this process (rocprofv3 has to own it), so the
	# live line carries null and the committed PMC summary is quoted beside it with its source.
	return {"bound": bound, "achieved": achieved, "peak": peak, "unit": unit, "frac": achieved / peak, "traffic": None,
			"traffic_from_profiles": traffic, "traffic_source": source,
			"kernel": KERNEL_NAMES[kind], "launches_per_step": r["launches"], "avg_launch_us": 1e3 * r["ms"] / max(r["launches"], 1),
			"algorithmic_work_per_launch": r["work"] / max(r["launches"], 1), "per_kernel_ms": breakdown}
