"""ctypes binding of libttk.so (include/ttk.h).  The product path has NO fallback: if the library is missing or a
call fails, an exception is raised -- nothing here or above ever routes through torch ops or the oracle instead."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Dict, List, Sequence

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTK_LIB") or os.path.join(HERE, "libttk.so")   # TTK_LIB: A/B runs of an experimental build

TTK_F32, TTK_BF16 = 0, 1
TTK_F16 = 4
TTK_FP8W = 2
TTK_FP8 = 3      # diffusion handle only: fp8 activations into the block GEMMs as well (fp8 MFMA); elsewhere it means fp8w
DTYPES = {"f32": TTK_F32, "fp32": TTK_F32, "float32": TTK_F32, "bf16": TTK_BF16, "bfloat16": TTK_BF16, "f16": TTK_F16, "fp16": TTK_F16, "float16": TTK_F16, "half": TTK_F16, "fp8w": TTK_FP8W, "fp8": TTK_FP8}


class TTKError(RuntimeError):
	pass


class WeightView(C.Structure):
	_fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("ndim", C.c_int), ("shape", C.c_int64 * 4)]


class ARConfigC(C.Structure):
	_fields_ = [(n, C.c_int) for n in (
		"layers", "model_dim", "heads", "max_mel_seq_len", "max_text_seq_len", "number_text_tokens_p1", "number_mel_codes",
		"start_text_token", "stop_text_token", "start_mel_token", "stop_mel_token", "dtype", "max_batch", "max_ctx")]


class DiffConfigC(C.Structure):
	_fields_ = [(n, C.c_int) for n in (
		"model_channels", "num_layers", "in_channels", "in_latent_channels", "out_channels", "num_heads", "dtype")]


class CondConfigC(C.Structure):
	_fields_ = [(n, C.c_int) for n in ("in_channels", "channels", "num_heads", "num_blocks", "stem", "relpos", "pool", "dtype")]


class StepC(C.Structure):
	_fields_ = [("t", C.c_int64)] + [(n, C.c_float) for n in (
		"sqrt_recip_ac", "sqrt_recipm1_ac", "sqrt_ac_prev", "sqrt_1m_ac_prev", "coef1", "coef2", "min_log", "max_log", "cfk")] + [
		("sampler", C.c_int), ("nonzero", C.c_int)]


class SampleArgs(C.Structure):
	"""ttk_sample_args (include/ttk.h)"""
	_fields_ = [("scores", C.c_void_p), ("ld", C.c_int64), ("B", C.c_int), ("V", C.c_int), ("q", C.c_void_p), ("ldq", C.c_int64),
				("suppress", C.c_void_p), ("temperature", C.c_float), ("top_k", C.c_int), ("top_p", C.c_float),
				("repetition_penalty", C.c_float), ("stop_token", C.c_int64), ("unfinished", C.c_void_p), ("tok", C.c_void_p),
				("ids", C.c_void_p), ("ids_ld", C.c_int64), ("ids_cols", C.c_int64), ("col", C.c_void_p), ("history", C.c_void_p),
				("hist_ld", C.c_int64), ("hist_off", C.c_int64), ("live_rows", C.c_void_p), ("all_done", C.c_void_p), ("typical_mass", C.c_float)]


class ProfResult(C.Structure):
	_fields_ = [("ms", C.c_double), ("launches", C.c_int64), ("work", C.c_double)]


# every symbol include/ttk.h declares: (restype, argtypes)
_P, _I, _L = C.c_void_p, C.c_int, C.c_int64
SYMBOLS = {
	"ttk_version": (_I, []),
	"ttk_last_error": (C.c_char_p, []),
	"ttk_prof_begin": (_I, []),
	"ttk_prof_end": (_I, [C.POINTER(ProfResult), _I]),
	"ttk_ar_create": (_I, [C.POINTER(_P), C.POINTER(ARConfigC), C.POINTER(WeightView), _I]),
	"ttk_ar_destroy": (_I, [_P]),
	"ttk_ar_prefill": (_I, [_P, _P, _I, _P, _I, _I, _P, _P]),
	"ttk_ar_prefill_prompted": (_I, [_P, _P, _I, _P, _I, _P, _I, _I, _I, _P, _P]),
	"ttk_ar_decode": (_I, [_P, _P, _P, _P, _P]),
	"ttk_ar_decode_next": (_I, [_P, _P, _P, _P]),
	"ttk_ar_last_hidden": (_I, [_P, _P, _P]),
	"ttk_ar_set_hidden_ring": (_I, [_P, _P, _P, _L, _P]),
	"ttk_ar_health": (_I, [_P, C.POINTER(C.c_int), _P]),
	"ttk_ar_set_noise": (_I, [_P, _P, _P, _P]),
	"ttk_ar_prefill_lines": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
	"ttk_graph_launch": (_I, [_P, _P]),
	"ttk_exponential_like_torch": (_I, [_P, _L, _L, _L, _L, _L, _L, _P]),
	"ttk_ar_sample_next": (_I, [_P, C.POINTER(SampleArgs), _P]),
	"ttk_sample_step_warped": (_I, [C.POINTER(SampleArgs), _P]),
	"ttk_ar_latents": (_I, [_P, _P, _P, _I, _P, _I, _I, _P, _P]),
	"ttk_voc_create": (_I, [C.POINTER(_P), _P, C.POINTER(WeightView), _I]),
	"ttk_voc_destroy": (_I, [_P]),
	"ttk_voc_inference": (_I, [_P, _P, _I, _I, _P, _P]),
	"ttk_clvp_create": (_I, [C.POINTER(_P), _P, C.POINTER(WeightView), _I]),
	"ttk_clvp_destroy": (_I, [_P]),
	"ttk_clvp_score": (_I, [_P, _P, _I, _I, _P, _I, _I, _P, _P]),
	"ttk_cond_create": (_I, [C.POINTER(_P), C.POINTER(CondConfigC), C.POINTER(WeightView), _I]),
	"ttk_cond_destroy": (_I, [_P]),
	"ttk_cond_encode": (_I, [_P, _P, _I, _I, _P, _P]),
	"ttk_mel_create": (_I, [C.POINTER(_P), _P, C.POINTER(WeightView), _I]),
	"ttk_mel_destroy": (_I, [_P]),
	"ttk_mel_forward": (_I, [_P, _P, _I, _I, _P, _P]),
	"ttk_resample_fir": (_I, [_P, _I, _I, _P, _I, _I, _I, _P, _I, _P]),
	"ttk_gemm_nt": (_I, [_I, _P, _P, _I, _I, _I, C.c_float, _P, _P, _P]),
	"ttk_fp8_round_weights": (_I, [_P, _L, C.POINTER(C.c_float), _P]),
	"ttk_sample_step": (_I, [_P, _L, _I, _I, _P, _L, _P, C.c_float, _L, _P, _P, _P, _L, _L, _P, _P, _L, _L, _P, _P, _P]),
	"ttk_ar_decode_geometry": (_I, [_I, _I, _I, C.POINTER(C.c_int32)]),
	"ttk_diff_create": (_I, [C.POINTER(_P), C.POINTER(DiffConfigC), C.POINTER(WeightView), _I]),
	"ttk_diff_destroy": (_I, [_P]),
	"ttk_diff_precompute": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
	"ttk_diff_forward": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
	"ttk_diff_begin": (_I, [_P, _P, _I, _I, _P]),
	"ttk_diff_step": (_I, [_P, _P, C.POINTER(StepC), _P, _P]),
	"ttk_diff_sample_ddim": (_I, [_P, _P, _P, _I, _I, C.POINTER(StepC), _I, _P]),
	"ttk_diff_sample_p": (_I, [_P, _P, _P, _I, _I, C.POINTER(StepC), _I, _P, _P]),
	"ttk_diff_sample_ddim_lines": (_I, [_P, _P, _P, _I, _I, C.POINTER(C.c_int), C.POINTER(StepC), _I, _P]),
}

_lib = None


def build(force: bool = False) -> str:
	"""Compile libttk.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
	if force:
		subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "clean"], check=True, capture_output=True)
	r = subprocess.run(["make", "-C", os.path.join(HERE, "csrc"), "-j8"], capture_output=True, text=True)
	if r.returncode != 0:
		raise TTKError("building libttk.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
	return LIB_PATH


def load():
	global _lib
	if _lib is not None:
		return _lib
	if not os.path.exists(LIB_PATH):
		raise TTKError(f"{LIB_PATH} is missing: build it with `make -C tortoise_tts_amd/csrc` "
					   "(or `python -c 'import __graft_entry__ as g; g.build()'`). There is no fallback path.")
	lib = C.CDLL(LIB_PATH)
	for name, (res, args) in SYMBOLS.items():
		fn = getattr(lib, name)   # AttributeError if the ABI lost a symbol
		fn.restype, fn.argtypes = res, args
	if lib.ttk_version() != 1:
		raise TTKError(f"libttk ABI version {lib.ttk_version()} != 1")
	_lib = lib
	return lib


def check(rc: int, what: str):
	if rc != 0:
		raise TTKError(f"{what} failed ({rc}): {load().ttk_last_error().decode()}")


def stream_ptr() -> int:
	return torch.cuda.current_stream().cuda_stream


def ptr(t: torch.Tensor | None) -> int | None:
	return None if t is None else t.data_ptr()


def require_cuda(*tensors: torch.Tensor):
	for t in tensors:
		if t is not None and not t.is_cuda:
			raise TTKError("libttk takes device tensors; got a CPU tensor (the HIP path has no CPU fallback)")


def weight_views(sd: Dict[str, torch.Tensor], names: Sequence[str]):
	"""(array of ttk_weight_view, keepalive list) for f32 contiguous tensors (host or device)."""
	keep: List[torch.Tensor] = []
	arr = (WeightView * len(names))()
	for i, n in enumerate(names):
		t = sd[n].detach().to(torch.float32).contiguous()
		keep.append(t)
		arr[i].name = n.encode()
		arr[i].data = t.data_ptr()
		arr[i].ndim = min(t.dim(), 4)
		for j in range(4):
			arr[i].shape[j] = t.shape[j] if j < t.dim() else 1
	return arr, keep
