"""BigVGAN vocoder on libttk (the "vocoder tail" of BASELINE config 5): the reference's
`vocoder.inference(mels)` call (`inference.py:416-417`, `models/bigvgan.py:522-534`) over `ttk_voc_*`.

Host-side pieces that are Python in the reference too: the Kaiser-windowed sinc low-pass every anti-aliased activation uses
(`kaiser_sinc_filter1d`, bigvgan.py:40-69) and the folding of `torch.nn.utils.weight_norm` parameters into plain weights (the
checkpoint stores `weight_g` / `weight_v`; `BigVGAN.remove_weight_norm`, :512-520, does the same fold before deployment).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Mapping

import torch

from . import _lib
from .weights import VocoderConfig, vocoder_shapes


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> torch.Tensor:
	"""bigvgan.py:40-69 for even kernel sizes: windowed sinc, normalised to unit DC gain."""
	half = kernel_size // 2
	A = 2.285 * (half - 1) * math.pi * (4 * half_width) + 7.95
	if A > 50.0:
		beta = 0.1102 * (A - 8.7)
	elif A >= 21.0:
		beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
	else:
		beta = 0.0
	window = torch.kaiser_window(kernel_size, beta=beta, periodic=False)
	time = torch.arange(-half, half) + 0.5
	f = 2 * cutoff * window * torch.sinc(2 * cutoff * time)
	return f / f.sum()


def aa_filter() -> torch.Tensor:
	"""`Activation1d(up_ratio=2, down_ratio=2, kernel 12)` (bigvgan.py:158-181): cutoff 0.5 / 2, half-width 0.6 / 2 for both resamplers."""
	return kaiser_sinc_filter1d(0.25, 0.3, 12)


def fold_weight_norm(sd: Mapping[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
	"""`weight_g` / `weight_v` (weight_norm over dim 0; also the `parametrizations.weight.original0/1` spelling of newer torch) ->
	`weight = g * v / ||v||`, the norm taken over every dimension but the first; constant filter buffers are dropped."""
	out: Dict[str, torch.Tensor] = {}
	for k, v in sd.items():
		if k.endswith(".filter"):
			continue
		if k.endswith(".weight_g") or k.endswith(".parametrizations.weight.original0"):
			continue
		if k.endswith(".weight_v") or k.endswith(".parametrizations.weight.original1"):
			base = k[:-len(".weight_v")] if k.endswith(".weight_v") else k[:-len(".parametrizations.weight.original1")]
			gk = base + (".weight_g" if k.endswith(".weight_v") else ".parametrizations.weight.original0")
			g = sd[gk].to(torch.float32)
			v = v.to(torch.float32)
			norm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
			out[base + ".weight"] = v * (g / norm)
		else:
			out[k] = v
	return out


class VocConfigC(C.Structure):
	_fields_ = [("num_mels", C.c_int), ("n_ups", C.c_int), ("up_rate", C.c_int * 8), ("up_kernel", C.c_int * 8), ("ch0", C.c_int),
				("n_kernels", C.c_int), ("rb_kernel", C.c_int * 4), ("rb_dil", (C.c_int * 3) * 4), ("snake_logscale", C.c_int),
				("dtype", C.c_int)]


class BigVGAN:
	"""`vocoder = load_model("bigvgan")` of the reference, inference side only."""

	def __init__(self, state_dict: Mapping[str, torch.Tensor], cfg: VocoderConfig = VocoderConfig(), dtype: str = "bf16", device: str = "cuda:0"):
		self.cfg = cfg
		self.device = torch.device(device)
		if self.device.type != "cuda":
			raise _lib.TTKError("tortoise_tts_amd runs on an MI355X only (device must be cuda:N)")
		if dtype not in ("bf16", "bfloat16", "f32", "fp32", "float32"):
			raise _lib.TTKError("the vocoder runs in 'bf16' or 'f32'")
		self.lib = _lib.load()
		self.hop_length = cfg.hop_size
		self.mel_channel = cfg.num_mels
		sd = fold_weight_norm(state_dict)
		names = list(vocoder_shapes(cfg).keys())
		missing = [n for n in names if n not in sd]
		if missing:
			raise _lib.TTKError(f"state_dict lacks {len(missing)} vocoder tensors, e.g. {missing[:3]}")
		sd = {n: sd[n] for n in names}
		sd["__aa_filter"] = aa_filter()
		names.append("__aa_filter")
		views, keep = _lib.weight_views(sd, names)
		c = VocConfigC()
		c.num_mels, c.n_ups, c.ch0, c.n_kernels = cfg.num_mels, len(cfg.upsample_rates), cfg.upsample_initial_channel, len(cfg.resblock_kernel_sizes)
		for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
			c.up_rate[i], c.up_kernel[i] = u, k
		for j, (k, d) in enumerate(zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes)):
			c.rb_kernel[j] = k
			for m in range(3):
				c.rb_dil[j][m] = d[m]
		c.snake_logscale, c.dtype = int(cfg.snake_logscale), _lib.DTYPES[dtype]
		self._h = C.c_void_p()
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_voc_create(C.byref(self._h), C.byref(c), views, len(names)), "ttk_voc_create")
		del keep

	def __del__(self):
		h = getattr(self, "_h", None)
		if h:
			self.lib.ttk_voc_destroy(h)
			self._h = None

	def eval(self, inference: bool = False):
		return self

	def to(self, *a, **k):
		return self

	@torch.inference_mode()
	def inference(self, c: torch.Tensor, z=None) -> torch.Tensor:
		"""bigvgan.py:522-534: c [B, num_mels, T] log-mel -> audio [B, 1, T * hop_length] in [-1, 1] (z, the unused noise input, is ignored
		as the reference's forward ignores it)."""
		if c.dim() != 3 or c.shape[1] != self.cfg.num_mels:
			raise _lib.TTKError(f"mel must be [B, {self.cfg.num_mels}, T], got {tuple(c.shape)}")
		c = c.to(self.device, torch.float32).contiguous()
		B, _, T = c.shape
		audio = torch.empty((B, 1, T * self.hop_length), device=self.device, dtype=torch.float32)
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_voc_inference(self._h, c.data_ptr(), B, T, audio.data_ptr(), _lib.stream_ptr()), "ttk_voc_inference")
		return audio
