"""CLVP candidate scoring on libttk: the reference's `clvp(text_tokens.repeat(B, 1), codes,
return_loss=False)` call (`inference.py:392-396`, `models/clvp.py:100-131`) over `ttk_clvp_*`, plus the candidate choice it feeds.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping

import torch

from . import _lib
from .weights import CLVPConfig, clvp_shapes


def pack_state_dict(sd: Mapping[str, torch.Tensor], cfg: CLVPConfig) -> Dict[str, torch.Tensor]:
	"""`CLVP.state_dict()` -> what ttk_clvp_create reads: q / k / v projections of every attention stacked into one matrix (one GEMM), and
	the rotary frequencies computed exactly as `RotaryEmbedding(32)` does (xtransformers.py:266-270, dim = max(64 // 2, 32))."""
	out = {k: v for k, v in sd.items() if "inv_freq" not in k}
	for enc in ("text_transformer", "speech_transformer"):
		for i in range(cfg.depth):
			p = f"{enc}.transformer.attn_layers.layers.{2 * i}.1.wrap."
			out[p + "__qkv.weight"] = torch.cat([sd[p + f"to_{c}.weight"].to(torch.float32) for c in "qkv"], dim=0)
	out["temperature"] = sd["temperature"].to(torch.float32).reshape(1)
	out["__rotary_inv_freq"] = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
	return out


class CLVPConfigC(C.Structure):
	_fields_ = [(n, C.c_int) for n in ("dim", "heads", "depth", "inner", "num_text_tokens", "num_speech_tokens", "dtype")]


class CLVP:
	"""`clvp = load_model("clvp")` of the reference, scoring side only (eval mode: the random token masks of training are off)."""

	def __init__(self, state_dict: Mapping[str, torch.Tensor], cfg: CLVPConfig = CLVPConfig(), dtype: str = "bf16", device: str = "cuda:0"):
		self.cfg = cfg
		self.device = torch.device(device)
		if self.device.type != "cuda":
			raise _lib.TTKError("tortoise_tts_amd runs on an MI355X only (device must be cuda:N)")
		if dtype not in ("bf16", "bfloat16", "f32", "fp32", "float32"):
			raise _lib.TTKError("CLVP runs in 'bf16' or 'f32'")
		self.lib = _lib.load()
		missing = [n for n in clvp_shapes(cfg) if n not in state_dict]
		if missing:
			raise _lib.TTKError(f"state_dict lacks {len(missing)} CLVP tensors, e.g. {missing[:3]}")
		sd = pack_state_dict(state_dict, cfg)
		names = [n for n in sd if not n.endswith(("to_q.weight", "to_k.weight", "to_v.weight"))]
		views, keep = _lib.weight_views(sd, names)
		c = CLVPConfigC(cfg.dim, cfg.heads, cfg.depth, cfg.inner, cfg.num_text_tokens, cfg.num_speech_tokens, _lib.DTYPES[dtype])
		self._h = C.c_void_p()
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_clvp_create(C.byref(self._h), C.byref(c), views, len(names)), "ttk_clvp_create")
		del keep

	def __del__(self):
		h = getattr(self, "_h", None)
		if h:
			self.lib.ttk_clvp_destroy(h)
			self._h = None

	def eval(self):
		return self

	def to(self, *a, **k):
		return self

	@torch.inference_mode()
	def forward(self, text: torch.Tensor, speech_tokens: torch.Tensor, return_loss: bool = False) -> torch.Tensor:
		"""clvp.py:100-131: text [B, Tt] (or [1, Tt]), speech_tokens [B, M] int64 -> similarity [B] f32."""
		if return_loss:
			raise NotImplementedError("the contrastive training loss is not on the inference path")
		c = self.cfg
		text = text.to(self.device, torch.int64).contiguous()
		speech_tokens = speech_tokens.to(self.device, torch.int64).contiguous()
		B, M = speech_tokens.shape
		if text.dim() != 2 or text.shape[0] not in (1, B):
			raise _lib.TTKError(f"text must be [1, Tt] or [{B}, Tt], got {tuple(text.shape)}")
		if int(text.max()) >= c.num_text_tokens or int(text.min()) < 0 or int(speech_tokens.max()) >= c.num_speech_tokens or int(speech_tokens.min()) < 0:
			raise _lib.TTKError("token id outside the embedding table (the reference's nn.Embedding raises IndexError here)")
		if text.shape[0] == B and B > 1 and bool((text == text[:1]).all()):
			text = text[:1].contiguous()                 # `text_tokens.repeat(B, 1)`: encode the line once
		scores = torch.empty(B, device=self.device, dtype=torch.float32)
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_clvp_score(self._h, text.data_ptr(), text.shape[0], text.shape[1], speech_tokens.data_ptr(), B, M, scores.data_ptr(),
											   _lib.stream_ptr()), "ttk_clvp_score")
		return scores

	__call__ = forward
