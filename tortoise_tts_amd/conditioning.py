"""Conditioning-latent encoders on libttk: `UnifiedVoice.get_conditioning` (models/unified_voice.py:535-542
over ConditioningEncoder :269-293) and `DiffusionTTS.get_conditioning` (models/diffusion.py:1477-1485 over contextual_embedder :1441-1447),
both over `ttk_cond_*`.  One-off per voice: the results are the `cond_latent [b, 1024]` / `diffusion_latents [b, 2048]` the hot path takes.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Mapping

import torch

from . import _lib
from .diffusion import relbias_table
from .weights import ARConfig, DiffusionConfig, ar_conditioning_shapes, diffusion_conditioning_shapes

STEM_CONV1, STEM_DOWN4 = 0, 1


def _pack_blocks(sd: Mapping[str, torch.Tensor], src: str, first: int, n: int, head_dim: int, relpos: bool) -> Dict[str, torch.Tensor]:
	out = {}
	for i in range(n):
		p, q = f"{src}{first + i}.", f"blocks.{i}."
		for leaf in ("norm.weight", "norm.bias", "qkv.weight", "qkv.bias", "proj_out.weight", "proj_out.bias"):
			out[q + leaf] = sd[p + leaf]
		if relpos:
			out[q + "__relbias"] = relbias_table(sd[p + "relative_pos_embeddings.relative_attention_bias.weight"], head_dim)
	return out


class _CondEncoder:
	def __init__(self, packed: Dict[str, torch.Tensor], cfgc: _lib.CondConfigC, device: str):
		self.device = torch.device(device)
		if self.device.type != "cuda":
			raise _lib.TTKError("tortoise_tts_amd runs on an MI355X only (device must be cuda:N)")
		self.lib = _lib.load()
		self.in_channels, self.channels = cfgc.in_channels, cfgc.channels
		names = list(packed.keys())
		views, keep = _lib.weight_views(packed, names)
		self._h = C.c_void_p()
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_cond_create(C.byref(self._h), C.byref(cfgc), views, len(names)), "ttk_cond_create")
		del keep

	def __del__(self):
		h = getattr(self, "_h", None)
		if h:
			self.lib.ttk_cond_destroy(h)
			self._h = None

	def eval(self):
		return self

	def to(self, *a, **k):
		return self

	@torch.inference_mode()
	def forward(self, mel: torch.Tensor) -> torch.Tensor:
		"""One clip per row: mel [b, bands, T] -> [b, channels] f32."""
		if mel.dim() != 3 or mel.shape[1] != self.in_channels:
			raise _lib.TTKError(f"mel must be [b, {self.in_channels}, T], got {tuple(mel.shape)}")
		if mel.shape[0] == 0 or mel.shape[2] == 0:
			raise _lib.TTKError("empty conditioning clip")
		mel = mel.to(self.device, torch.float32).contiguous()
		out = torch.empty(mel.shape[0], self.channels, device=self.device, dtype=torch.float32)
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_cond_encode(self._h, mel.data_ptr(), mel.shape[0], mel.shape[2], out.data_ptr(), _lib.stream_ptr()), "ttk_cond_encode")
		return out

	__call__ = forward


def _check_dtype(dtype: str) -> int:
	if dtype not in ("bf16", "bfloat16", "f32", "fp32", "float32"):
		raise _lib.TTKError("the conditioning encoders run in 'bf16' or 'f32'")
	return _lib.DTYPES[dtype]


class ConditioningEncoder(_CondEncoder):
	"""`UnifiedVoice.conditioning_encoder` + `UnifiedVoice.get_conditioning`; takes the parent's state_dict (keys `conditioning_encoder.*`)."""

	def __init__(self, state_dict: Mapping[str, torch.Tensor], cfg: ARConfig = ARConfig(), dtype: str = "bf16", device: str = "cuda:0",
				 spec_dim: int = 80, attn_blocks: int = 6):
		missing = [n for n in ar_conditioning_shapes(cfg, spec_dim, attn_blocks) if n not in state_dict]
		if missing:
			raise _lib.TTKError(f"state_dict lacks {len(missing)} conditioning_encoder tensors, e.g. {missing[:3]}")
		d = cfg.model_dim
		packed = {"stem.0.weight": state_dict["conditioning_encoder.init.weight"].reshape(d, spec_dim), "stem.0.bias": state_dict["conditioning_encoder.init.bias"]}
		packed.update(_pack_blocks(state_dict, "conditioning_encoder.attn.", 0, attn_blocks, cfg.head_dim, False))
		super().__init__(packed, _lib.CondConfigC(spec_dim, d, cfg.heads, attn_blocks, STEM_CONV1, 0, 0, _check_dtype(dtype)), device)

	@torch.inference_mode()
	def get_conditioning(self, speech_conditioning_input: torch.Tensor) -> torch.Tensor:
		"""unified_voice.py:535-542: [b, 80, T] or [b, n, 80, T] -> mean over the n clips of the per-clip encodings [b, model_dim]."""
		x = speech_conditioning_input.unsqueeze(1) if speech_conditioning_input.dim() == 3 else speech_conditioning_input
		b, n = x.shape[0], x.shape[1]
		enc = self.forward(x.reshape(b * n, x.shape[2], x.shape[3]))        # the clips of one call share T: one batched pass
		return enc.reshape(b, n, -1).mean(dim=1)


class ContextualEmbedder(_CondEncoder):
	"""`DiffusionTTS.contextual_embedder` + `DiffusionTTS.get_conditioning`; takes the parent's state_dict (keys `contextual_embedder.*`)."""

	def __init__(self, state_dict: Mapping[str, torch.Tensor], cfg: DiffusionConfig = DiffusionConfig(), dtype: str = "bf16", device: str = "cuda:0"):
		missing = [n for n in diffusion_conditioning_shapes(cfg) if n not in state_dict]
		if missing:
			raise _lib.TTKError(f"state_dict lacks {len(missing)} contextual_embedder tensors, e.g. {missing[:3]}")
		ch = cfg.model_channels
		packed = {
			"stem.0.weight": state_dict["contextual_embedder.0.weight"].reshape(ch, cfg.in_channels * 3), "stem.0.bias": state_dict["contextual_embedder.0.bias"],
			"stem.1.weight": state_dict["contextual_embedder.1.weight"].reshape(2 * ch, ch * 3), "stem.1.bias": state_dict["contextual_embedder.1.bias"],
		}
		packed.update(_pack_blocks(state_dict, "contextual_embedder.", 2, 5, 2 * ch // cfg.num_heads, True))
		super().__init__(packed, _lib.CondConfigC(cfg.in_channels, 2 * ch, cfg.num_heads, 5, STEM_DOWN4, 1, 1, _check_dtype(dtype)), device)

	@torch.inference_mode()
	def get_conditioning(self, conditioning_input: torch.Tensor) -> torch.Tensor:
		"""diffusion.py:1477-1485: [b, 100, T] or [b, n, 100, T] -> mean over all positions of all n clips [b, 2 * model_channels].  The
		clips of one call have one length, so the mean of the concatenation is the mean of the per-clip means."""
		x = conditioning_input.unsqueeze(1) if conditioning_input.dim() == 3 else conditioning_input
		b, n = x.shape[0], x.shape[1]
		enc = self.forward(x.reshape(b * n, x.shape[2], x.shape[3]))
		return enc.reshape(b, n, -1).mean(dim=1)
