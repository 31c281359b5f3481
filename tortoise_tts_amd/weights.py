"""Weight tables of the two hot-path networks, in the reference's `state_dict` key names, and a
seeded synthetic generator for them.

The key names/shapes ARE the de-facto wire format of the reference checkpoints
(`/root/reference/tortoise_tts/models/__init__.py:104-110,163-167` loads plain `state_dict`s of
`UnifiedVoice` `models/unified_voice.py:334-451` and `DiffusionTTS` `models/diffusion.py:1389-1465`).
No pretrained weights exist offline, so every test, fixture and bench run uses
`synth_state_dict`: each tensor is drawn from its own CPU mt19937 stream keyed by (seed, crc32(name)),
so any subset can be regenerated bit-identically on any machine without the reference.
"""
from __future__ import annotations

import dataclasses
import zlib
from typing import Dict, Tuple

import torch


@dataclasses.dataclass(frozen=True)
class ARConfig:
	"""`UnifiedVoice.__init__` defaults, models/unified_voice.py:335-355."""
	layers: int = 30
	model_dim: int = 1024
	heads: int = 16
	max_text_tokens: int = 402
	max_mel_tokens: int = 604
	max_conditioning_inputs: int = 2
	number_text_tokens: int = 255
	start_text_token: int = 255
	stop_text_token: int = 0
	number_mel_codes: int = 8194
	start_mel_token: int = 8192
	stop_mel_token: int = 8193
	mel_length_compression: int = 1024

	@property
	def head_dim(self):
		return self.model_dim // self.heads

	@property
	def max_mel_seq_len(self):  # unified_voice.py:405
		return self.max_mel_tokens + 2 + self.max_conditioning_inputs

	@property
	def max_text_seq_len(self):  # unified_voice.py:406
		return self.max_text_tokens + 2


@dataclasses.dataclass(frozen=True)
class DiffusionConfig:
	"""`DiffusionTTS.__init__` defaults, models/diffusion.py:1390-1404."""
	model_channels: int = 1024
	num_layers: int = 10
	in_channels: int = 100
	in_latent_channels: int = 1024
	out_channels: int = 200
	num_heads: int = 16

	@property
	def head_dim(self):
		return self.model_channels // self.num_heads


@dataclasses.dataclass(frozen=True)
class VocoderConfig:
	"""BigVGAN generator hyper-parameters (`BigVGAN.__init__`, models/bigvgan.py:419-486, reads them from a JSON that the reference
	downloads, models/__init__.py:129-137).  The JSON is not available offline: these are the published `bigvgan_24khz_100band`
	values entered by hand -- an ASSUMPTION; their product of upsample rates (256) matches the hop the
	reference's mel length formula implies (inference.py:400)."""
	num_mels: int = 100
	upsample_rates: Tuple[int, ...] = (4, 4, 2, 2, 2, 2)
	upsample_kernel_sizes: Tuple[int, ...] = (8, 8, 4, 4, 4, 4)
	upsample_initial_channel: int = 1536
	resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
	resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
	snake_logscale: bool = True
	sampling_rate: int = 24000

	@property
	def hop_size(self):
		h = 1
		for u in self.upsample_rates:
			h *= u
		return h

	def stage_channels(self, i):   # channels after upsampler i
		return self.upsample_initial_channel // (2 ** (i + 1))

	def as_json(self):
		"""the dict `BigVGAN(data=...)` takes (only the generator's keys)"""
		return dict(resblock="1", num_mels=self.num_mels, upsample_rates=list(self.upsample_rates),
					upsample_kernel_sizes=list(self.upsample_kernel_sizes), upsample_initial_channel=self.upsample_initial_channel,
					resblock_kernel_sizes=list(self.resblock_kernel_sizes), resblock_dilation_sizes=[list(d) for d in self.resblock_dilation_sizes],
					activation="snakebeta", snake_logscale=self.snake_logscale, n_fft=1024, hop_size=self.hop_size,
					sampling_rate=self.sampling_rate)


@dataclasses.dataclass(frozen=True)
class CLVPConfig:
	"""`CLVP.__init__` defaults, models/clvp.py:29-46 (the x-transformers branch, `use_xformers=True`)."""
	dim: int = 768                      # dim_text = dim_speech = dim_latent
	num_text_tokens: int = 256
	num_speech_tokens: int = 8192
	depth: int = 20                     # text_enc_depth = speech_enc_depth
	heads: int = 12                     # head width is x-transformers' DEFAULT_DIM_HEAD = 64
	ff_mult: int = 2

	@property
	def inner(self):
		return self.dim * self.ff_mult


AR_SMALL = ARConfig(layers=2, model_dim=128, heads=2)
CLVP_SMALL = CLVPConfig(dim=128, depth=2, heads=2)
CLVP_FULL = CLVPConfig()
VOC_SMALL = VocoderConfig(upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4), upsample_initial_channel=128,
						  resblock_kernel_sizes=(3, 7), resblock_dilation_sizes=((1, 3, 5), (1, 3, 5)))
VOC_FULL = VocoderConfig()
AR_FULL = ARConfig()
DIFF_SMALL = DiffusionConfig(model_channels=128, num_layers=2, in_latent_channels=128, num_heads=2)
DIFF_FULL = DiffusionConfig()


def ar_shapes(c: ARConfig) -> Dict[str, Tuple[int, ...]]:
	"""Hot-path subset of `UnifiedVoice.state_dict()` (conditioning_encoder / text_head are off-path)."""
	d = c.model_dim
	s: Dict[str, Tuple[int, ...]] = {
		"text_embedding.weight": (c.number_text_tokens + 1, d),
		"mel_embedding.weight": (c.number_mel_codes, d),
		"mel_pos_embedding.emb.weight": (c.max_mel_seq_len, d),
		"text_pos_embedding.emb.weight": (c.max_text_seq_len, d),
		"gpt.ln_f.weight": (d,), "gpt.ln_f.bias": (d,),
		"final_norm.weight": (d,), "final_norm.bias": (d,),
		"mel_head.weight": (c.number_mel_codes, d), "mel_head.bias": (c.number_mel_codes,),
	}
	for i in range(c.layers):
		p = f"gpt.h.{i}."
		s.update({
			p + "ln_1.weight": (d,), p + "ln_1.bias": (d,),
			p + "attn.c_attn.weight": (d, 3 * d), p + "attn.c_attn.bias": (3 * d,),   # HF Conv1D: [in, out]
			p + "attn.c_proj.weight": (d, d), p + "attn.c_proj.bias": (d,),
			p + "ln_2.weight": (d,), p + "ln_2.bias": (d,),
			p + "mlp.c_fc.weight": (d, 4 * d), p + "mlp.c_fc.bias": (4 * d,),
			p + "mlp.c_proj.weight": (4 * d, d), p + "mlp.c_proj.bias": (d,),
		})
	return s


def _attn_shapes(p: str, ch: int, heads: int):
	return {
		p + "norm.weight": (ch,), p + "norm.bias": (ch,),
		p + "qkv.weight": (3 * ch, ch, 1), p + "qkv.bias": (3 * ch,),
		p + "proj_out.weight": (ch, ch, 1), p + "proj_out.bias": (ch,),
		p + "relative_pos_embeddings.relative_attention_bias.weight": (32, heads),
	}


def _resblock_shapes(p: str, ch: int):
	return {
		p + "in_layers.0.weight": (ch,), p + "in_layers.0.bias": (ch,),
		p + "in_layers.2.weight": (ch, ch, 1), p + "in_layers.2.bias": (ch,),
		p + "emb_layers.1.weight": (2 * ch, ch), p + "emb_layers.1.bias": (2 * ch,),
		p + "out_layers.0.weight": (ch,), p + "out_layers.0.bias": (ch,),
		p + "out_layers.3.weight": (ch, ch, 3), p + "out_layers.3.bias": (ch,),
	}


def diffusion_shapes(c: DiffusionConfig) -> Dict[str, Tuple[int, ...]]:
	"""Hot-path subset of `DiffusionTTS.state_dict()` (contextual_embedder / code_converter /
	code_embedding / mel_head are off-path: conditioning latents are inputs here (`diffusion_conditioning_shapes` lists the encoder's tensors) and
	aligned conditioning is always a latent at inference, inference.py:402)."""
	ch = c.model_channels
	s: Dict[str, Tuple[int, ...]] = {
		"unconditioned_embedding": (1, ch, 1),
		"inp_block.weight": (ch, c.in_channels, 3), "inp_block.bias": (ch,),
		"time_embed.0.weight": (ch, ch), "time_embed.0.bias": (ch,),
		"time_embed.2.weight": (ch, ch), "time_embed.2.bias": (ch,),
		"code_norm.weight": (ch,), "code_norm.bias": (ch,),
		"latent_conditioner.0.weight": (ch, c.in_latent_channels, 3), "latent_conditioner.0.bias": (ch,),
		"integrating_conv.weight": (ch, 2 * ch, 1), "integrating_conv.bias": (ch,),
		"out.0.weight": (ch,), "out.0.bias": (ch,),
		"out.2.weight": (c.out_channels, ch, 3), "out.2.bias": (c.out_channels,),
	}
	for i in range(1, 5):
		s.update(_attn_shapes(f"latent_conditioner.{i}.", ch, c.num_heads))
	for i in range(3):
		s.update(_resblock_shapes(f"conditioning_timestep_integrator.{i}.resblk.", ch))
		s.update(_attn_shapes(f"conditioning_timestep_integrator.{i}.attn.", ch, c.num_heads))
	for i in range(c.num_layers):
		s.update(_resblock_shapes(f"layers.{i}.resblk.", ch))
		s.update(_attn_shapes(f"layers.{i}.attn.", ch, c.num_heads))
	for i in range(c.num_layers, c.num_layers + 3):
		s.update(_resblock_shapes(f"layers.{i}.", ch))
	return s


def ar_conditioning_shapes(c: ARConfig, spec_dim: int = 80, attn_blocks: int = 6) -> Dict[str, Tuple[int, ...]]:
	"""`UnifiedVoice.conditioning_encoder` (ConditioningEncoder, models/unified_voice.py:269-293, built at :397 with
	`num_attn_heads=heads`): 1x1 conv spec_dim -> model_dim, then AttentionBlocks without relative position bias."""
	d = c.model_dim
	s: Dict[str, Tuple[int, ...]] = {"conditioning_encoder.init.weight": (d, spec_dim, 1), "conditioning_encoder.init.bias": (d,)}
	for i in range(attn_blocks):
		blk = _attn_shapes(f"conditioning_encoder.attn.{i}.", d, c.heads)
		s.update({k: v for k, v in blk.items() if "relative_pos" not in k})
	return s


def diffusion_conditioning_shapes(c: DiffusionConfig) -> Dict[str, Tuple[int, ...]]:
	"""`DiffusionTTS.contextual_embedder` (models/diffusion.py:1441-1447): two stride-2 k=3 convs in_channels -> ch -> 2ch, then
	five AttentionBlocks of 2ch channels and `num_heads` heads (head width 2ch / heads = 128) with relative position bias."""
	ch = c.model_channels
	s: Dict[str, Tuple[int, ...]] = {
		"contextual_embedder.0.weight": (ch, c.in_channels, 3), "contextual_embedder.0.bias": (ch,),
		"contextual_embedder.1.weight": (2 * ch, ch, 3), "contextual_embedder.1.bias": (2 * ch,),
	}
	for i in range(2, 7):
		s.update(_attn_shapes(f"contextual_embedder.{i}.", 2 * ch, c.num_heads))
	return s


def vocoder_shapes(c: VocoderConfig) -> Dict[str, Tuple[int, ...]]:
	"""`BigVGAN.state_dict()` with weight norm folded (plain `weight` instead of `weight_g` / `weight_v`) and without the constant
	anti-aliasing filter buffers (identical for every Activation1d; recomputed, `vocoder.aa_filter`)."""
	ch0 = c.upsample_initial_channel
	s: Dict[str, Tuple[int, ...]] = {"conv_pre.weight": (ch0, c.num_mels, 7), "conv_pre.bias": (ch0,)}
	nk = len(c.resblock_kernel_sizes)
	ch = ch0
	for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
		cin, ch = ch0 // (2 ** i), ch0 // (2 ** (i + 1))
		s[f"ups.{i}.0.weight"] = (cin, ch, k)                 # ConvTranspose1d: [in, out, k]
		s[f"ups.{i}.0.bias"] = (ch,)
		for j, kk in enumerate(c.resblock_kernel_sizes):
			p = f"resblocks.{i * nk + j}."
			for m in range(3):
				s[p + f"convs1.{m}.weight"] = (ch, ch, kk); s[p + f"convs1.{m}.bias"] = (ch,)
				s[p + f"convs2.{m}.weight"] = (ch, ch, kk); s[p + f"convs2.{m}.bias"] = (ch,)
			for m in range(6):
				s[p + f"activations.{m}.act.alpha"] = (ch,); s[p + f"activations.{m}.act.beta"] = (ch,)
	s["activation_post.act.alpha"] = (ch,); s["activation_post.act.beta"] = (ch,)
	s["conv_post.weight"] = (1, ch, 7); s["conv_post.bias"] = (1,)
	return s


def clvp_shapes(c: CLVPConfig) -> Dict[str, Tuple[int, ...]]:
	"""`CLVP.state_dict()` without the constant rotary `inv_freq` buffers.  Layer 2i is attention, 2i+1 feed-forward; index `.0.0` is the
	pre-branch RMSNorm, `.1.wrap` the block inside arch_utils.CheckpointedLayer."""
	d, qk = c.dim, c.heads * 64
	s: Dict[str, Tuple[int, ...]] = {"temperature": (), "text_emb.weight": (c.num_text_tokens, d), "speech_emb.weight": (c.num_speech_tokens, d),
									 "to_text_latent.weight": (d, d), "to_speech_latent.weight": (d, d)}
	for enc in ("text_transformer", "speech_transformer"):
		p = enc + ".transformer."
		for i in range(c.depth):
			a, f = p + f"attn_layers.layers.{2 * i}.", p + f"attn_layers.layers.{2 * i + 1}."
			s[a + "0.0.g"] = (d,)
			for w in ("to_q", "to_k", "to_v"):
				s[a + f"1.wrap.{w}.weight"] = (qk, d)
			s[a + "1.wrap.to_out.weight"] = (d, qk); s[a + "1.wrap.to_out.bias"] = (d,)
			s[f + "0.0.g"] = (d,)
			s[f + "1.wrap.net.0.proj.weight"] = (2 * c.inner, d); s[f + "1.wrap.net.0.proj.bias"] = (2 * c.inner,)
			s[f + "1.wrap.net.3.weight"] = (d, c.inner); s[f + "1.wrap.net.3.bias"] = (d,)
		s[p + "norm.weight"] = (d,); s[p + "norm.bias"] = (d,)
	return s


def _gain_for(name: str, shape: Tuple[int, ...]) -> Tuple[str, float]:
	"""(kind, std) of the synthetic draw for a key.  Chosen so that every op on the path is exercised
	with O(1) activations: norm scales near 1, matrices fan-in scaled, residual-branch outputs damped.
	The reference zero-initialises `proj_out` (models/arch_utils.py:172); a zero matrix would leave
	the attention untested, so it is drawn like every other matrix."""
	leaf = name.rsplit(".", 1)[-1]
	if name == "temperature":
		return "normal", 0.3
	if leaf == "g":                     # RMSNorm gain
		return "norm_scale", 0.1
	if leaf in ("alpha", "beta"):     # snake parameters, log scale: exp(.) stays within ~[0.5, 2]
		return "normal", 0.3
	if name.startswith("ups.") and leaf == "weight":   # ConvTranspose1d [in, out, k]: each output sample sees in * k / stride taps
		return "normal", 1.0 / (shape[0] * 2) ** 0.5
	if ".convs2." in name and leaf == "weight":
		return "normal", 0.5 / (shape[1] * shape[2]) ** 0.5
	if name == "unconditioned_embedding":
		return "normal", 1.0
	if "relative_attention_bias" in name:
		return "normal", 0.2
	if "embedding" in name:   # token / position tables
		return "normal", 0.5
	is_norm = (len(shape) == 1 and leaf == "weight")
	if is_norm:
		return "norm_scale", 0.1
	if leaf == "bias":
		return "normal", 0.05
	fan_in = shape[0] if ("gpt.h." in name) else 1     # HF Conv1D is [in, out]
	if fan_in == 1:
		fan_in = 1
		for s in shape[1:]:
			fan_in *= s
	gain = 1.0
	if name.endswith("c_proj.weight") or "proj_out" in name or "out_layers.3" in name:
		gain = 0.5
	if "emb_layers" in name:
		gain = 0.5
	return "normal", gain / (fan_in ** 0.5)


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
	g = torch.Generator(device="cpu")
	g.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 63 - 1))
	kind, std = _gain_for(name, shape)
	t = torch.randn(shape, generator=g, dtype=torch.float32) * std
	if kind == "norm_scale":
		t = t + 1.0
	return t


def round_to_bf16(t: torch.Tensor) -> torch.Tensor:
	"""f32 tensor whose values are exactly representable in bf16 (round-to-nearest-even)."""
	return t.to(torch.bfloat16).to(torch.float32)


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int, bf16_exact: bool = False) -> Dict[str, torch.Tensor]:
	"""Seeded synthetic weights.  `bf16_exact` rounds every matrix (ndim >= 2) to a bf16-representable
	value so an f32 oracle and a bf16-weight device path consume the same numbers."""
	out = {}
	for name, shape in shapes.items():
		t = synth_tensor(name, shape, seed)
		if bf16_exact and t.dim() >= 2:
			t = round_to_bf16(t)
		out[name] = t
	return out


# ------------------------------------------------------------------------------------------------------------------------------------
# The "stress" family: the seeded weights above moved to the numerical regime a TRAINED checkpoint lives in.  `gain / sqrt(fan_in)` Gaussians
# give logits of std 1 (max token probability < 1 %), attention scores below 8 and unit-scale residual streams; trained TorToiSe weights
# (models/__init__.py:23-44, download only) give peaked softmax rows, |score| in the tens, outlier residual channels and GroupNorm groups of
# almost no variance.  The transforms are deterministic functions of the seeded tensors, so the fixture generator (oracle/make_golden.py, through
# the reference's classes) and the tests (oracle and HIP path, on any box) build identical weights.
# ------------------------------------------------------------------------------------------------------------------------------------
AR_STRESS_CHANNELS = (5, 77)          # the two residual channels of the `outlier` variant (both < AR_SMALL.model_dim)
DIFF_STRESS_GROUP = 3                 # the GroupNorm32 group every ResBlock's in_layers conv leaves almost constant
DIFF_STRESS_FLAT_VALUE = 3.0
DIFF_STRESS_FLAT_GAIN = 1e-2


def stress_ar(sd: Dict[str, torch.Tensor], c: ARConfig, variant: str = "peaked") -> Dict[str, torch.Tensor]:
	"""`peaked`: `mel_head.weight` x 8 (logits of std ~8: max token probability > 0.5, top-k / top-p / typical cuts inside a sharp distribution) and the
	q / k columns (+ biases) of every `attn.c_attn` x 4 (GPT-2 scores x 16: attention rows close to one-hot, |score| in the tens).
	`outlier`: `peaked` plus two channels of the four embedding tables x 300 -- massive activations in fixed residual channels, the stream a LayerNorm
	(and the decode path's folded LayerNorm) then has to normalise."""
	assert variant in ("peaked", "outlier"), variant
	out, d = dict(sd), c.model_dim
	out["mel_head.weight"] = sd["mel_head.weight"] * 8.0
	for i in range(c.layers):
		for leaf in ("weight", "bias"):
			k = f"gpt.h.{i}.attn.c_attn.{leaf}"
			t = sd[k].clone()
			t[..., :2 * d] *= 4.0
			out[k] = t
	if variant == "outlier":
		for k in ("text_embedding.weight", "mel_embedding.weight", "mel_pos_embedding.emb.weight", "text_pos_embedding.emb.weight"):
			t = sd[k].clone()
			t[:, list(AR_STRESS_CHANNELS)] *= 300.0
			out[k] = t
	return out


def stress_diffusion(sd: Dict[str, torch.Tensor], c: DiffusionConfig) -> Dict[str, torch.Tensor]:
	"""Every AttentionBlock: the q / k rows of `qkv` (head-major [H][3][hd] split, arch_utils.py:76-79) x 4 and `relative_attention_bias` x 10 -- scores of
	+-60 .. 100, softmax rows close to one-hot.  Every ResBlock: `emb_layers` x 4 (large scale / shift) and the `in_layers` conv rows of GroupNorm group
	DIFF_STRESS_GROUP x 1e-2 with bias 3.0: `out_layers.0` normalises a group whose channels are 3.0 +- 1e-2 (mean^2 / variance ~ 1e5)."""
	out, ch = dict(sd), c.model_channels
	hd = c.head_dim
	groups = 32
	if ch <= 16:                               # arch_utils.py:24-44 normalization()
		groups = 8
	elif ch <= 64:
		groups = 16
	while ch % groups:
		groups //= 2
	cpg = ch // groups
	qk_rows = torch.tensor([r for r in range(3 * ch) if r % (3 * hd) < 2 * hd])
	flat = slice(DIFF_STRESS_GROUP * cpg, (DIFF_STRESS_GROUP + 1) * cpg)
	for k, v in sd.items():
		if k.endswith("qkv.weight") or k.endswith("qkv.bias"):
			if v.shape[0] != 3 * ch:          # the conditioning encoders' wider blocks are not on this path
				continue
			t = v.clone()
			t[qk_rows] *= 4.0
			out[k] = t
		elif k.endswith("relative_attention_bias.weight"):
			out[k] = v * 10.0
		elif ".emb_layers.1." in k:
			out[k] = v * 4.0
		elif k.endswith("in_layers.2.weight"):
			t = v.clone()
			t[flat] *= DIFF_STRESS_FLAT_GAIN
			out[k] = t
		elif k.endswith("in_layers.2.bias"):
			t = v.clone()
			t[flat] = DIFF_STRESS_FLAT_VALUE
			out[k] = t
	return out


def n_params(shapes: Dict[str, Tuple[int, ...]]) -> int:
	n = 0
	for s in shapes.values():
		k = 1
		for x in s:
			k *= x
		n += k
	return n
