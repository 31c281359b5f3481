"""TEST INFRASTRUCTURE -- CPU restatement (plain torch f32) of the two conditioning-latent encoders, SURVEY.md section 8(f) row 4.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

  ar_get_conditioning         UnifiedVoice.get_conditioning   models/unified_voice.py:535-542 over ConditioningEncoder :269-293
  diffusion_get_conditioning  DiffusionTTS.get_conditioning   models/diffusion.py:1477-1485 over contextual_embedder :1441-1447
  attention_block             AttentionBlock._forward + QKVAttentionLegacy   models/arch_utils.py:136-190, :59-94

Pinned by tests/golden/cond_small.npz and cond_full.npz, which oracle/make_golden.py writes by running the reference classes
themselves on the repo's seeded synthetic weights.
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F
from torch import Tensor

from tortoise_oracle import group_norm32, rel_pos_bias

W = Dict[str, Tensor]


def normalization_groups(channels: int) -> int:
	"""arch_utils.py:27-44 `normalization`: 32 groups, fewer for narrow tensors, halved until it divides."""
	groups = 32
	if channels <= 16:
		groups = 8
	elif channels <= 64:
		groups = 16
	while channels % groups != 0:
		groups = int(groups / 2)
	return groups


def attention_block(w: W, p: str, x: Tensor, heads: int) -> Tensor:
	"""x + proj_out(attention(qkv(GN(x)))); head-major [H, 3, ch] channel split, q and k each scaled by ch^-1/4, optional
	relative position bias scaled by sqrt(ch) (arch_utils.py:174), softmax in float."""
	b, c, T = x.shape
	h = group_norm32(x, w[p + "norm.weight"], w[p + "norm.bias"], normalization_groups(c))
	qkv = F.conv1d(h, w[p + "qkv.weight"], w[p + "qkv.bias"])
	ch = c // heads
	q, k, v = qkv.reshape(b * heads, ch * 3, T).split(ch, dim=1)
	s = 1 / math.sqrt(math.sqrt(ch))
	weight = torch.einsum("bct,bcs->bts", q * s, k * s)
	key = p + "relative_pos_embeddings.relative_attention_bias.weight"
	if key in w:
		weight = (weight.reshape(b, heads, T, T) + rel_pos_bias(w[key], T, T, ch ** 0.5)).reshape(b * heads, T, T)
	weight = torch.softmax(weight.float(), dim=-1)
	a = torch.einsum("bts,bcs->bct", weight, v).reshape(b, -1, T)
	return x + F.conv1d(a, w[p + "proj_out.weight"], w[p + "proj_out.bias"])


def conditioning_encoder(w: W, mel: Tensor, heads: int, attn_blocks: int = 6, mean: bool = False) -> Tensor:
	"""ConditioningEncoder.forward (unified_voice.py:286-292): mel [b, 80, T] -> [b, d]; position 0 unless `mean`."""
	h = F.conv1d(mel, w["conditioning_encoder.init.weight"], w["conditioning_encoder.init.bias"])
	for i in range(attn_blocks):
		h = attention_block(w, f"conditioning_encoder.attn.{i}.", h, heads)
	return h.mean(dim=2) if mean else h[:, :, 0]


def ar_get_conditioning(w: W, mels: Tensor, heads: int, attn_blocks: int = 6) -> Tensor:
	"""unified_voice.py:535-542: [b, 80, T] or [b, n, 80, T] -> mean over the n clips of the per-clip encodings, [b, d]."""
	x = mels.unsqueeze(1) if mels.dim() == 3 else mels
	return torch.stack([conditioning_encoder(w, x[:, j], heads, attn_blocks) for j in range(x.shape[1])], dim=1).mean(dim=1)


def contextual_embedder(w: W, mel: Tensor, heads: int) -> Tensor:
	"""diffusion.py:1441-1447: mel [b, 100, T] -> [b, 2ch, ceil(ceil(T/2)/2)]."""
	h = F.conv1d(mel, w["contextual_embedder.0.weight"], w["contextual_embedder.0.bias"], stride=2, padding=1)
	h = F.conv1d(h, w["contextual_embedder.1.weight"], w["contextual_embedder.1.bias"], stride=2, padding=1)
	for i in range(2, 7):
		h = attention_block(w, f"contextual_embedder.{i}.", h, heads)
	return h


def diffusion_get_conditioning(w: W, mels: Tensor, heads: int) -> Tensor:
	"""diffusion.py:1477-1485: per-clip embeddings concatenated along time, then the mean over time: [b, 2ch]."""
	x = mels.unsqueeze(1) if mels.dim() == 3 else mels
	return torch.cat([contextual_embedder(w, x[:, j], heads) for j in range(x.shape[1])], dim=-1).mean(dim=-1)
