"""CPU oracle of CLVP scoring: TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A functional fp32 restatement of the reference's CLVP (x-transformers branch), own code, pinned against the reference class run in the
build container (`oracle/make_golden.py clvp_small`).  Line numbers: models/clvp.py and models/xtransformers.py under /root/reference/tortoise_tts.

  CLVP.forward               clvp.py:100-136     embeddings -> two encoders -> masked mean -> linear -> L2 normalise -> dot * exp(temperature)
  ContinuousTransformerWrapper xtransformers.py:1189-1248   no absolute positions (rotary encoder), final nn.LayerNorm
  AttentionLayers.forward    :841-1015           depth x { x + attn(rmsnorm(x)); x + ff(rmsnorm(x)) }, rotary table of 32 features
  Attention.forward          :578-731            bias-free q/k/v, rotary on the first 32 features of q, k AND v, softmax(q k^T / 8) v, to_out with bias
  RMSNorm                    :337-346            x / max(||x|| * dim^-0.5, 1e-8) * g
  FeedForward / GLU          :431-478            (W x + b) split in two halves: value * gelu(gate); then Linear
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
ROT = 32     # max(dim_head // 2, 32), xtransformers.py:787


def rmsnorm(x: Tensor, g: Tensor) -> Tensor:
	norm = torch.norm(x, dim=-1, keepdim=True) * (x.shape[-1] ** -0.5)
	return x / norm.clamp(min=1e-8) * g


def rotary_table(n: int) -> Tensor:
	inv_freq = 1.0 / (10000 ** (torch.arange(0, ROT, 2).float() / ROT))
	freqs = torch.einsum("i,j->ij", torch.arange(n).float(), inv_freq)
	return torch.cat((freqs, freqs), dim=-1)                       # [n, 32]


def apply_rotary(t: Tensor, freqs: Tensor) -> Tensor:
	"""t [..., n, 32]: t * cos + rotate_half(t) * sin with rotate_half((x1, x2)) = (-x2, x1) over the two halves of 16."""
	x1, x2 = t[..., : ROT // 2], t[..., ROT // 2:]
	return t * freqs.cos() + torch.cat((-x2, x1), dim=-1) * freqs.sin()


def attention(x: Tensor, w: Dict[str, Tensor], p: str, heads: int, freqs: Tensor) -> Tensor:
	b, n, _ = x.shape
	q, k, v = (F.linear(x, w[p + f"to_{c}.weight"]).view(b, n, heads, 64).transpose(1, 2) for c in "qkv")
	q, k, v = (torch.cat((apply_rotary(t[..., :ROT], freqs), t[..., ROT:]), dim=-1) for t in (q, k, v))
	dots = torch.einsum("bhid,bhjd->bhij", q, k) * (64 ** -0.5)
	out = torch.einsum("bhij,bhjd->bhid", dots.softmax(dim=-1), v)
	out = out.transpose(1, 2).reshape(b, n, heads * 64)
	return F.linear(out, w[p + "to_out.weight"], w[p + "to_out.bias"])


def feed_forward(x: Tensor, w: Dict[str, Tensor], p: str) -> Tensor:
	val, gate = F.linear(x, w[p + "net.0.proj.weight"], w[p + "net.0.proj.bias"]).chunk(2, dim=-1)
	return F.linear(val * F.gelu(gate), w[p + "net.3.weight"], w[p + "net.3.bias"])


class CLVPOracle:
	def __init__(self, w: Dict[str, Tensor], cfg):
		self.w, self.cfg = w, cfg

	def encode(self, which: str, ids: Tensor) -> Tensor:
		"""which in {'text', 'speech'}; ids [B, S] -> L2-normalised latent [B, dim]."""
		c, w = self.cfg, self.w
		x = w[which + "_emb.weight"][ids]
		p = which + "_transformer.transformer."
		freqs = rotary_table(ids.shape[1])
		for i in range(c.depth):
			a, f = p + f"attn_layers.layers.{2 * i}.", p + f"attn_layers.layers.{2 * i + 1}."
			x = x + attention(rmsnorm(x, w[a + "0.0.g"]), w, a + "1.wrap.", c.heads, freqs)
			x = x + feed_forward(rmsnorm(x, w[f + "0.0.g"]), w, f + "1.wrap.")
		x = F.layer_norm(x, (c.dim,), w[p + "norm.weight"], w[p + "norm.bias"], 1e-5)
		lat = F.linear(x.mean(dim=1), w[f"to_{which}_latent.weight"])          # masked_mean with an all-true mask (eval mode, clvp.py:110-113)
		return F.normalize(lat, p=2, dim=-1)

	def forward(self, text: Tensor, speech_tokens: Tensor) -> Tensor:
		"""clvp.py:100-131 with return_loss=False: one score per (text row, speech row) pair."""
		t, s = self.encode("text", text), self.encode("speech", speech_tokens)
		return (t * s).sum(dim=-1) * self.w["temperature"].exp()
