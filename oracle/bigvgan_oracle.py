"""CPU oracle of the BigVGAN generator (BASELINE config 5's "vocoder tail").

TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rules as tortoise_oracle.py).  A functional fp32 restatement of
/root/reference/tortoise_tts/models/bigvgan.py (own code; line numbers below refer to that file), pinned against the reference class
itself run in the build container on the repo's synthetic weights (`oracle/make_golden.py vocoder_small`, `tests/golden/`).

  kaiser_sinc_filter1d  :40-69     the one 12-tap low-pass every Activation1d uses (cutoff 0.25, half-width 0.3)
  UpSample1d            :113-136   replicate pad 5, 2x zero-stuffing transposed conv with the filter, x2, trim 15 | 15
  DownSample1d          :139-153   replicate pad 5 | 6, stride-2 correlation with the filter
  SnakeBeta             :237-295   x + sin^2(x * e^alpha) / (e^beta + 1e-9)        (alpha_logscale)
  AMPBlock1             :306-358   3 x { act, dilated conv, act, conv, + x }
  BigVGAN.forward       :488-510   conv_pre, 6 x { transposed conv, mean of 3 AMP blocks }, act, conv_post, tanh
  BigVGAN.inference     :522-534   append 10 frames of -11.5129, run, drop the last 10 hops, clamp to [-1, 1]
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> Tensor:
	""":40-69 (even kernel sizes only, which is all the model uses)."""
	half = kernel_size // 2
	delta_f = 4 * half_width
	A = 2.285 * (half - 1) * math.pi * delta_f + 7.95
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


def aa_filter() -> Tensor:
	"""Activation1d(up_ratio=2, down_ratio=2, kernel 12) :158-181: both resamplers get cutoff 0.5/2, half-width 0.6/2."""
	return kaiser_sinc_filter1d(0.25, 0.3, 12)


def upsample2(x: Tensor, f: Tensor) -> Tensor:
	""":113-136 with ratio 2, kernel 12.  x [B, C, L] -> [B, C, 2L]."""
	C = x.shape[1]
	x = F.pad(x, (5, 5), mode="replicate")
	y = 2 * F.conv_transpose1d(x, f.view(1, 1, -1).expand(C, -1, -1), stride=2, groups=C)
	return y[..., 15:-15]


def downsample2(x: Tensor, f: Tensor) -> Tensor:
	""":139-153 / :72-110 with ratio 2, kernel 12."""
	C = x.shape[1]
	x = F.pad(x, (5, 6), mode="replicate")
	return F.conv1d(x, f.view(1, 1, -1).expand(C, -1, -1), stride=2, groups=C)


def snakebeta(x: Tensor, alpha: Tensor, beta: Tensor, logscale: bool) -> Tensor:
	a, b = alpha[None, :, None], beta[None, :, None]
	if logscale:
		a, b = torch.exp(a), torch.exp(b)
	return x + (1.0 / (b + 1e-9)) * torch.sin(x * a) ** 2


def activation1d(x: Tensor, alpha: Tensor, beta: Tensor, f: Tensor, logscale: bool) -> Tensor:
	return downsample2(snakebeta(upsample2(x, f), alpha, beta, logscale), f)


class BigVGANOracle:
	def __init__(self, w: Dict[str, Tensor], cfg):
		self.w, self.cfg = w, cfg
		self.f = aa_filter()

	def _act(self, x, prefix):
		return activation1d(x, self.w[prefix + "act.alpha"], self.w[prefix + "act.beta"], self.f, self.cfg.snake_logscale)

	def amp_block(self, x: Tensor, n: int, k: int, dil) -> Tensor:
		p = f"resblocks.{n}."
		for m in range(3):
			xt = self._act(x, p + f"activations.{2 * m}.")
			xt = F.conv1d(xt, self.w[p + f"convs1.{m}.weight"], self.w[p + f"convs1.{m}.bias"], dilation=dil[m], padding=(k * dil[m] - dil[m]) // 2)
			xt = self._act(xt, p + f"activations.{2 * m + 1}.")
			xt = F.conv1d(xt, self.w[p + f"convs2.{m}.weight"], self.w[p + f"convs2.{m}.bias"], padding=(k - 1) // 2)
			x = xt + x
		return x

	def forward(self, mel: Tensor) -> Tensor:
		c, w = self.cfg, self.w
		x = F.conv1d(mel, w["conv_pre.weight"], w["conv_pre.bias"], padding=3)
		nk = len(c.resblock_kernel_sizes)
		for i, (u, k) in enumerate(zip(c.upsample_rates, c.upsample_kernel_sizes)):
			x = F.conv_transpose1d(x, w[f"ups.{i}.0.weight"], w[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
			xs = None
			for j in range(nk):
				y = self.amp_block(x, i * nk + j, c.resblock_kernel_sizes[j], c.resblock_dilation_sizes[j])
				xs = y if xs is None else xs + y
			x = xs / nk
		x = self._act(x, "activation_post.")
		x = F.conv1d(x, w["conv_post.weight"], w["conv_post.bias"], padding=3)
		return torch.tanh(x)

	def inference(self, mel: Tensor) -> Tensor:
		""":522-534.  mel [B, num_mels, T] (denormalised log-mel) -> audio [B, 1, T * hop] in [-1, 1]."""
		pad = torch.full((mel.shape[0], self.cfg.num_mels, 10), -11.5129)
		audio = self.forward(torch.cat((mel, pad), dim=2))
		audio = audio[:, :, :-(self.cfg.hop_size * 10)]
		return audio.clamp(min=-1, max=1)


def fold_weight_norm(sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
	"""`weight_g` / `weight_v` (torch.nn.utils.weight_norm, dim 0) -> `weight = g * v / ||v||`, norm over all dims but 0."""
	out = {}
	for k, v in sd.items():
		if k.endswith(".weight_v"):
			g = sd[k[:-2] + "_g"]
			norm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, *([1] * (v.dim() - 1)))
			out[k[:-2]] = v * (g / norm)
		elif k.endswith(".weight_g"):
			continue
		else:
			out[k] = v
	return out
