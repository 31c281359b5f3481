"""Conditioning-latent encoders on libttk against the reference's `get_conditioning` outputs
(tests/golden/cond_small.npz, cond_full.npz) and the oracle.  GPU only; calls go through the C ABI (`ttk_cond_*`)."""
import numpy as np
import pytest
import torch

import cond_oracle as CO
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
	return torch.from_numpy(np.asarray(a))


def rel_l2(a, b):
	return float((a - b).norm() / b.norm().clamp_min(1e-12))


@pytest.mark.parametrize("name,ar_cfg,diff_cfg", [("cond_small", W.AR_SMALL, W.DIFF_SMALL), ("cond_full", W.AR_FULL, W.DIFF_FULL)])
def test_fp32_equals_reference(golden, name, ar_cfg, diff_cfg):
	"""fp32 mode vs the REFERENCE modules' outputs: both stems, head widths 64 and 128, with and without relative bias, 1 and 2 clips"""
	from tortoise_tts_amd.conditioning import ConditioningEncoder, ContextualEmbedder
	g = golden(name)
	seed = int(g["seed"])
	enc_ar = ConditioningEncoder(W.synth_state_dict(W.ar_conditioning_shapes(ar_cfg), seed), ar_cfg, dtype="f32", device=DEV)
	enc_df = ContextualEmbedder(W.synth_state_dict(W.diffusion_conditioning_shapes(diff_cfg), seed + 1), diff_cfg, dtype="f32", device=DEV)
	mel_ar, mel_df = t(g["mel_ar"]).to(DEV), t(g["mel_diff"]).to(DEV)
	for got, key in ((enc_ar.get_conditioning(mel_ar), "ar_latent"), (enc_ar.get_conditioning(mel_ar[:, 0]), "ar_latent_single"),
					 (enc_df.get_conditioning(mel_df), "diff_latent"), (enc_df.get_conditioning(mel_df[:, 0]), "diff_latent_single")):
		ref = t(g[key])
		assert got.shape == ref.shape and got.dtype == torch.float32
		assert (got.cpu() - ref).abs().max().item() < 5e-4 * max(1.0, ref.abs().max().item()), key


@pytest.mark.parametrize("b,T", [(1, 1), (1, 2), (3, 5), (2, 517), (1, 400)])
def test_small_shapes_vs_oracle(b, T):
	"""ragged lengths incl. a single frame (stride-2 stem: T=1 -> 1 position, T=2 -> 1, T=5 -> 2) and the reference's clip lengths
	(132300 samples / 256 = 517 frames for the AR encoder, 102400 / 256 = 400 for the diffusion one, emb/mel.py:50-82)"""
	from tortoise_tts_amd.conditioning import ConditioningEncoder, ContextualEmbedder
	w_ar = W.synth_state_dict(W.ar_conditioning_shapes(W.AR_SMALL), 73)
	w_df = W.synth_state_dict(W.diffusion_conditioning_shapes(W.DIFF_SMALL), 74)
	g = torch.Generator().manual_seed(b * 1000 + T)
	mel_ar = torch.randn(b, 80, T, generator=g) * 2 - 4
	mel_df = torch.randn(b, 100, T, generator=g) * 2 - 4
	with torch.inference_mode():
		ra = CO.ar_get_conditioning(w_ar, mel_ar, W.AR_SMALL.heads)
		rd = CO.diffusion_get_conditioning(w_df, mel_df, W.DIFF_SMALL.num_heads)
	ga = ConditioningEncoder(w_ar, W.AR_SMALL, dtype="f32", device=DEV).get_conditioning(mel_ar.to(DEV))
	gd = ContextualEmbedder(w_df, W.DIFF_SMALL, dtype="f32", device=DEV).get_conditioning(mel_df.to(DEV))
	assert (ga.cpu() - ra).abs().max().item() < 5e-4 * max(1.0, ra.abs().max().item())
	assert (gd.cpu() - rd).abs().max().item() < 5e-4 * max(1.0, rd.abs().max().item())


def test_full_size_bf16_vs_oracle():
	"""full-size encoders at the reference's clip lengths, bf16 arithmetic on bf16-exact weights vs the f32 oracle on the same weights.
	Tolerance: relative L2 < 3e-2 (bf16 GEMM operands through 6 / 5 residual attention blocks), and repeatable bit for bit."""
	from tortoise_tts_amd.conditioning import ConditioningEncoder, ContextualEmbedder
	w_ar = W.synth_state_dict(W.ar_conditioning_shapes(W.AR_FULL), 75, bf16_exact=True)
	w_df = W.synth_state_dict(W.diffusion_conditioning_shapes(W.DIFF_FULL), 76, bf16_exact=True)
	g = torch.Generator().manual_seed(9)
	mel_ar = torch.randn(1, 2, 80, 517, generator=g) * 2 - 4
	mel_df = torch.randn(1, 2, 100, 400, generator=g) * 2 - 4
	with torch.inference_mode():
		ra = CO.ar_get_conditioning(w_ar, mel_ar, 16)
		rd = CO.diffusion_get_conditioning(w_df, mel_df, 16)
	ea = ConditioningEncoder(w_ar, W.AR_FULL, dtype="bf16", device=DEV)
	ed = ContextualEmbedder(w_df, W.DIFF_FULL, dtype="bf16", device=DEV)
	ga, gd = ea.get_conditioning(mel_ar.to(DEV)), ed.get_conditioning(mel_df.to(DEV))
	assert ga.shape == (1, 1024) and gd.shape == (1, 2048)
	assert torch.isfinite(ga).all() and torch.isfinite(gd).all()
	assert rel_l2(ga.cpu(), ra) < 3e-2, rel_l2(ga.cpu(), ra)
	assert rel_l2(gd.cpu(), rd) < 3e-2, rel_l2(gd.cpu(), rd)
	assert torch.equal(ga, ea.get_conditioning(mel_ar.to(DEV))) and torch.equal(gd, ed.get_conditioning(mel_df.to(DEV)))


def test_argument_errors():
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.conditioning import ConditioningEncoder, ContextualEmbedder
	w_ar = W.synth_state_dict(W.ar_conditioning_shapes(W.AR_SMALL), 73)
	enc = ConditioningEncoder(w_ar, W.AR_SMALL, dtype="f32", device=DEV)
	with pytest.raises(_lib.TTKError, match=r"\[b, 80, T\]"):
		enc(torch.zeros(1, 100, 8))
	with pytest.raises(_lib.TTKError, match="empty"):
		enc(torch.zeros(1, 80, 0))
	with pytest.raises(_lib.TTKError, match="lacks"):
		ContextualEmbedder(w_ar, W.DIFF_SMALL, dtype="f32", device=DEV)
	with pytest.raises(_lib.TTKError, match="bf16"):
		ConditioningEncoder(w_ar, W.AR_SMALL, dtype="fp8w", device=DEV)
