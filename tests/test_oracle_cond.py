"""The conditioning-encoder oracle (oracle/cond_oracle.py) against the reference's `get_conditioning` methods run in the build
container (tests/golden/cond_small.npz, cond_full.npz; unified_voice.py:535-542, diffusion.py:1477-1485)."""
import numpy as np
import pytest
import torch

import cond_oracle as CO
from tortoise_tts_amd import weights as W


def t(a):
	return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name,ar_cfg,diff_cfg", [("cond_small", W.AR_SMALL, W.DIFF_SMALL), ("cond_full", W.AR_FULL, W.DIFF_FULL)])
def test_get_conditioning(golden, name, ar_cfg, diff_cfg):
	g = golden(name)
	seed = int(g["seed"])
	w_ar = W.synth_state_dict(W.ar_conditioning_shapes(ar_cfg), seed)
	w_df = W.synth_state_dict(W.diffusion_conditioning_shapes(diff_cfg), seed + 1)
	mel_ar, mel_df = t(g["mel_ar"]), t(g["mel_diff"])
	with torch.inference_mode():
		a = CO.ar_get_conditioning(w_ar, mel_ar, ar_cfg.heads)
		a1 = CO.ar_get_conditioning(w_ar, mel_ar[:, 0], ar_cfg.heads)
		d = CO.diffusion_get_conditioning(w_df, mel_df, diff_cfg.num_heads)
		d1 = CO.diffusion_get_conditioning(w_df, mel_df[:, 0], diff_cfg.num_heads)
	assert a.shape == (mel_ar.shape[0], ar_cfg.model_dim) and d.shape == (mel_df.shape[0], 2 * diff_cfg.model_channels)
	for got, key in ((a, "ar_latent"), (a1, "ar_latent_single"), (d, "diff_latent"), (d1, "diff_latent_single")):
		ref = t(g[key])
		assert (got - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item()), key
	assert float((t(g["ar_latent"]) - t(g["ar_latent_single"])).abs().max()) > 1e-3     # the second clip matters
	if "diff_embed" in g:
		with torch.inference_mode():
			e = CO.contextual_embedder(w_df, mel_df[:, 0], diff_cfg.num_heads)
		assert e.shape == tuple(g["diff_embed"].shape) and (e - t(g["diff_embed"])).abs().max().item() < 2e-4


def test_shapes_and_groups():
	assert CO.normalization_groups(2048) == 32 and CO.normalization_groups(128) == 32 and CO.normalization_groups(48) == 16
	# parameter counts of the two sub-modules at full size (6 and 5 AttentionBlocks, no relative bias in the AR encoder)
	assert W.n_params(W.ar_conditioning_shapes(W.AR_FULL)) == 80 * 1024 + 1024 + 6 * (2 * 1024 + 3 * 1024 * 1024 + 3 * 1024 + 1024 * 1024 + 1024)
	assert W.n_params(W.diffusion_conditioning_shapes(W.DIFF_FULL)) == (1024 * 100 * 3 + 1024) + (2048 * 1024 * 3 + 2048) + 5 * (
		2 * 2048 + 3 * 2048 * 2048 + 3 * 2048 + 2048 * 2048 + 2048 + 32 * 16)
