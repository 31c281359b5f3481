"""CLVP scoring on libttk against the reference model's scores (tests/golden/clvp_small.npz) and the oracle.
GPU only; calls go through the C ABI (`ttk_clvp_*`)."""
import numpy as np
import pytest
import torch

import clvp_oracle as CO
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
	return torch.from_numpy(np.asarray(a))


def test_small_scores_equal_reference(golden):
	from tortoise_tts_amd.clvp import CLVP
	g = golden("clvp_small")
	cfg = W.CLVP_SMALL
	sd = W.synth_state_dict(W.clvp_shapes(cfg), int(g["seed"]))
	text, codes = t(g["text"]), t(g["codes"])
	ref = t(g["scores"])
	m32 = CLVP(sd, cfg, dtype="f32", device=DEV)
	s32 = m32(text.repeat(codes.shape[0], 1).to(DEV), codes.to(DEV), return_loss=False)
	assert s32.shape == ref.shape and (s32.cpu() - ref).abs().max().item() < 1e-4            # fp32 mode vs the REFERENCE class
	assert torch.equal(m32(text.to(DEV), codes.to(DEV)), s32)                                 # [1, Tt] text == the repeated form
	mb = CLVP(sd, cfg, dtype="bf16", device=DEV)
	sb = mb(text.to(DEV), codes.to(DEV))
	assert (sb.cpu() - ref).abs().max().item() < 3e-2                                         # similarity * e^temperature, |.| <~ 1.3
	with pytest.raises(NotImplementedError):
		m32(text.to(DEV), codes.to(DEV), return_loss=True)
	with pytest.raises(Exception, match="outside the embedding table"):
		m32(text.to(DEV), torch.full((2, 4), cfg.num_speech_tokens))


@pytest.mark.parametrize("B,Tt,M", [(1, 1, 1), (3, 70, 130), (16, 5, 64)])
def test_small_shapes_vs_oracle(B, Tt, M):
	from tortoise_tts_amd.clvp import CLVP
	cfg = W.CLVP_SMALL
	sd = W.synth_state_dict(W.clvp_shapes(cfg), 62)
	g = torch.Generator().manual_seed(B * 1000 + M)
	text = torch.randint(0, cfg.num_text_tokens, (B, Tt), generator=g)          # a different text per row (Bt == B branch)
	codes = torch.randint(0, cfg.num_speech_tokens, (B, M), generator=g)
	with torch.inference_mode():
		ref = CO.CLVPOracle(sd, cfg).forward(text, codes)
	got = CLVP(sd, cfg, dtype="f32", device=DEV)(text.to(DEV), codes.to(DEV))
	assert (got.cpu() - ref).abs().max().item() < 1e-4


def test_full_size_determinism_and_ranking():
	"""the 244 M-parameter configuration at the benchmark's shape (64 text tokens, 16 candidates x 250 codes), bf16: finite, repeatable,
	and it ranks candidates like its own fp32 mode up to near-ties"""
	from tortoise_tts_amd.clvp import CLVP
	cfg = W.CLVP_FULL
	sd = W.synth_state_dict(W.clvp_shapes(cfg), 63)
	g = torch.Generator().manual_seed(5)
	text = torch.randint(1, 255, (1, 64), generator=g).to(DEV)
	codes = torch.randint(0, 8192, (16, 250), generator=g).to(DEV)
	mb = CLVP(sd, cfg, dtype="bf16", device=DEV)
	a, b = mb(text, codes), mb(text, codes)
	assert a.shape == (16,) and torch.isfinite(a).all() and torch.equal(a, b)
	del mb
	f = CLVP(sd, cfg, dtype="f32", device=DEV)(text, codes)
	assert (a - f).abs().max().item() < 5e-2
