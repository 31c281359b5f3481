"""The CLVP oracle (oracle/clvp_oracle.py) against the reference model run in the build container (tests/golden/clvp_small.npz)."""
import numpy as np
import torch

import clvp_oracle as CO
from tortoise_tts_amd import weights as W


def t(a):
	return torch.from_numpy(np.asarray(a))


def test_state_dict_names_are_the_reference_ones(golden):
	g = golden("clvp_small")
	assert {str(k) for k in g["keys"]} == set(W.clvp_shapes(W.CLVP_SMALL))
	assert W.n_params(W.clvp_shapes(W.CLVP_FULL)) == 243_846_145


def test_encoder_and_scores(golden):
	g = golden("clvp_small")
	cfg = W.CLVP_SMALL
	sd = W.synth_state_dict(W.clvp_shapes(cfg), int(g["seed"]))
	orc = CO.CLVPOracle(sd, cfg)
	text, codes = t(g["text"]), t(g["codes"])
	with torch.inference_mode():
		scores = orc.forward(text.repeat(codes.shape[0], 1), codes)
	assert scores.shape == (codes.shape[0],)
	assert (scores - t(g["scores"])).abs().max().item() < 2e-5
	assert float(t(g["scores"]).std()) > 1e-3                    # candidates are told apart
	# rotary really is applied to the values too (xtransformers.py:622-626): without it the scores move
	saved = CO.apply_rotary
	try:
		calls = {"n": 0}
		def only_qk(tt, f):
			calls["n"] += 1
			return saved(tt, f) if calls["n"] % 3 != 0 else tt
		CO.apply_rotary = only_qk
		with torch.inference_mode():
			wrong = orc.forward(text.repeat(codes.shape[0], 1), codes)
	finally:
		CO.apply_rotary = saved
	assert (wrong - t(g["scores"])).abs().max().item() > 1e-4
