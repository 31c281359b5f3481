"""The oracle (oracle/tortoise_oracle.py) against vectors produced by the reference itself
(oracle/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W


def t(a):
	return torch.from_numpy(np.asarray(a))


def close(a, b, atol, rtol=0.0):
	a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
	err = (a - b).abs().max().item()
	assert err <= atol + rtol * b.abs().max().item(), f"max abs err {err:.3e}"
	return err


@pytest.mark.parametrize("steps", [4, 30, 80, 200])
def test_schedule_tables_bit_exact(golden, steps):
	g = golden("schedule")
	s = O.SpacedSchedule(steps=steps)
	assert s.timestep_map == g[f"map_{steps}"].tolist()
	for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
				"posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
		assert np.array_equal(getattr(s, name), g[f"{name}_{steps}"]), name      # float64, bit for bit


def test_schedule_known_map():
	assert O.SpacedSchedule(steps=4).timestep_map == [0, 1333, 2666, 3999]


def _ar(golden, name, cfg):
	g = golden(name)
	w = W.synth_state_dict(W.ar_shapes(cfg), int(g["seed"]))
	return g, O.AROracle(w, cfg)


def test_ar_small_prefill_decode_latents(golden):
	g, ar = _ar(golden, "ar_small", W.AR_SMALL)
	B = int(g["B"])
	text, cond = t(g["text"]), t(g["cond"])
	with torch.inference_mode():
		prefix = ar.prefix_embeddings(cond, text)
		logits, past, _ = ar.prefill(prefix, B)
		close(logits[:, -1], g["prefill_logits"], 2e-5)
		toks = t(g["dec_tokens"])
		for k in range(1, toks.shape[1] + 1):
			lg, past, _ = ar.decode(toks[:, k - 1], k, past)
			close(lg, g["decode_logits"][:, k - 1], 2e-5)
		lat = ar.forward_latents(cond.repeat(B, 1), text.repeat(B, 1), t(g["codes"]))
		close(lat, g["latents"], 2e-5)


def test_dense_teacher_forced_pass_equals_cached_decode(golden):
	"""AROracle.teacher_forced_logits (one dense pass, used by the full-size GPU tests at ctx 318 / 750) against the reference's
	own KV-cached logits of the golden fixture and against the oracle's cached steps on a longer run."""
	g, ar = _ar(golden, "ar_small", W.AR_SMALL)
	text, cond, toks = t(g["text"]), t(g["cond"]), t(g["dec_tokens"])
	n = toks.shape[1]
	with torch.inference_mode():
		dense = ar.teacher_forced_logits(cond, text, toks, list(range(n + 1)))
		close(dense[:, 0], g["prefill_logits"], 2e-5)
		close(dense[:, 1:], g["decode_logits"], 2e-5)
		toks = torch.randint(0, 8192, (3, 40), generator=torch.Generator().manual_seed(3))
		steps = [0, 1, 17, 39, 40]
		dense = ar.teacher_forced_logits(cond, text, toks, steps)
		lg, past, _ = ar.prefill(ar.prefix_embeddings(cond, text), 3)
		cached = {0: lg[:, -1]}
		for k in range(1, 41):
			lg, past, _ = ar.decode(toks[:, k - 1], k, past)
			cached[k] = lg
		for i, j in enumerate(steps):
			close(dense[:, i], cached[j], 5e-5)


def test_ar_decode_position_quirk_matters(golden):
	"""k+1 indexing (unified_voice.py:214): using k instead must NOT reproduce the reference."""
	g, ar = _ar(golden, "ar_small", W.AR_SMALL)
	B = int(g["B"])
	with torch.inference_mode():
		prefix = ar.prefix_embeddings(t(g["cond"]), t(g["text"]))
		_, past, _ = ar.prefill(prefix, B)
		tok = t(g["dec_tokens"])[:, 0]
		emb = ar.w["mel_embedding.weight"][tok] + ar.w["mel_pos_embedding.emb.weight"][1]
		hidden, _ = O.gpt2_stack(ar.w, ar.cfg.layers, ar.cfg.heads, emb.unsqueeze(1), past)
		wrong = ar.lm_head(hidden)[:, -1]
	assert (wrong - t(g["decode_logits"][:, 0])).abs().max() > 1e-2


def test_diff_small(golden):
	g = golden("diff_small")
	cfg = W.DIFF_SMALL
	d = O.DiffusionOracle(W.synth_state_dict(W.diffusion_shapes(cfg), int(g["seed"])), cfg)
	T = int(g["T"])
	with torch.inference_mode():
		E = d.timestep_independent(t(g["latents"]), t(g["cond"]), T)
		close(E, g["E"], 2e-5)
		x, ts = t(g["x"]), t(g["t"])
		close(d.forward(x, ts, E), g["y_cond"], 5e-5)
		close(d.forward(x, ts, E, conditioning_free=True), g["y_uncond"], 5e-5)
		for sampler in ("ddim", "p"):
			for cf in (True, False):
				s = O.SpacedSchedule(steps=4, cond_free=cf)
				torch.manual_seed(int(g["sampler_seed"]))
				mel = s.sample_loop(d, t(g["noise"]), E[:1], sampler=sampler)
				close(mel, g[f"{sampler}_cf{int(cf)}"], 2e-4)


def test_rel_pos_bucket_edges():
	rel = torch.arange(-200, 201)
	b = O.rel_pos_bucket(rel)
	assert b.min() >= 0 and b.max() <= 31
	# rel = k - q; n = -rel; n < 0 (key after query) adds 16
	assert b[200] == 0 and b[201] == 17 and b[199] == 1          # rel 0, +1, -1
	assert O.rel_pos_bucket(torch.tensor([5])) == 16 + 5 and O.rel_pos_bucket(torch.tensor([-5])) == 5
	assert O.rel_pos_bucket(torch.tensor([-1000])) == 15 and O.rel_pos_bucket(torch.tensor([1000])) == 31


@pytest.mark.parametrize("name,cfgname", [("ar_full", "AR_FULL")])
def test_ar_full_slices(golden, name, cfgname):
	g, ar = _ar(golden, name, getattr(W, cfgname))
	cols = t(g["logit_cols"])
	B = int(g["B"])
	with torch.inference_mode():
		prefix = ar.prefix_embeddings(t(g["cond"]), t(g["text"]))
		logits, past, _ = ar.prefill(prefix, B)
		close(logits[:, -1][:, cols], g["prefill_logits"], 1e-4)
		toks = t(g["dec_tokens"])
		for k in range(1, toks.shape[1] + 1):
			lg, past, _ = ar.decode(toks[:, k - 1], k, past)
			close(lg[:, cols], g["decode_logits"][:, k - 1], 1e-4)
		lat = ar.forward_latents(t(g["cond"]).repeat(B, 1), t(g["text"]).repeat(B, 1), t(g["codes"]))
		close(lat[:, :, :128], g["latents"], 1e-4)


def test_diff_full(golden):
	g = golden("diff_full")
	cfg = W.DIFF_FULL
	d = O.DiffusionOracle(W.synth_state_dict(W.diffusion_shapes(cfg), int(g["seed"])), cfg)
	with torch.inference_mode():
		E = d.timestep_independent(t(g["latents"]), t(g["cond"]), int(g["T"]))
		close(E, g["E"], 1e-4)
		close(d.forward(t(g["x"]), t(g["t"]), E), g["y_cond"], 2e-4)
		close(d.forward(t(g["x"]), t(g["t"]), E, conditioning_free=True), g["y_uncond"], 2e-4)
