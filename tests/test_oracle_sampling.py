"""The restated generate() loop (a5).  The reference's own fork of the loop cannot run on transformers 5.15, so
it is pinned against what it forks: the warpers against the installed HF classes the reference instantiates
(stream_generator.py:80-85), and the loop itself -- processor order, softmax + multinomial and its generator consumption,
pad-after-EOS, stopping, max_length, GenerationConfig defaults -- against the installed `GenerationMixin.generate` driving a
model-free stub (oracle/stub_lm.py): ids stored in tests/golden/hf_sample_loop.npz and, when transformers is importable, live."""
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W


def test_warpers_match_hf_classes():
	lp = pytest.importorskip("transformers.generation.logits_process")
	g = torch.Generator().manual_seed(5)
	scores = torch.randn(4, 8194, generator=g) * 3
	ids = torch.randint(0, 8194, (4, 20), generator=g)
	assert torch.equal(O.warp_temperature(scores, 0.8), lp.TemperatureLogitsWarper(0.8)(ids, scores))
	assert torch.equal(O.warp_top_k(scores, 16), lp.TopKLogitsWarper(top_k=16, min_tokens_to_keep=1)(ids, scores))
	assert torch.equal(O.warp_top_p(scores, 0.7), lp.TopPLogitsWarper(top_p=0.7, min_tokens_to_keep=1)(ids, scores))
	assert torch.equal(O.warp_repetition_penalty(ids, scores, 2.0), lp.RepetitionPenaltyLogitsProcessor(2.0)(ids, scores.clone()))
	assert torch.equal(O.warp_suppress(scores, [8193]), lp.SuppressTokensLogitsProcessor([8193])(ids, scores))


def _ar():
	cfg = W.AR_SMALL
	return O.AROracle(W.synth_state_dict(W.ar_shapes(cfg), 11), cfg)


def test_loop_is_seeded_and_fixed_length_with_suppressed_stop():
	ar = _ar()
	text = torch.randint(1, 255, (1, 9), generator=torch.Generator().manual_seed(1))
	cond = torch.randn(1, 128, generator=torch.Generator().manual_seed(2))
	with torch.inference_mode():
		a = O.inference_speech(ar, cond, text, num_return_sequences=3, max_generate_length=7, temperature=0.8, suppress_tokens=[8193])
		torch.manual_seed(1234)                               # the loop reseeds with 0 itself (stream_generator.py:296)
		b = O.inference_speech(ar, cond, text, num_return_sequences=3, max_generate_length=7, temperature=0.8, suppress_tokens=[8193])
	assert a.shape == (3, 7) and torch.equal(a, b)
	assert (a != 8193).all() and len({tuple(r.tolist()) for r in a}) == 3      # candidates differ


def test_loop_pads_after_eos_and_stops_when_all_finished():
	ar = _ar()
	# make EOS overwhelmingly likely after the first step by biasing the head
	ar.w = dict(ar.w)
	ar.w["mel_head.bias"] = ar.w["mel_head.bias"].clone()
	ar.w["mel_head.bias"][8193] = 50.0
	text = torch.randint(1, 255, (1, 5), generator=torch.Generator().manual_seed(3))
	cond = torch.randn(1, 128, generator=torch.Generator().manual_seed(4))
	with torch.inference_mode():
		out = O.inference_speech(ar, cond, text, num_return_sequences=2, max_generate_length=20)
	assert out.shape[1] == 1 and (out == 8193).all()


def test_stop_and_calm_postprocessing():
	codes = torch.tensor([[5, 6, 8193, 8193, 8193, 8193, 8193], [1, 2, 3, 4, 5, 6, 7]])
	fixed = O.fix_stop_tokens(codes, 8193)
	assert fixed[0].tolist() == [5, 6, 83, 83, 45, 45, 248]
	assert fixed[1].tolist() == [1, 2, 3, 4, 5, 6, 7]            # no stop token: untouched (reference would raise, inference.py:355)
	c = torch.tensor([[1] + [83] * 12])
	lat = torch.zeros(1, 13, 4)
	assert O.trim_calm_tokens(c, lat).shape[1] == 9


def _stub_ids(name):
	import stub_lm
	name_, seed, bias, B, N, kw = next(c for c in stub_lm.CASES if c[0] == name)
	ar = stub_lm.StubAR(W.AR_SMALL, stub_lm.make_table(seed, bias))
	with torch.inference_mode():
		ids = O.inference_speech(ar, torch.zeros(1, 1), torch.zeros(1, stub_lm.PREFIX - 3, dtype=torch.long), num_return_sequences=B,
								 max_generate_length=N, **kw)
	return ids, (seed, bias, B, N, kw)


@pytest.mark.parametrize("name", ["plain", "stops_early", "all_warpers", "suppress", "top_p_only", "hf_defaults", "sixteen_candidates"])
def test_loop_equals_huggingface_generate_on_stub_model(golden, name):
	import json
	import stub_lm
	g = golden("hf_sample_loop")
	ids, (seed, bias, B, N, kw) = _stub_ids(name)
	assert json.loads(str(g["kw::" + name])) == json.loads(json.dumps(kw))          # the fixture was made with these arguments
	want = torch.from_numpy(g["ids::" + name])
	assert ids.shape == want.shape and torch.equal(ids, want)
	if name == "stops_early":
		assert want.shape[1] < N and (want == stub_lm.STOP).any(dim=1).all()          # the case does stop early, on different steps
		assert len({int((r == stub_lm.STOP).float().argmax()) for r in want}) > 1
	if name == "suppress":
		assert not (want == 5).any() and not (want == 17).any() and not (want == stub_lm.STOP).any()
	# live, against the transformers installed next to this test (same loop the fixture came from)
	pytest.importorskip("transformers")
	live = stub_lm.hf_generate(stub_lm.make_table(seed, bias), B, N, kw)
	assert torch.equal(live, want)


def test_omitted_top_k_means_generation_config_default():
	"""`hf_defaults` passes nothing: HF applies top_k = 50; the same call with top_k=0 must differ (so the default is not a no-op)."""
	a, _ = _stub_ids("hf_defaults")
	import stub_lm
	ar = stub_lm.StubAR(W.AR_SMALL, stub_lm.make_table(6, 5.0))
	with torch.inference_mode():
		b = O.inference_speech(ar, torch.zeros(1, 1), torch.zeros(1, stub_lm.PREFIX - 3, dtype=torch.long), num_return_sequences=3, max_generate_length=25, top_k=0)
		c = O.inference_speech(ar, torch.zeros(1, 1), torch.zeros(1, stub_lm.PREFIX - 3, dtype=torch.long), num_return_sequences=3, max_generate_length=25, top_k=50)
	assert torch.equal(a, c) and not (a.shape == b.shape and torch.equal(a, b))


def test_setup_seed_equals_reference(golden):
	"""the product's `setup_seed` (sampling.py) and the oracle's seeding leave the generators where the reference's setup_seed does"""
	import random

	import numpy as np

	from tortoise_tts_amd.sampling import setup_seed
	g = golden("wrapper")
	for seed in (0, 7):
		setup_seed(seed)
		assert np.array_equal(torch.rand(4).numpy(), g[f"torch_{seed}"])
		assert np.array_equal(np.random.rand(4), g[f"numpy_{seed}"])
		assert np.array_equal(np.array([random.random() for _ in range(4)]), g[f"python_{seed}"])
	torch.manual_seed(123)
	setup_seed(-1)
	assert np.array_equal(torch.rand(4).numpy(), g["torch_keep"])
	assert g["fake_ids"].shape == (2, 5 + 4) and (g["fake_ids"][:, :-1] == 1).all() and (g["fake_ids"][:, -1] == W.AR_SMALL.start_mel_token).all()


def test_sample_stream_equals_the_references_own_loop(golden):
	"""a5/a6 pinned by the reference's own loop code: `NewGenerationMixin.sample_stream` (stream_generator.py:911-1190) driven on the
	reference's GPT2InferenceModel (oracle/make_golden.py: sample_stream_case) -- yielded tokens AND latents of every case; and
	`inference_speech`, which restates `generate()`'s sample branch over the same loop body, must give those ids too."""
	import json
	g = golden("sample_stream")
	names = sorted({k.split("::")[0] for k in g})
	assert len(names) >= 6
	cfg = W.AR_SMALL
	for name in names:
		meta = json.loads(str(g[f"{name}::meta"]))
		sd = W.synth_state_dict(W.ar_shapes(cfg), meta["weight_seed"])
		if meta["stop_bias"]:
			sd["mel_head.bias"] = sd["mel_head.bias"].clone()
			sd["mel_head.bias"][cfg.stop_mel_token] += meta["stop_bias"]
		ar = O.AROracle(sd, cfg)
		text, cond = torch.from_numpy(g[f"{name}::text"]), torch.from_numpy(g[f"{name}::cond"])
		kw = dict(meta["kw"])
		kw.setdefault("top_k", 0)
		want_ids, want_lat = torch.from_numpy(g[f"{name}::ids"]), torch.from_numpy(g[f"{name}::latents"])
		if kw.pop("prompt_len", None):
			# a prompted continuation: the reference's loop was given ids = [fake prefix | one prompt per row].  The oracle's loop takes the same rows;
			# `inference_speech(input_tokens=)` builds its rows as the reference's wrapper does (tiled, then expanded by generate: nrs ** 2 of them) and must equal
			# the oracle's loop on exactly those rows, prompt columns in front (unified_voice.py:651-668)
			prompt = torch.from_numpy(g[f"{name}::prompt"])
			with torch.inference_mode():
				pairs = list(O.sample_stream(ar, cond, text, num_return_sequences=meta["B"], max_generate_length=meta["max_new"], prompt=prompt, **kw))
				nrs = prompt.shape[0]
				ids = O.inference_speech(ar, cond, text, num_return_sequences=nrs, max_generate_length=meta["max_new"], input_tokens=prompt, **kw)
				rows = prompt.repeat_interleave(nrs, 0)
				again = list(O.sample_stream(ar, cond, text, num_return_sequences=nrs * nrs, max_generate_length=meta["max_new"], prompt=rows, **kw))
			assert len(pairs) == want_ids.shape[1] == meta["max_new"] - prompt.shape[1], name       # "max new" counts the prompt tokens (:660)
			assert torch.equal(torch.stack([t for t, _ in pairs], 1), want_ids), name
			assert (torch.stack([l for _, l in pairs], 1) - want_lat).abs().max() < 2e-5, name
			assert ids.shape == (nrs * nrs, meta["max_new"]) and torch.equal(ids[:, :prompt.shape[1]], rows), name
			assert torch.equal(ids[:, prompt.shape[1]:], torch.stack([t for t, _ in again], 1)), name
			continue
		with torch.inference_mode():
			pairs = list(O.sample_stream(ar, cond, text, num_return_sequences=meta["B"], max_generate_length=meta["max_new"], **kw))
			ids = O.inference_speech(ar, cond, text, num_return_sequences=meta["B"], max_generate_length=meta["max_new"], **kw)
		assert len(pairs) == want_ids.shape[1], name                                 # every token is yielded, the last one too
		assert torch.equal(torch.stack([t for t, _ in pairs], 1), want_ids), name
		assert (torch.stack([l for _, l in pairs], 1) - want_lat).abs().max() < 2e-5, name
		assert torch.equal(ids, want_ids), name
