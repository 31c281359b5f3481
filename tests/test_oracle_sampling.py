"""The restated generate() loop pieces (a5).  The reference's own loop cannot run on transformers 5.15
(SURVEY.md 8c), so the warpers are pinned against the installed HF classes the reference instantiates
(stream_generator.py:80-85) and the loop's control flow is checked by its stated invariants."""
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
