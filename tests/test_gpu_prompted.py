"""`inference_speech(..., input_tokens=)`: the prompted continuation of unified_voice.py:651-668 (VERDICT r03 missing #5; `TTS.inference` never passes it).
The oracle's version is pinned by the reference's own sample_stream run on prompted rows (tests/golden/sample_stream.npz "prompted", tests/test_oracle_sampling.py);
here the product's ids must equal the oracle's bit for bit in f32 -- prompt tokens in front, counted in max_generate_length, num_return_sequences ** 2 rows as the
reference's wrapper produces them -- and the generator must stand where the oracle's stands afterwards.  GPU only; calls go through the C ABI."""
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import _lib
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
	return torch.Generator().manual_seed(seed)


def build(sd, dtype="f32", **kw):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	return UnifiedVoice(sd, W.AR_SMALL, dtype=dtype, device=DEV, max_batch=16, max_ctx=96, **kw)


@pytest.mark.parametrize("rows,nrs,kw,graph", [
	(1, 3, dict(temperature=0.8, top_k=0), True),                                              # one prompt for all: it joins the shared prefix
	(2, 4, dict(temperature=0.9, top_k=20, top_p=0.9, repetition_penalty=2.0), True),          # two prompts, 16 rows, the penalty sees the prompt tokens
	(2, 2, dict(temperature=0.8, top_k=0, suppress_tokens=[8193]), False),                     # eager loop
])
def test_prompted_continuation_equals_the_oracle(rows, nrs, kw, graph):
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 5)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][cfg.stop_mel_token] += 3.0                                            # rows stop at different steps
	text = torch.randint(1, 255, (1, 9), generator=gen(1))
	cond = torch.randn(1, cfg.model_dim, generator=gen(2))
	prompt = torch.randint(0, 8192, (rows, 6), generator=gen(3))
	model = build(sd, use_graph=graph)
	with torch.inference_mode():
		want = O.inference_speech(O.AROracle(sd, cfg), cond, text, num_return_sequences=nrs, max_generate_length=20, input_tokens=prompt, sample_device="cuda", **kw)
		after_want = torch.rand(3, device=DEV)
		got = model.inference_speech(cond.to(DEV), text.to(DEV), input_tokens=prompt.to(DEV), num_return_sequences=nrs, max_generate_length=20, do_sample=True, **kw)
		after_got = torch.rand(3, device=DEV)
	assert got.shape[0] == nrs * nrs and torch.equal(got[:, :6].cpu(), prompt.repeat(nrs // rows, 1).repeat_interleave(nrs, 0))
	assert got.shape == want.shape and torch.equal(got.cpu(), want), (got.shape, want.shape)
	assert torch.equal(after_got, after_want)                                                # the draws that follow are the reference's stream
	# the plain call on the same model afterwards is untouched by the prompt state (columns, history, noise offset are per call)
	with torch.inference_mode():
		a = model.inference_speech(cond.to(DEV), text.to(DEV), num_return_sequences=nrs * nrs, max_generate_length=20, do_sample=True, **kw)
		b = O.inference_speech(O.AROracle(sd, cfg), cond, text, num_return_sequences=nrs * nrs, max_generate_length=20, sample_device="cuda", **kw)
	assert torch.equal(a.cpu(), b)


def test_prompted_continuation_bf16_runs_and_rejects_bad_prompts():
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 5)
	model = build(sd, "bf16")
	text = torch.randint(1, 255, (1, 9), generator=gen(1)).to(DEV)
	cond = torch.randn(1, cfg.model_dim, generator=gen(2)).to(DEV)
	prompt = torch.randint(0, 8192, (1, 4), generator=gen(3)).to(DEV)
	kw = dict(do_sample=True, temperature=0.8, top_k=0, suppress_tokens=[8193])
	with torch.inference_mode():
		a = model.inference_speech(cond, text, input_tokens=prompt, num_return_sequences=2, max_generate_length=12, **kw)
		b = model.inference_speech(cond, text, input_tokens=prompt, num_return_sequences=2, max_generate_length=12, **kw)
	assert a.shape == (4, 12) and torch.equal(a, b) and torch.equal(a[:, :4], prompt.expand(4, -1))
	with pytest.raises(ValueError):
		model.inference_speech(cond, text, input_tokens=torch.zeros((3, 2), dtype=torch.long), num_return_sequences=4, **kw)          # 4 % 3 != 0
	with pytest.raises(ValueError):
		model.inference_speech(cond, text, input_tokens=torch.full((1, 2), 8193), num_return_sequences=1, max_generate_length=8, **kw)   # a stop token inside the prompt
	with pytest.raises(ValueError):
		model.inference_speech(cond, text, input_tokens=prompt, num_return_sequences=1, max_generate_length=4, **kw)                  # no room left
	with pytest.raises(IndexError):
		model.inference_speech(cond, text, input_tokens=torch.full((1, 2), 9000), num_return_sequences=1, max_generate_length=8, **kw)
	with pytest.raises(_lib.TTKError):
		model.inference_speech(cond, text, input_tokens=prompt, num_return_sequences=5, max_generate_length=12, **kw)                 # 25 rows > max_batch
