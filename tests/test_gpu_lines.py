"""Several text lines sampled as ONE decode batch (UnifiedVoice.inference_speech_lines / ttk_ar_prefill_lines): every line's ids, and the
generator position after it, equal the per-line `inference_speech` call bit for bit -- lines of different length, rows that stop at different
steps, warpers on, one and two MFMA row tiles, graph and eager, head-drawn and torch-drawn noise.  GPU only; calls go through the C ABI."""
import pytest
import torch

from tortoise_tts_amd import _lib
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _model(cfg, dtype, max_batch, max_ctx, stop_bias=4.0, **kw):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	sd = W.synth_state_dict(W.ar_shapes(cfg), 31)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][cfg.stop_mel_token] += stop_bias
	return UnifiedVoice(sd, cfg, dtype=dtype, device=DEV, max_batch=max_batch, max_ctx=max_ctx, **kw)


def _texts(lens, seed=5):
	g = torch.Generator().manual_seed(seed)
	return [torch.randint(1, 255, (1, n), generator=g).to(DEV) for n in lens]


def _check(ar, al, texts, C, kw):
	with torch.inference_mode():
		want, after = [], []
		for t in texts:
			want.append(ar.inference_speech(al, t, num_return_sequences=C, **kw))
			after.append(torch.rand(3, device=DEV))
		got = ar.inference_speech_lines(al, texts, num_return_sequences=C, **kw)
		assert len(got) == len(texts)
		for g, (a, b) in enumerate(zip(got, want)):
			assert a.shape == b.shape and torch.equal(a, b), (g, a.shape, b.shape)
			ar.position_rng_after_line(g)
			assert torch.equal(torch.rand(3, device=DEV), after[g]), g
	return want


@pytest.mark.parametrize("dtype,own,graph", [("f32", "1", True), ("bf16", "1", True), ("bf16", "0", True), ("f32", "1", False)])
def test_lines_of_different_length_equal_the_per_line_calls(dtype, own, graph, monkeypatch):
	monkeypatch.setenv("TTK_AR_OWN_RNG", own)
	cfg = W.AR_SMALL
	ar = _model(cfg, dtype, max_batch=16, max_ctx=160, use_graph=graph)
	al = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(9)).to(DEV)
	kw = dict(do_sample=True, temperature=0.8, top_k=16, top_p=0.9, repetition_penalty=2.0, max_generate_length=60)
	want = _check(ar, al, _texts([21, 3, 40]), 5, kw)
	assert len({w.shape[1] for w in want}) > 1          # the lines really end at different steps
	_check(ar, al, _texts([7, 7], seed=6), 8, kw)         # equal lengths, a full 16-row tile, the captured step of another state


def test_two_row_tiles_and_a_line_that_runs_to_the_limit():
	"""32 rows = two MFMA row tiles per launch (the per-line call has one): same bits; one line cannot stop (its stop logit is suppressed by
	the other lines' loop end only), so the batch runs to max_generate_length while the others are cut where they finished"""
	cfg = W.AR_SMALL
	ar = _model(cfg, "bf16", max_batch=32, max_ctx=128, stop_bias=6.0)
	al = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(10)).to(DEV)
	kw = dict(do_sample=True, temperature=0.9, top_k=0, max_generate_length=24)
	_check(ar, al, _texts([12, 30], seed=7), 16, kw)
	ar2 = _model(cfg, "bf16", max_batch=32, max_ctx=128, stop_bias=-30.0)      # no row ever stops: every line is max_generate_length long
	want = _check(ar2, al, _texts([5, 9, 2, 14], seed=8), 8, kw)
	assert all(w.shape[1] == 24 for w in want)


def test_full_size_two_lines_of_sixteen():
	"""the benchmark's models and candidate count: 2 x 16 candidates (two row tiles, K = 1024 / 4096 GEMVs, split-K combine) vs the per-line calls"""
	cfg = W.AR_FULL
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(cfg), 0), cfg, dtype="bf16", device=DEV, max_batch=32, max_ctx=64 + 4 + 40 + 8)
	al = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(11)).to(DEV)
	kw = dict(do_sample=True, temperature=0.8, top_k=0, max_generate_length=40, suppress_tokens=[cfg.stop_mel_token])
	_check(ar, al, _texts([64, 37], seed=12), 16, kw)


def test_argument_errors():
	cfg = W.AR_SMALL
	ar = _model(cfg, "f32", max_batch=8, max_ctx=64)
	al = torch.randn(1, cfg.model_dim).to(DEV)
	kw = dict(do_sample=True, max_generate_length=8)
	with pytest.raises(_lib.TTKError, match="max_batch"):
		ar.inference_speech_lines(al, _texts([3, 4, 5]), num_return_sequences=3, **kw)
	with pytest.raises(_lib.TTKError, match="max_ctx"):
		ar.inference_speech_lines(al, _texts([3, 55]), num_return_sequences=2, **kw)
	with pytest.raises(IndexError):
		ar.inference_speech_lines(al, [torch.full((1, 4), 300, device=DEV), _texts([3])[0]], num_return_sequences=2, **kw)


def test_typical_sampling_in_line_batches_and_against_the_torch_op_form():
	"""typical sampling runs inside the sampling kernel since round 3: a line batch equals the per-line calls (as for every other warper), and both
	equal a model built with hf_exact_top_p=True, which runs the reference's TypicalLogitsWarper as torch ops in front of the kernel"""
	cfg = W.AR_SMALL
	ar = _model(cfg, "f32", max_batch=8, max_ctx=96)
	al = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(3)).to(DEV)
	kw = dict(do_sample=True, temperature=0.9, top_k=0, max_generate_length=12, typical_sampling=True, typical_mass=0.8)
	_check(ar, al, _texts([5, 8], seed=2), 3, kw)
	exact = _model(cfg, "f32", max_batch=8, max_ctx=96, hf_exact_top_p=True)
	same = ar
	t = _texts([7], seed=5)[0]
	with torch.inference_mode():
		a = same.inference_speech(al, t, num_return_sequences=4, **kw)
		off_a = torch.cuda.default_generators[0].get_offset()
		b = exact.inference_speech(al, t, num_return_sequences=4, **kw)
		assert torch.equal(a, b) and torch.cuda.default_generators[0].get_offset() == off_a
	with pytest.raises(NotImplementedError):
		ar.inference_speech_lines(al, _texts([5, 8]), num_return_sequences=2, do_sample=True, input_tokens=torch.zeros(1, 2))


@pytest.mark.parametrize("penalty,graph", [(1.0, True), (2.0, True), (2.0, False)])
def test_hf_exact_top_p_line_batches_equal_the_per_line_calls(penalty, graph):
	"""hf_exact_top_p=True with top_p < 1 sends a LINE BATCH through the torch-op warper chain in front of the kernel (in_kernel=False, lines > 1):
	graphable without a repetition penalty (no growing history slice), eager with one.  Each line must still equal its own call, generator position included,
	and -- away from f32 cumsum ties -- the in-kernel cut of a default model"""
	cfg = W.AR_SMALL
	exact = _model(cfg, "f32", max_batch=12, max_ctx=128, hf_exact_top_p=True, use_graph=graph)
	al = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(13)).to(DEV)
	kw = dict(do_sample=True, temperature=0.8, top_k=24, top_p=0.85, repetition_penalty=penalty, max_generate_length=30)
	texts = _texts([11, 4, 26], seed=14)
	want = _check(exact, al, texts, 4, kw)
	st = next(reversed(exact._states.values()))
	assert st.lines == 3 and not st.in_kernel and st.graphable == (penalty == 1.0)
	plain = _model(cfg, "f32", max_batch=12, max_ctx=128, use_graph=graph)
	with torch.inference_mode():
		got = plain.inference_speech_lines(al, texts, num_return_sequences=4, **kw)
	assert all(torch.equal(a, b) for a, b in zip(got, want))
