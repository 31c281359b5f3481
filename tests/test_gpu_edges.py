"""GPU parity on the edges the domain has: ragged/odd lengths, a single candidate, many candidates, the streaming generator,
and size-independent properties at BASELINE sizes (full-size models)."""
import numpy as np
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
	return torch.Generator().manual_seed(seed)


def maxerr(a, b):
	return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


@pytest.fixture(scope="module")
def small_ar():
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	sd = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 31)
	return UnifiedVoice(sd, W.AR_SMALL, dtype="f32", device=DEV, max_batch=32, max_ctx=160), O.AROracle(sd, W.AR_SMALL)


@pytest.fixture(scope="module")
def small_diff():
	from tortoise_tts_amd.diffusion import DiffusionTTS
	sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32)
	return DiffusionTTS(sd, W.DIFF_SMALL, dtype="f32", device=DEV), O.DiffusionOracle(sd, W.DIFF_SMALL)


@pytest.mark.parametrize("B,Tt,M", [(1, 1, 1), (3, 5, 7), (17, 11, 33), (32, 2, 70)])
def test_latent_pass_ragged_shapes(small_ar, B, Tt, M):
	"""Sequence lengths that are not multiples of any tile (S = Tt + M + 5), one to 32 candidates."""
	model, oracle = small_ar
	text = torch.randint(1, 255, (B, Tt), generator=gen(B))
	cond = torch.randn(B, 128, generator=gen(B + 1))
	codes = torch.randint(0, 8192, (B, M), generator=gen(B + 2))
	got = model.forward(cond.to(DEV), text.to(DEV), torch.tensor([Tt] * B), codes.to(DEV), torch.tensor([M * 1024] * B), return_latent=True, clip_inputs=False)
	with torch.inference_mode():
		ref = oracle.forward_latents(cond, text, codes)
	assert got.shape == (B, M, 128) and maxerr(got, ref) < 2e-4


def test_latent_pass_applies_mel_padding(small_ar):
	"""set_mel_padding (unified_voice.py:494-506): codes past wav_lengths // 1024 + 1 become the stop token."""
	model, oracle = small_ar
	text = torch.randint(1, 255, (2, 4), generator=gen(5))
	cond = torch.randn(2, 128, generator=gen(6))
	codes = torch.randint(0, 8192, (2, 12), generator=gen(7))
	got = model.forward(cond.to(DEV), text.to(DEV), torch.tensor([4, 4]), codes.to(DEV), torch.tensor([5 * 1024, 12 * 1024]), return_latent=True, clip_inputs=False)
	padded = codes.clone()
	padded[0, 6:] = 8193
	with torch.inference_mode():
		ref = oracle.forward_latents(cond, text, padded)
	assert maxerr(got, ref) < 2e-4


@pytest.mark.parametrize("B", [1, 5, 16, 32])
def test_decode_batch_sizes_teacher_forced(small_ar, B):
	"""KV-cached decode at several candidate counts (MFMA M-tiles of 16), logits vs the oracle for 5 forced tokens."""
	model, oracle = small_ar
	text = torch.randint(1, 255, (1, 6), generator=gen(B))
	cond = torch.randn(1, 128, generator=gen(B + 9))
	toks = torch.randint(0, 8192, (B, 5), generator=gen(B + 10))
	logits = model._prefill(cond.to(DEV), text.to(DEV), B)
	with torch.inference_mode():
		ref, past, _ = oracle.prefill(oracle.prefix_embeddings(cond, text), B)
		assert maxerr(logits, ref[:, -1]) < 2e-4
		for k in range(1, 6):
			model._decode(toks[:, k - 1].contiguous().to(DEV), logits)
			ref, past, _ = oracle.decode(toks[:, k - 1], k, past)
			assert maxerr(logits, ref) < 2e-4, k


@pytest.mark.parametrize("kw", [dict(temperature=0.8, suppress_tokens=[8193]), dict(temperature=0.7, top_k=16, top_p=0.9, repetition_penalty=2.0)])
def test_streaming_generator_matches_oracle_tokens_and_latents(small_ar, kw):
	"""a6: get_generator yields (codes, final_norm(hidden)) per token with the semantics the reference's own `sample_stream` has
	(stream_generator.py:1172; pinned in tests/test_oracle_sampling.py against tests/golden/sample_stream.npz): the latent of the
	forward the token was sampled from, every token yielded."""
	model, oracle = small_ar
	text = torch.randint(1, 255, (1, 6), generator=gen(40))
	cond = torch.randn(1, 128, generator=gen(41))
	ids = model.compute_embeddings(cond.to(DEV), text.to(DEV))
	out = list(model.get_generator(inputs=ids, max_length=ids.shape[1] + 8, do_sample=True, num_return_sequences=2, **kw))
	with torch.inference_mode():
		ref = list(O.sample_stream(oracle, cond, text, num_return_sequences=2, max_generate_length=8, sample_device="cuda", **kw))
	assert len(out) == len(ref) == 8
	for (tok, lat), (rtok, rlat) in zip(out, ref):
		assert torch.equal(tok.cpu(), rtok) and maxerr(lat, rlat) < 2e-4


def test_streaming_generator_stops_with_the_last_row(small_ar):
	"""rows that finish yield the pad token; the stream ends right after the token with which the last row finishes"""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 31)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][cfg.stop_mel_token] += 7.0
	model, oracle = UnifiedVoice(sd, cfg, dtype="f32", device=DEV, max_batch=4, max_ctx=128), O.AROracle(sd, cfg)
	text = torch.randint(1, 255, (1, 7), generator=gen(42))
	cond = torch.randn(1, 128, generator=gen(43))
	ids = model.compute_embeddings(cond.to(DEV), text.to(DEV))
	out = list(model.get_generator(inputs=ids, max_length=ids.shape[1] + 80, temperature=0.8, top_k=0, do_sample=True, num_return_sequences=3))
	with torch.inference_mode():
		ref = list(O.sample_stream(oracle, cond, text, num_return_sequences=3, max_generate_length=80, temperature=0.8, top_k=0, sample_device="cuda"))
	assert 1 <= len(out) == len(ref) < 80
	for (tok, lat), (rtok, rlat) in zip(out, ref):
		assert torch.equal(tok.cpu(), rtok) and maxerr(lat, rlat) < 2e-4
	assert bool((out[-1][0] == cfg.stop_mel_token).all())


def test_streaming_generator_is_the_sampling_loop_token_by_token(small_ar):
	"""a6 on the fused path: the generator's tokens are the columns of what `inference_speech` returns for the same call, the
	latents stay valid after it ends, the torch generator is left where the non-streaming loop leaves it; a consumer that walks away after three
	tokens leaves it three draws in and the model usable (ring and noise switched off again)"""
	model, _ = small_ar
	text = torch.randint(1, 255, (1, 9), generator=gen(50)).to(DEV)
	cond = torch.randn(1, 128, generator=gen(51)).to(DEV)
	kw = dict(temperature=0.9, top_k=40, top_p=0.95, repetition_penalty=1.5)
	ids = model.compute_embeddings(cond, text)
	out = list(model.get_generator(inputs=ids, max_length=ids.shape[1] + 30, do_sample=True, num_return_sequences=4, **kw))
	off_stream = torch.cuda.default_generators[0].get_offset()
	toks = torch.stack([t for t, _ in out], 1).clone()
	lats = torch.stack([l for _, l in out], 0).clone()
	want = model.inference_speech(cond, text, do_sample=True, num_return_sequences=4, max_generate_length=30, **kw)
	assert torch.equal(toks, want) and torch.cuda.default_generators[0].get_offset() == off_stream
	assert torch.isfinite(lats).all() and torch.equal(lats, torch.stack([l for _, l in out], 0))      # untouched by the call that followed
	# walk away after three tokens
	g3 = model.get_generator(inputs=ids, max_length=ids.shape[1] + 30, do_sample=True, num_return_sequences=4, **kw)
	first = [next(g3)[0].clone() for _ in range(3)]
	g3.close()
	torch.cuda.synchronize()
	assert torch.equal(torch.stack(first, 1), want[:, :3])
	torch.manual_seed(0); torch.cuda.manual_seed_all(0)
	ref = torch.empty((4, 8194), device=DEV)
	for _ in range(3):
		ref.exponential_(1)
	assert torch.cuda.default_generators[0].get_offset() == off_stream // toks.shape[1] * 3
	again = model.inference_speech(cond, text, do_sample=True, num_return_sequences=4, max_generate_length=30, **kw)
	assert torch.equal(again, want)


def test_a_second_generation_while_a_stream_is_open_is_refused(small_ar):
	"""one generation at a time per handle (its KV cache / noise / latent ring belong to the open stream): refused by name, and fine again after close()"""
	from tortoise_tts_amd import _lib
	model, _ = small_ar
	text = torch.randint(1, 255, (1, 9), generator=gen(70)).to(DEV)
	cond = torch.randn(1, 128, generator=gen(71)).to(DEV)
	kw = dict(temperature=0.9, top_k=0, do_sample=True, num_return_sequences=2)
	ids = model.compute_embeddings(cond, text)
	g1 = model.get_generator(inputs=ids, max_length=ids.shape[1] + 10, **kw)
	g2 = model.get_generator(inputs=ids, max_length=ids.shape[1] + 10, **kw)        # creating a second generator is fine: nothing has run yet
	first = next(g1)[0].clone()
	with pytest.raises(_lib.TTKError, match="streamed generation is still open"):
		model.inference_speech(cond, text, max_generate_length=10, **kw)
	with pytest.raises(_lib.TTKError, match="streamed generation is still open"):
		next(g2)
	with pytest.raises(_lib.TTKError, match="streamed generation is still open"):      # a line batch would overwrite the open stream's cache and latent ring
		model.inference_speech_lines(cond, [text, text[:, :5]], max_generate_length=10, **kw)
	rest = [t.clone() for t, _ in g1]                                                 # the open stream is unharmed by the refused calls
	want = model.inference_speech(cond, text, max_generate_length=10, **kw)
	assert torch.equal(torch.stack([first] + rest, 1), want)


def test_streaming_generator_twice_on_one_model_keeps_both_latent_sets(small_ar):
	"""two streamed generations of the same shape on one model: the second replays the token step captured by the first, and must write ITS latents into
	ITS buffer -- the first call's yielded latents stay what they were (the base of the latent buffer reaches the captured launch through device memory,
	not as an argument frozen at capture)"""
	model, _ = small_ar
	text = torch.randint(1, 255, (1, 9), generator=gen(60)).to(DEV)
	kw = dict(temperature=0.9, top_k=0, do_sample=True, num_return_sequences=4)
	runs = []
	for seed in (61, 62, 62):
		cond = torch.randn(1, 128, generator=gen(seed)).to(DEV)
		ids = model.compute_embeddings(cond, text)
		out = list(model.get_generator(inputs=ids, max_length=ids.shape[1] + 12, **kw))
		torch.cuda.synchronize()
		runs.append((out, torch.stack([l for _, l in out], 0).clone(), torch.stack([t for t, _ in out], 1).clone()))
	(out_a, lat_a, _), (out_b, lat_b, tok_b), (out_c, lat_c, tok_c) = runs
	assert torch.isfinite(lat_b).all() and torch.equal(tok_b, tok_c) and torch.equal(lat_b, lat_c)           # same call twice: same tokens, same latents
	assert not torch.equal(lat_a, lat_b)                                                                   # (a different voice gives different latents)
	assert torch.equal(torch.stack([l for _, l in out_a], 0), lat_a)                                       # the first call's views were not written by the later ones
	assert torch.equal(torch.stack([l for _, l in out_b], 0), lat_b)


def test_streamed_tokens_outlive_later_generations_of_the_same_shape(small_ar):
	"""the yielded tokens are the caller's, as the reference's fresh tensors are (stream_generator.py:1172) -- a consumer that collects the
	yielded tensors and stacks them AFTER another generation of the same shape (streamed or not; both share the cached state's id buffer) must
	still see its own tokens"""
	model, _ = small_ar
	text = torch.randint(1, 255, (1, 9), generator=gen(80)).to(DEV)
	kw = dict(temperature=0.9, top_k=0, do_sample=True, num_return_sequences=4)
	cond_a, cond_b = (torch.randn(1, 128, generator=gen(s)).to(DEV) for s in (81, 82))
	ids = model.compute_embeddings(cond_a, text)
	held = [t for t, _ in model.get_generator(inputs=ids, max_length=ids.shape[1] + 12, **kw)]          # not cloned: kept as yielded
	torch.cuda.synchronize()
	want = torch.stack(held, 1).clone()
	other = model.inference_speech(cond_b, text, max_generate_length=12, **kw)                            # same (B, max_new, warpers) state, non-streamed
	ids_b = model.compute_embeddings(cond_b, text)
	list(model.get_generator(inputs=ids_b, max_length=ids_b.shape[1] + 12, **kw))                         # and streamed
	torch.cuda.synchronize()
	assert not torch.equal(other[:, :want.shape[1]], want[:, :other.shape[1]])                            # the later runs really produced other tokens
	assert torch.equal(torch.stack(held, 1), want)


@pytest.mark.parametrize("b,M,T", [(1, 1, 4), (2, 7, 30), (1, 40, 174), (3, 70, 129)])
def test_diffusion_odd_lengths(small_diff, b, M, T):
	"""Frame counts that straddle the 64-key attention tiles and the 128-row GEMM tiles; nearest-neighbour expansion M -> T."""
	model, oracle = small_diff
	lat = torch.randn(b, M, 128, generator=gen(M))
	cond = torch.randn(b, 256, generator=gen(M + 1))
	x = torch.randn(b, 100, T, generator=gen(M + 2))
	t = torch.randint(0, 4000, (b,), generator=gen(M + 3))
	E = model.timestep_independent(lat.to(DEV), cond.to(DEV), T, False)
	with torch.inference_mode():
		Er = oracle.timestep_independent(lat, cond, T)
		assert maxerr(E, Er) < 2e-4
		assert maxerr(model(x.to(DEV), t.to(DEV), precomputed_aligned_embeddings=Er.to(DEV)), oracle.forward(x, t, Er)) < 5e-4
		assert maxerr(model(x.to(DEV), t.to(DEV), conditioning_free=True), oracle.forward(x, t, None, conditioning_free=True)) < 5e-4


def test_ddim_linearity_of_the_epilogue_and_clamp(small_diff):
	"""Property: with cond_free off the DDIM update is x0*sqrt(ab_prev) + sqrt(1-ab_prev)*eps' with x0 clamped to [-1, 1]; a
	huge start noise must therefore give |mel| <= sqrt(ab_prev) + sqrt(1-ab_prev)*|eps'| and stay finite."""
	from tortoise_tts_amd.diffusion import get_diffuser
	model, _ = small_diff
	T = 50
	E = torch.randn(1, 128, T, generator=gen(3)).to(DEV)
	mel = get_diffuser(steps=5, cond_free=False).sample_loop(model, (1, 100, T), sampler="ddim", noise=1e3 * torch.randn(1, 100, T, generator=gen(4)).to(DEV),
															 model_kwargs={"precomputed_aligned_embeddings": E})
	assert torch.isfinite(mel).all()


def test_whole_loop_ancestral_sampler_equals_the_per_step_calls(small_diff):
	"""ttk_diff_sample_p (two-stream whole loop, noise pre-drawn in loop order) == n x ttk_diff_step, bit for bit; 30 steps as train.py:178 runs."""
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.diffusion import get_diffuser
	model, _ = small_diff
	T, n = 70, 30
	E = torch.randn(1, 128, T, generator=gen(3)).to(DEV)
	x0 = torch.randn(1, 100, T, generator=gen(4)).to(DEV)
	d = get_diffuser(steps=n, cond_free=True)
	torch.manual_seed(5)
	whole = d.sample_loop(model, (1, 100, T), sampler="p", noise=x0, model_kwargs={"precomputed_aligned_embeddings": E})
	after = torch.rand(3, device=DEV)
	torch.manual_seed(5)
	x = x0.clone()
	_lib.check(model.lib.ttk_diff_begin(model._h, E.data_ptr(), 1, T, _lib.stream_ptr()), "ttk_diff_begin")
	for i in reversed(range(n)):
		nz = torch.randn_like(x)
		st = d.step_coefs(i, "p")
		_lib.check(model.lib.ttk_diff_step(model._h, x.data_ptr(), _lib.C.byref(st), nz.data_ptr(), _lib.stream_ptr()), "ttk_diff_step")
	assert torch.equal(whole, x) and torch.equal(after, torch.rand(3, device=DEV)) and torch.isfinite(x).all()
	# a ddim step list handed to the p entry (and the reverse) is refused
	steps = (_lib.StepC * 2)(*[d.step_coefs(i, "ddim") for i in range(2)])
	assert model.lib.ttk_diff_sample_p(model._h, x.data_ptr(), E.data_ptr(), 1, T, steps, 2, x.data_ptr(), None) != 0
	assert b"sampler" in model.lib.ttk_last_error()
	assert model.lib.ttk_diff_sample_p(model._h, x.data_ptr(), E.data_ptr(), 1, T, steps, 2, None, None) != 0


def test_full_size_bf16_properties():
	"""BASELINE-size models (configs[1] shapes, bf16): determinism of the sampled ids across two runs and across graph/eager,
	all ids in range, fixed length with the stop token suppressed; DDIM mel finite and in the clamp-implied range."""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
	ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL, dtype="bf16", device=DEV, max_batch=16, max_ctx=64 + 4 + 40 + 8)
	text = torch.randint(1, 255, (1, 64), generator=gen(1)).to(DEV)
	cond = torch.randn(1, 1024, generator=gen(2)).to(DEV)
	kw = dict(do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16, max_generate_length=40, suppress_tokens=[8193])
	a = ar.inference_speech(cond, text, **kw)
	b = ar.inference_speech(cond, text, **kw)
	ar.use_graph = False
	c = ar.inference_speech(cond, text, **kw)
	assert a.shape == (16, 40) and torch.equal(a, b) and torch.equal(a, c)
	assert int(a.min()) >= 0 and int(a.max()) < 8193
	lat = ar.forward(cond.expand(16, -1), text.expand(16, -1), torch.tensor([64] * 16), a, torch.tensor([40 * 1024] * 16), return_latent=True, clip_inputs=False)
	assert lat.shape == (16, 40, 1024) and torch.isfinite(lat).all()
	del ar
	df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype="bf16", device=DEV)
	T = 40 * 4 * 24000 // 22050
	E = df.timestep_independent(lat[:1], torch.randn(1, 2048, generator=gen(3)).to(DEV), T, False)
	noise = torch.randn(1, 100, T, generator=gen(4)).to(DEV)
	m1 = get_diffuser(6, True).sample_loop(df, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E})
	m2 = get_diffuser(6, True).sample_loop(df, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E})
	assert torch.equal(m1, m2) and torch.isfinite(m1).all() and m1.abs().max() <= 1.0 + 1e-5     # last step: ab_prev = 1 -> x = clamp(x0)


def test_full_size_fused_groupnorm_stats_match_oracle_and_unfused(monkeypatch):
	"""At T % 64 == 0 and C = 1024 the GEMM epilogues emit the GroupNorm statistics (csrc/gemm.hip); check that path against
	the CPU oracle and against the separate-kernel path (TTK_NO_FUSED_GN=1), full-size network, f32."""
	import os
	from tortoise_tts_amd.diffusion import DiffusionTTS
	sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 9)
	fused = DiffusionTTS(sd, W.DIFF_FULL, dtype="f32", device=DEV)
	monkeypatch.setenv("TTK_NO_FUSED_GN", "1")
	plain = DiffusionTTS(sd, W.DIFF_FULL, dtype="f32", device=DEV)
	monkeypatch.delenv("TTK_NO_FUSED_GN")
	T = 128
	x = torch.randn(1, 100, T, generator=gen(1))
	E = torch.randn(1, 1024, T, generator=gen(2))
	t = torch.tensor([1500])
	a = fused(x.to(DEV), t.to(DEV), precomputed_aligned_embeddings=E.to(DEV))
	b = plain(x.to(DEV), t.to(DEV), precomputed_aligned_embeddings=E.to(DEV))
	assert maxerr(a, b) < 1e-4
	with torch.inference_mode():
		ref = O.DiffusionOracle(sd, W.DIFF_FULL).forward(x, t, E)
	assert maxerr(a, ref) < 1e-3


def test_compute_embeddings_fake_id_row_equals_reference(small_ar, golden):
	"""the id row handed to generate(): ones with start_mel last, prefix + 1 long (unified_voice.py:614-630), from the reference itself"""
	model = small_ar[0] if isinstance(small_ar, tuple) else small_ar
	want = torch.from_numpy(golden("wrapper")["fake_ids"])
	text = torch.randint(1, 255, (2, 5), generator=gen(1)).to(DEV)
	got = model.compute_embeddings(torch.zeros(2, W.AR_SMALL.model_dim, device=DEV), text)
	assert got.dtype == torch.long and torch.equal(got.cpu(), want)


def test_new_handles_reject_bad_arguments_with_messages():
	"""vocoder / CLVP / fp8 helper: configuration and argument errors come back as codes + text, never as crashes"""
	import ctypes as C
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.clvp import CLVP
	from tortoise_tts_amd.vocoder import BigVGAN
	lib = _lib.load()
	# vocoder: kernel size not a multiple of the rate; odd channel halving; missing tensor; wrong dtype
	bad = W.VocoderConfig(upsample_rates=(4, 2), upsample_kernel_sizes=(7, 4), upsample_initial_channel=128, resblock_kernel_sizes=(3, 7),
						  resblock_dilation_sizes=((1, 3, 5), (1, 3, 5)))
	with pytest.raises(_lib.TTKError, match="upsampler 0"):
		BigVGAN(W.synth_state_dict(W.vocoder_shapes(bad), 1), bad, dtype="f32", device=DEV)
	sd = W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), 1)
	with pytest.raises(_lib.TTKError, match="lacks"):
		BigVGAN({k: v for k, v in sd.items() if k != "conv_post.bias"}, W.VOC_SMALL, dtype="f32", device=DEV)
	with pytest.raises(_lib.TTKError, match="bf16"):
		BigVGAN(sd, W.VOC_SMALL, dtype="fp8w", device=DEV)
	assert lib.ttk_voc_inference(None, None, 1, 1, None, None) != 0 and b"ttk_voc_inference" in lib.ttk_last_error()
	# CLVP: head width must be 64; shape errors on the text argument
	with pytest.raises(_lib.TTKError, match="head width"):
		c = W.CLVPConfig(dim=96, depth=1, heads=2)
		CLVP(W.synth_state_dict(W.clvp_shapes(c), 1), c, dtype="f32", device=DEV)
	m = CLVP(W.synth_state_dict(W.clvp_shapes(W.CLVP_SMALL), 1), W.CLVP_SMALL, dtype="f32", device=DEV)
	with pytest.raises(_lib.TTKError, match="text must be"):
		m(torch.zeros(2, 4, dtype=torch.long), torch.zeros(3, 5, dtype=torch.long))
	assert lib.ttk_clvp_score(None, None, 1, 1, None, 1, 1, None, None) != 0 and b"ttk_clvp_score" in lib.ttk_last_error()
	# fp8 rounding helper
	assert lib.ttk_fp8_round_weights(None, 0, None, None) != 0 and b"ttk_fp8_round_weights" in lib.ttk_last_error()
	# destroying null handles is a no-op
	assert lib.ttk_voc_destroy(None) == 0 and lib.ttk_clvp_destroy(None) == 0


def test_two_stream_ddim_loop_equals_the_sequential_one(monkeypatch):
	"""`ttk_diff_sample_ddim` runs step j+1's conditioning integrator on a side stream beside step j's body (it depends on the timestep and the
	staged embedding only, diffusion.py:1549-1556); same kernels on the same data, so the mel must equal the one-stream loop bit for bit --
	with and without conditioning-free guidance, odd and even step counts, two utterances back to back on one handle."""
	from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
	cfg = W.DIFF_SMALL
	sd = W.synth_state_dict(W.diffusion_shapes(cfg), 77)
	monkeypatch.setenv("TTK_DIFF_PIPE", "0")
	seq = DiffusionTTS(sd, cfg, dtype="bf16", device=DEV)
	monkeypatch.setenv("TTK_DIFF_PIPE", "1")
	pip = DiffusionTTS(sd, cfg, dtype="bf16", device=DEV)
	g = torch.Generator().manual_seed(5)
	for T, steps, cf in ((70, 5, True), (129, 2, True), (33, 4, False), (70, 3, True)):
		E = torch.randn(1, cfg.model_channels, T, generator=g).to(DEV)
		noise = torch.randn(1, 100, T, generator=g).to(DEV)
		outs = [get_diffuser(steps, cf).sample_loop(m, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E}) for m in (seq, pip, pip)]
		assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]), (T, steps, cf)
