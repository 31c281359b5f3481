"""Handles reused across calls of different shape give what a fresh handle gives (state that outlives a call -- captured token steps, generation
states, grow-only work buffers, per-utterance precomputations, the hidden ring -- must never leak from one call into the next).

The reference builds nothing per call that survives it (HF `generate` and the diffusion loops are stateless apart from the module weights), so the
product's caches have no counterpart there; every case compares a long-lived handle with freshly built ones BIT FOR BIT.  GPU only; calls go
through the C ABI."""
import pytest
import torch

from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
	return torch.Generator().manual_seed(seed)


def new_ar(dtype="f32", max_batch=8):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	return UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 31), W.AR_SMALL, dtype=dtype, device=DEV, max_batch=max_batch, max_ctx=160)


def new_diff(dtype="f32"):
	from tortoise_tts_amd.diffusion import DiffusionTTS
	return DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32), W.DIFF_SMALL, dtype=dtype, device=DEV)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_sampling_calls_of_changing_shape_on_one_model(dtype):
	"""text lengths, candidate counts, lengths and warpers change from call to call (several generation states and captured steps alive at once,
	the least recently used one evicted and rebuilt); a streamed generation in between; every result equals a fresh model's"""
	calls = [
		dict(Tt=9, B=4, M=20, kw=dict(temperature=0.8, top_k=0)),
		dict(Tt=23, B=4, M=20, kw=dict(temperature=0.8, top_k=0)),                       # same state, longer prefix
		dict(Tt=5, B=8, M=12, kw=dict(temperature=1.0, top_k=30, top_p=0.9)),
		dict(Tt=9, B=2, M=33, kw=dict(temperature=0.7, top_k=0, repetition_penalty=2.0)),
		dict(Tt=14, B=1, M=20, kw=dict(temperature=0.8, top_k=0)),
		dict(Tt=9, B=3, M=16, kw=dict(temperature=0.9, top_k=0, suppress_tokens=[W.AR_SMALL.stop_mel_token])),
		dict(Tt=9, B=4, M=20, kw=dict(temperature=0.8, top_k=0)),                       # the first state again, after four others
	]
	keep = new_ar(dtype)
	for i, c in enumerate(calls):
		text = torch.randint(1, 255, (1, c["Tt"]), generator=gen(100 + i)).to(DEV)
		cond = torch.randn(1, 128, generator=gen(200 + i)).to(DEV)
		args = dict(do_sample=True, num_return_sequences=c["B"], max_generate_length=c["M"], **c["kw"])
		with torch.inference_mode():
			got = keep.inference_speech(cond, text, **args)
			off_keep = torch.cuda.default_generators[0].get_offset()
			want = new_ar(dtype).inference_speech(cond, text, **args)
			off_new = torch.cuda.default_generators[0].get_offset()
		assert torch.equal(got, want) and off_keep == off_new, (i, c)
		if i == 2:      # a streamed generation on the long-lived model between two ordinary ones
			ids = keep.compute_embeddings(cond, text)
			out = list(keep.get_generator(inputs=ids, max_length=ids.shape[1] + 10, do_sample=True, num_return_sequences=4, temperature=0.8, top_k=0))
			fresh = new_ar(dtype)
			ids2 = fresh.compute_embeddings(cond, text)
			ref = list(fresh.get_generator(inputs=ids2, max_length=ids2.shape[1] + 10, do_sample=True, num_return_sequences=4, temperature=0.8, top_k=0))
			assert len(out) == len(ref)
			for (t, l), (rt, rl) in zip(out, ref):
				assert torch.equal(t, rt) and torch.equal(l, rl)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_diffusion_loops_of_changing_length_on_one_handle(dtype):
	"""frame counts go up and down (grow-only buffers keep the larger capacity, the per-utterance precomputation of ttk_diff_begin is redone), a ragged
	batch of two lines in between, both samplers: every result equals a fresh handle's"""
	from tortoise_tts_amd.diffusion import get_diffuser
	keep = new_diff(dtype)
	C = W.DIFF_SMALL.model_channels
	plan = [("ddim", 64), ("ddim", 200), ("p", 40), ("ddim", 64), ("lines", (128, 70)), ("ddim", 129), ("ddim", 200)]
	for i, (kind, T) in enumerate(plan):
		with torch.inference_mode():
			if kind == "lines":
				noises = [torch.randn((1, 100, t), generator=gen(300 + i + j)).to(DEV) for j, t in enumerate(T)]
				Es = [torch.randn((1, C, t), generator=gen(400 + i + j)).to(DEV) for j, t in enumerate(T)]
				got = get_diffuser(steps=5, cond_free=True).sample_loop_lines(keep, noises, Es)
				want = get_diffuser(steps=5, cond_free=True).sample_loop_lines(new_diff(dtype), noises, Es)
				for a, b in zip(got, want):
					assert torch.equal(a, b), (i, kind, T)
				continue
			noise = torch.randn((1, 100, T), generator=gen(300 + i)).to(DEV)
			E = torch.randn((1, C, T), generator=gen(400 + i)).to(DEV)
			run = lambda m: (torch.manual_seed(7), torch.cuda.manual_seed_all(7),
							 get_diffuser(steps=5, cond_free=True).sample_loop(m, (1, 100, T), sampler=kind, noise=noise, model_kwargs={"precomputed_aligned_embeddings": E}))[2]
			got, want = run(keep), run(new_diff(dtype))
		torch.cuda.synchronize()
		assert torch.isfinite(got).all() and torch.equal(got, want), (i, kind, T, (got - want).abs().max().item())


def _poison_freed_device_memory():
	"""every block the caching allocator holds but has handed to nobody gets 0xFF bytes (NaN as floats, -1 as ids): a captured launch that still reads a
	freed per-call tensor then computes garbage instead of silently finding the old values, and one that WRITES there lands in nobody's tensor.  The blocks
	stay with the allocator (no empty_cache): a stale access must show up as a wrong result, never as a GPU fault."""
	torch.cuda.synchronize()
	sizes = [b["size"] for seg in torch.cuda.memory_snapshot() for b in seg["blocks"] if b["state"] == "inactive"]
	hold = []
	for n in sorted(sizes, reverse=True):      # best fit hands each block back; all of them are held until the end so none is returned twice
		t = torch.empty(n, dtype=torch.uint8, device=DEV)
		t.fill_(0xFF)
		hold.append(t)
	torch.cuda.synchronize()
	del hold


def test_captured_token_step_replayed_after_its_calls_temporaries_are_gone(monkeypatch):
	"""VERDICT r03 next #6b: the class of round 3's streaming leak (a capture freezing a per-call pointer).  TTK_DEBUG_POISON=1 makes the library fill the
	dense passes' scratch with 0xFF when a prefill / latent pass ends; between the calls every freed torch block is filled the same way and the call's
	own tensors are dropped.  The second and third generation of each shape REPLAY the step the first one captured -- streamed and not, line batches,
	with a latent pass of a different size in between (grow-only scratch reallocated) -- and must equal a fresh model bit for bit."""
	monkeypatch.setenv("TTK_DEBUG_POISON", "1")
	keep = new_ar("bf16", max_batch=8)
	assert keep.use_graph
	kw = dict(do_sample=True, temperature=0.8, top_k=0)
	for rnd in range(3):
		text = torch.randint(1, 255, (1, 9 + 7 * rnd), generator=gen(500 + rnd)).to(DEV)
		cond = torch.randn(1, 128, generator=gen(510 + rnd)).to(DEV)
		with torch.inference_mode():
			got = keep.inference_speech(cond, text, num_return_sequences=4, max_generate_length=24, **kw)
			ids = keep.compute_embeddings(cond, text)
			streamed = list(keep.get_generator(inputs=ids, max_length=ids.shape[1] + 14, num_return_sequences=4, **kw))
			lines = keep.inference_speech_lines(cond, [text, text[:, :5]], num_return_sequences=4, max_generate_length=20, **kw)
			lat = keep.forward(cond.expand(4, -1), text.expand(4, -1), torch.tensor([text.shape[1]] * 4), got, torch.tensor([got.shape[1] * 1024] * 4),
							   return_latent=True, clip_inputs=False)
			torch.cuda.synchronize()
			got, lat = got.cpu(), lat.cpu()
			streamed = [(t.cpu(), l.cpu()) for t, l in streamed]
			lines = [l.cpu() for l in lines]
			del ids
			_poison_freed_device_memory()
			fresh = new_ar("bf16", max_batch=8)
			want = fresh.inference_speech(cond, text, num_return_sequences=4, max_generate_length=24, **kw)
			ids2 = fresh.compute_embeddings(cond, text)
			want_streamed = list(fresh.get_generator(inputs=ids2, max_length=ids2.shape[1] + 14, num_return_sequences=4, **kw))
			want_lines = fresh.inference_speech_lines(cond, [text, text[:, :5]], num_return_sequences=4, max_generate_length=20, **kw)
			want_lat = fresh.forward(cond.expand(4, -1), text.expand(4, -1), torch.tensor([text.shape[1]] * 4), want, torch.tensor([want.shape[1] * 1024] * 4),
									 return_latent=True, clip_inputs=False)
		assert torch.equal(got, want.cpu()) and torch.isfinite(lat).all() and torch.equal(lat, want_lat.cpu()), rnd
		assert len(streamed) == len(want_streamed) and all(torch.equal(a, c.cpu()) and torch.equal(b, d.cpu()) for (a, b), (c, d) in zip(streamed, want_streamed)), rnd
		assert all(torch.equal(a, b.cpu()) for a, b in zip(lines, want_lines)), rnd
		del fresh, want, want_streamed, want_lines, want_lat, ids2
		_poison_freed_device_memory()
	st = [s for s in keep._states.values()]
	assert any(s.graph is not None for s in st) and any(s.stream_graph is not None for s in st)      # the replayed steps really were captured ones


def test_more_decode_handles_than_position_line_slots():
	"""round 5: the decode attention reads {cache length, shared-prefix length} of up to 8 handles from ONE 64-byte line at a link-time address (csrc/attn.hip: the request
	leaves beside the kernel-argument loads instead of behind them); a ninth handle keeps the words in its own allocation and its launches take the pointer form.  Ten live
	handles on the same weights: every one gives the same ids (free-running, graph replay included), a slot is handed back when a handle goes and reused by the next one."""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 11)
	text = torch.randint(1, 255, (1, 9), generator=torch.Generator().manual_seed(1)).to(DEV)
	cond = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(2)).to(DEV)
	kw = dict(do_sample=True, num_return_sequences=3, max_generate_length=20, temperature=0.8, top_k=0)
	models = [UnifiedVoice(sd, cfg, dtype="f32", device=DEV, max_batch=4, max_ctx=96) for _ in range(10)]
	want = models[0].inference_speech(cond, text, **kw)
	for i in (3, 7, 8, 9, 0):                       # slots 3, 7; the two handles beyond the line; the first again (its captured step replayed)
		got = models[i].inference_speech(cond, text, **kw)
		assert torch.equal(got, want), i
	del models[2], models[5]
	torch.cuda.synchronize()
	fresh = [UnifiedVoice(sd, cfg, dtype="f32", device=DEV, max_batch=4, max_ctx=96) for _ in range(3)]      # two take the freed slots, the third has none
	for m in fresh + [models[-1]]:
		assert torch.equal(m.inference_speech(cond, text, **kw), want)
