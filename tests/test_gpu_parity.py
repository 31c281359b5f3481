"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors produced by the reference and against
the CPU oracle on the same seeded inputs.  f32 mode is the exact-parity mode; bf16 mode carries stated tolerances."""
import numpy as np
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
	return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
	return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


def relerr(a, b):
	a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
	return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def lib():
	from tortoise_tts_amd import _lib
	return _lib.load()          # fails loudly if libttk.so is missing


def make_ar(cfg, seed, dtype, bf16_exact=False, **kw):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	sd = W.synth_state_dict(W.ar_shapes(cfg), seed, bf16_exact=bf16_exact)
	return UnifiedVoice(sd, cfg, dtype=dtype, device=DEV, **kw), sd


def make_diff(cfg, seed, dtype, bf16_exact=False):
	from tortoise_tts_amd.diffusion import DiffusionTTS
	sd = W.synth_state_dict(W.diffusion_shapes(cfg), seed, bf16_exact=bf16_exact)
	return DiffusionTTS(sd, cfg, dtype=dtype, device=DEV), sd


# ------------------------------------------------------------------------------------------------ AR, exact mode
def _ar_golden_run(model, g, B):
	text, cond = t(g["text"]).to(DEV), t(g["cond"]).to(DEV)
	logits = model._prefill(cond, text, B)
	out = {"prefill": logits.clone()}
	toks = t(g["dec_tokens"]).to(DEV)
	dec = []
	for k in range(toks.shape[1]):
		model._decode(toks[:, k].contiguous(), logits)
		dec.append(logits.clone())
	out["decode"] = torch.stack(dec, 1)
	out["latents"] = model.forward(cond.repeat(B, 1), text.repeat(B, 1), torch.tensor([text.shape[1]] * B, dtype=torch.int32),
								   t(g["codes"]).to(DEV), torch.tensor([g["codes"].shape[1] * 1024] * B), return_latent=True, clip_inputs=False)
	torch.cuda.synchronize()
	return out


def test_ar_small_f32_vs_reference_golden(lib, golden):
	g = golden("ar_small")
	model, _ = make_ar(W.AR_SMALL, int(g["seed"]), "f32", max_batch=4, max_ctx=64)
	out = _ar_golden_run(model, g, int(g["B"]))
	assert maxerr(out["prefill"], g["prefill_logits"]) < 1e-4
	assert maxerr(out["decode"], g["decode_logits"]) < 1e-4
	assert maxerr(out["latents"], g["latents"]) < 1e-4


def test_ar_full_f32_vs_reference_golden(lib, golden):
	g = golden("ar_full")
	model, _ = make_ar(W.AR_FULL, int(g["seed"]), "f32", max_batch=2, max_ctx=64)
	out = _ar_golden_run(model, g, int(g["B"]))
	cols = t(g["logit_cols"]).to(DEV)
	assert maxerr(out["prefill"][:, cols], g["prefill_logits"]) < 5e-4
	assert maxerr(out["decode"][:, :, cols], g["decode_logits"]) < 5e-4
	assert maxerr(out["latents"][:, :, :128], g["latents"]) < 5e-4


def test_ar_small_bf16_tolerance(lib, golden):
	"""bf16 weights/operands, f32 accumulate: logits within 3e-2 relative L2 of the f32 reference vectors."""
	g = golden("ar_small")
	model, _ = make_ar(W.AR_SMALL, int(g["seed"]), "bf16", max_batch=4, max_ctx=64)
	out = _ar_golden_run(model, g, int(g["B"]))
	assert relerr(out["prefill"], g["prefill_logits"]) < 3e-2
	assert relerr(out["decode"], g["decode_logits"]) < 3e-2
	assert relerr(out["latents"], g["latents"]) < 3e-2


@pytest.mark.parametrize("use_graph", [False, True])
def test_inference_speech_ids_bit_exact_f32(lib, use_graph):
	"""Sampled mel-token ids, seed 0, identical to the oracle's (CPU f32 logits, torch.multinomial on the same device type)."""
	cfg = W.AR_SMALL
	model, sd = make_ar(cfg, 11, "f32", max_batch=4, max_ctx=96, use_graph=use_graph)
	text = torch.randint(1, 255, (1, 9), generator=torch.Generator().manual_seed(1))
	cond = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(2))
	kw = dict(num_return_sequences=3, max_generate_length=24, temperature=0.8, top_k=0, top_p=1.0, repetition_penalty=1.0)
	with torch.inference_mode():
		ref = O.inference_speech(O.AROracle(sd, cfg), cond, text, sample_device="cuda", **kw)
	got = model.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, num_beams=1, length_penalty=1.0, **kw)
	assert got.shape == ref.shape and torch.equal(got.cpu(), ref)


def test_inference_speech_warpers_and_repetition_penalty_f32(lib):
	cfg = W.AR_SMALL
	model, sd = make_ar(cfg, 11, "f32", max_batch=4, max_ctx=96)
	text = torch.randint(1, 255, (1, 6), generator=torch.Generator().manual_seed(3))
	cond = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(4))
	kw = dict(num_return_sequences=2, max_generate_length=12, temperature=0.7, top_k=16, top_p=0.9, repetition_penalty=2.0,
			  suppress_tokens=[8193])
	with torch.inference_mode():
		ref = O.inference_speech(O.AROracle(sd, cfg), cond, text, sample_device="cuda", **kw)
	got = model.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, **kw)
	assert torch.equal(got.cpu(), ref)


def test_inference_speech_stops_and_pads(lib):
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 11)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][8193] = 50.0
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	model = UnifiedVoice(sd, cfg, dtype="f32", device=DEV, max_batch=2, max_ctx=64)
	text = torch.randint(1, 255, (1, 5), generator=torch.Generator().manual_seed(3)).to(DEV)
	cond = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(4)).to(DEV)
	out = model.inference_speech(cond, text, num_return_sequences=2, max_generate_length=20, do_sample=True)
	assert out.shape[1] == 1 and (out == 8193).all()


# ------------------------------------------------------------------------------------------------ diffusion, exact mode
def test_diff_small_f32_vs_reference_golden(lib, golden):
	from tortoise_tts_amd.diffusion import get_diffuser
	g = golden("diff_small")
	model, _ = make_diff(W.DIFF_SMALL, int(g["seed"]), "f32")
	T = int(g["T"])
	E = model.timestep_independent(t(g["latents"]).to(DEV), t(g["cond"]).to(DEV), T, False)
	assert maxerr(E, g["E"]) < 1e-4
	x, ts = t(g["x"]).to(DEV), t(g["t"]).to(DEV)
	Eg = t(g["E"]).to(DEV)
	assert maxerr(model(x, ts, precomputed_aligned_embeddings=Eg), g["y_cond"]) < 2e-4
	assert maxerr(model(x, ts, precomputed_aligned_embeddings=Eg, conditioning_free=True), g["y_uncond"]) < 2e-4
	for sampler in ("ddim", "p"):
		for cf in (True, False):
			torch.manual_seed(int(g["sampler_seed"]))
			mel = get_diffuser(steps=4, cond_free=cf).sample_loop(model, (1, 100, T), sampler=sampler, noise=t(g["noise"]).to(DEV),
																  model_kwargs={"precomputed_aligned_embeddings": Eg[:1]}, progress=False)
			if sampler == "ddim":     # deterministic given the start noise
				assert maxerr(mel, g[f"ddim_cf{int(cf)}"]) < 1e-3
			else:                      # ancestral noise comes from the device generator: compare with the oracle instead
				assert mel.shape == (1, 100, T) and torch.isfinite(mel).all()


def test_diff_full_f32_vs_reference_golden(lib, golden):
	g = golden("diff_full")
	model, _ = make_diff(W.DIFF_FULL, int(g["seed"]), "f32")
	T = int(g["T"])
	E = model.timestep_independent(t(g["latents"]).to(DEV), t(g["cond"]).to(DEV), T, False)
	assert maxerr(E, g["E"]) < 5e-4
	x, ts, Eg = t(g["x"]).to(DEV), t(g["t"]).to(DEV), t(g["E"]).to(DEV)
	assert maxerr(model(x, ts, precomputed_aligned_embeddings=Eg), g["y_cond"]) < 1e-3
	assert maxerr(model(x, ts, precomputed_aligned_embeddings=Eg, conditioning_free=True), g["y_uncond"]) < 1e-3


def test_diff_small_bf16_tolerance(lib, golden):
	from tortoise_tts_amd.diffusion import get_diffuser
	g = golden("diff_small")
	model, _ = make_diff(W.DIFF_SMALL, int(g["seed"]), "bf16")
	T = int(g["T"])
	x, ts, Eg = t(g["x"]).to(DEV), t(g["t"]).to(DEV), t(g["E"]).to(DEV)
	assert relerr(model.timestep_independent(t(g["latents"]).to(DEV), t(g["cond"]).to(DEV), T, False), g["E"]) < 3e-2
	assert relerr(model(x, ts, precomputed_aligned_embeddings=Eg), g["y_cond"]) < 5e-2
	mel = get_diffuser(steps=4, cond_free=True).sample_loop(model, (1, 100, T), sampler="ddim", noise=t(g["noise"]).to(DEV),
															model_kwargs={"precomputed_aligned_embeddings": Eg[:1]})
	assert relerr(mel, g["ddim_cf1"]) < 8e-2


def test_p_sampler_matches_oracle_with_same_noise(lib, golden):
	"""Ancestral sampler with the noise drawn on the device and replayed into the oracle."""
	from tortoise_tts_amd.diffusion import get_diffuser
	g = golden("diff_small")
	cfg = W.DIFF_SMALL
	model, sd = make_diff(cfg, int(g["seed"]), "f32")
	T = int(g["T"])
	Eg = t(g["E"])[:1]
	noise0 = t(g["noise"])
	torch.manual_seed(77)
	mel = get_diffuser(steps=4, cond_free=True).sample_loop(model, (1, 100, T), sampler="p", noise=noise0.to(DEV),
															model_kwargs={"precomputed_aligned_embeddings": Eg.to(DEV)})
	torch.manual_seed(77)
	draws = [torch.randn(1, 100, T, device=DEV).cpu() for _ in range(4)]
	sched = O.SpacedSchedule(steps=4, cond_free=True)
	d = O.DiffusionOracle(sd, cfg)
	x = noise0.clone()
	with torch.inference_mode():
		for j, i in enumerate(reversed(range(4))):
			mean, log_var, _ = sched.p_mean_variance(d, x, i, Eg)
			x = mean + (0.0 if i == 0 else 1.0) * torch.exp(0.5 * log_var) * draws[j]
	assert maxerr(mel, x) < 1e-3


def test_multinomial1_is_torch_multinomial_on_device():
	"""sampling.multinomial1 must stay bit-identical (values AND generator consumption) to torch.multinomial(p, 1)."""
	from tortoise_tts_amd.sampling import multinomial1
	for shape, seed in (((16, 8194), 0), ((3, 8194), 5), ((1, 17), 9)):
		p = torch.softmax(torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * 3, -1).to(DEV)
		torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
		a = [torch.multinomial(p, num_samples=1).squeeze(1) for _ in range(5)]
		after_a = torch.rand(4, device=DEV)
		torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
		b = [multinomial1(p) for _ in range(5)]
		after_b = torch.rand(4, device=DEV)
		assert all(torch.equal(x, y) for x, y in zip(a, b)) and torch.equal(after_a, after_b)


def test_fused_sample_step_is_the_reference_sampling_chain():
	"""ttk_sample_step == HF `_sample`'s per-token chain run as torch ops on the device (temperature warper, softmax,
	torch.multinomial, finished-row padding, id append, unfinished update), including the generator stream it consumes."""
	from tortoise_tts_amd import _lib
	lib = _lib.load()
	stop = 8193
	for B, V, temp, seed, sup in ((16, 8194, 0.8, 0, (8193,)), (3, 8194, 1.0, 5, ()), (5, 1000, 0.2, 9, (1, 7, 500)), (64, 8194, 1.3, 2, ())):
		mask = None
		if sup:
			mask = torch.zeros(V, dtype=torch.bool, device=DEV)
			mask[list(sup)] = True
		g = torch.Generator().manual_seed(seed)
		steps = 6
		logits = [(torch.randn((B, V), generator=g) * 4).to(DEV) for _ in range(steps)]
		for lg in logits:                      # make the stop token likely enough that rows finish inside the test
			lg[:, stop if stop < V else V - 1] += 6.0
		stop_id = stop if stop < V else V - 1
		# reference chain (torch ops)
		torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
		unf = torch.ones(B, dtype=torch.long, device=DEV)
		ref_ids = []
		for lg in logits:
			lg = lg if mask is None else torch.where(mask, -float("inf"), lg)
			probs = torch.softmax(lg / temp if temp != 1.0 else lg, dim=-1)
			nxt = torch.multinomial(probs, num_samples=1).squeeze(1)
			nxt = nxt * unf + stop_id * (1 - unf)
			unf = unf.mul((nxt != stop_id).long())
			ref_ids.append(nxt)
		ref_ids = torch.stack(ref_ids, 1)
		after_ref = torch.rand(4, device=DEV)
		# fused kernel
		torch.manual_seed(seed); torch.cuda.manual_seed_all(seed)
		unf2 = torch.ones(B, dtype=torch.long, device=DEV)
		tok = torch.empty(B, dtype=torch.long, device=DEV)
		ids = torch.full((B, steps - 1), -1, dtype=torch.long, device=DEV)       # one column short: the last write must be skipped
		col = torch.zeros(B, dtype=torch.long, device=DEV)
		hist = torch.full((B, 3 + steps), -7, dtype=torch.long, device=DEV)
		q = torch.empty((B, V), device=DEV)
		live = torch.full((1,), B, dtype=torch.int32, device=DEV)
		done = torch.zeros(1, dtype=torch.int32).pin_memory()          # the flag may live in pinned host memory
		toks, done_at = [], []
		for lg in logits:
			q.exponential_(1)
			_lib.check(lib.ttk_sample_step(lg.data_ptr(), lg.stride(0), B, V, q.data_ptr(), q.stride(0), _lib.ptr(mask), temp, stop_id, unf2.data_ptr(),
										   tok.data_ptr(), ids.data_ptr(), ids.stride(0), ids.shape[1], col.data_ptr(), hist.data_ptr(),
										   hist.stride(0), 3, live.data_ptr(), done.data_ptr(), _lib.stream_ptr()), "ttk_sample_step")
			torch.cuda.synchronize()
			done_at.append(int(done.item()))
			toks.append(tok.clone())
		after = torch.rand(4, device=DEV)
		toks = torch.stack(toks, 1)
		assert torch.equal(toks, ref_ids), (B, V, temp)
		assert torch.equal(ids, ref_ids[:, :steps - 1]) and torch.equal(hist[:, 3:], ref_ids) and bool((hist[:, :3] == -7).all())
		assert torch.equal(unf2, unf) and bool((col == steps).all()) and torch.equal(after, after_ref)
		# the all-done flag rises exactly at the step after which HF's `unfinished_sequences.max() == 0` holds
		# ... and holds the number of tokens sampled at that step from then on (0 before): a host that runs ahead can tell where the end was
		fin = [bool(((ref_ids[:, :k + 1] == stop_id).any(dim=1)).all()) for k in range(steps)]
		ref_done = [(fin.index(True) + 1) if f else 0 for f in fin]
		assert done_at == ref_done and int(live.item()) == int(unf.sum())
	# argument checking: a null pointer / bad temperature is an error code with a message, not a crash
	assert lib.ttk_sample_step(None, 0, 1, 1, None, 0, None, 1.0, 0, None, None, None, 0, 0, None, None, 0, 0, None, None, None) != 0
	assert b"ttk_sample_step" in lib.ttk_last_error()


@pytest.mark.parametrize("temp,top_k,top_p,pen", [(0.8, 16, 1.0, 1.0), (0.8, 0, 0.9, 1.0), (1.0, 0, 1.0, 2.0), (0.7, 16, 0.9, 2.0),
												   (1.3, 50, 0.5, 1.2), (0.2, 1, 1.0, 1.0), (0.8, 8000, 0.999, 1.0), (1.0, 8194, 0.05, 5.0)])
def test_warpers_inside_the_sampling_kernel_equal_the_torch_op_chain(temp, top_k, top_p, pen):
	"""ttk_sample_step_warped == the reference's processor / warper chain run as torch ops on the device (repetition penalty over
	input_ids, suppress_tokens, temperature, top-k, top-p -- the oracle's `process_logits`, HF's classes restated), then softmax +
	torch.multinomial, token by token with a growing history that contains repeats, including the generator stream it consumes."""
	from tortoise_tts_amd import _lib
	lib = _lib.load()
	B, V, stop, steps, off = 16, 8194, 8193, 8, 2
	g = torch.Generator().manual_seed(int(temp * 100) + top_k)
	logits = [(torch.randn((B, V), generator=g) * 4).to(DEV) for _ in range(steps)]
	for lg in logits:
		lg[:, :40] += 5.0                    # a few dominant ids: sampled tokens repeat, so the penalty meets duplicates in input_ids
	sup = [7, 8193]
	mask = torch.zeros(V, dtype=torch.bool, device=DEV)
	mask[sup] = True
	# reference chain (torch ops on the device)
	torch.manual_seed(3); torch.cuda.manual_seed_all(3)
	input_ids = torch.ones((B, off), dtype=torch.long, device=DEV)
	input_ids[:, -1] = 8192
	ref = []
	for lg in logits:
		sc = O.process_logits(input_ids, lg, temperature=temp, top_k=top_k, top_p=top_p, repetition_penalty=pen, suppress_tokens=sup)
		nxt = torch.multinomial(torch.softmax(sc, dim=-1), num_samples=1).squeeze(1)
		input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
		ref.append(nxt)
	ref = torch.stack(ref, 1)
	after_ref = torch.rand(4, device=DEV)
	# fused kernel
	torch.manual_seed(3); torch.cuda.manual_seed_all(3)
	unf = torch.ones(B, dtype=torch.long, device=DEV)
	tok = torch.empty(B, dtype=torch.long, device=DEV)
	ids = torch.full((B, steps), -1, dtype=torch.long, device=DEV)
	col = torch.zeros(B, dtype=torch.long, device=DEV)
	hist = torch.ones((B, off + steps), dtype=torch.long, device=DEV)
	hist[:, off - 1] = 8192
	q = torch.empty((B, V), device=DEV)
	a = _lib.SampleArgs()
	a.ld, a.B, a.V, a.q, a.ldq = V, B, V, q.data_ptr(), V
	a.suppress, a.temperature, a.top_k, a.top_p, a.repetition_penalty = mask.data_ptr(), temp, top_k, top_p, pen
	a.stop_token, a.unfinished, a.tok, a.ids, a.ids_ld, a.ids_cols, a.col = stop, unf.data_ptr(), tok.data_ptr(), ids.data_ptr(), steps, steps, col.data_ptr()
	a.history, a.hist_ld, a.hist_off = hist.data_ptr(), hist.stride(0), off
	for lg in logits:
		q.exponential_(1)
		a.scores = lg.data_ptr()
		_lib.check(lib.ttk_sample_step_warped(_lib.C.byref(a), _lib.stream_ptr()), "ttk_sample_step_warped")
	torch.cuda.synchronize()
	after = torch.rand(4, device=DEV)
	assert torch.equal(ids, ref), (ids != ref).sum().item()
	assert torch.equal(hist[:, off:], ref) and torch.equal(after, after_ref)
	if top_k == 1:
		assert torch.equal(ids, torch.stack([torch.where(mask, -float("inf"), lg).argmax(-1) for lg in logits], 1))   # greedy = argmax of what is left


def _probe_kept(lib, lg, probe, temp, top_k, top_p):
	"""is token probe[b] of row b still there after the kernel's warpers?  The noise is 1 everywhere except a vanishing value at the probe: a kept
	probe wins argmax(p / q) whatever its probability, a removed one (p = 0) cannot."""
	from tortoise_tts_amd import _lib
	B, V = lg.shape
	q = torch.ones((B, V), device=DEV)
	q[torch.arange(B, device=DEV), probe] = 1e-30
	unf = torch.ones(B, dtype=torch.long, device=DEV)
	tok = torch.empty(B, dtype=torch.long, device=DEV)
	ids = torch.full((B, 1), -1, dtype=torch.long, device=DEV)
	col = torch.zeros(B, dtype=torch.long, device=DEV)
	a = _lib.SampleArgs()
	a.scores, a.ld, a.B, a.V, a.q, a.ldq = lg.data_ptr(), V, B, V, q.data_ptr(), V
	a.temperature, a.top_k, a.top_p, a.repetition_penalty = temp, top_k, top_p, 1.0
	a.stop_token, a.unfinished, a.tok, a.ids, a.ids_ld, a.ids_cols, a.col = V + 5, unf.data_ptr(), tok.data_ptr(), ids.data_ptr(), 1, 1, col.data_ptr()
	_lib.check(lib.ttk_sample_step_warped(_lib.C.byref(a), _lib.stream_ptr()), "ttk_sample_step_warped")
	torch.cuda.synchronize()
	return tok == probe


def test_top_p_boundary_stress():
	"""the in-kernel top-p takes its cut from exact 2^-40 fixed-point masses, HF's TopPLogitsWarper from an f32 `cumsum` (a parallel
	scan: its rounding pattern is the library's) of the sorted f32 probabilities.  Probing the kernel's kept set at the boundary -- the least
	likely token the torch chain keeps, the most likely one it removes -- over 24 batches x 16 rows x 8 values of top_p (peaked and flat rows):
	the two disagree only where the ascending cumulative mass at that token, summed in f64, lies within f32 rounding of 1 - top_p (a tie the
	two roundings break differently), and in well under 1 % of the rows.  `UnifiedVoice(..., hf_exact_top_p=True)` runs HF's warper as torch
	ops in front of the kernel for callers that need its rounding bit for bit."""
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.sampling import LogitsPipeline
	lib = _lib.load()
	B, V = 16, 8194
	g = torch.Generator().manual_seed(77)
	checked, ties = 0, []

	def cum_mass_at(scaled_row, idx):
		"""ascending cumulative softmax mass up to and including token idx, in f64"""
		p = torch.softmax(scaled_row.double(), dim=-1)
		return float(p[scaled_row <= scaled_row[idx]].sum())

	for trial in range(24):
		scale = [0.5, 1.0, 2.0, 4.0, 8.0, 16.0][trial % 6]
		lg = (torch.randn((B, V), generator=g) * scale).to(DEV)
		temp = [1.0, 0.8, 1.3][trial % 3]
		for top_p in (0.05, 0.3, 0.5, 0.7, 0.9, 0.95, 0.99, 0.999):
			sc = LogitsPipeline(temperature=temp, top_p=top_p, vocab=V, device=DEV)(None, lg)
			kept = torch.isfinite(sc)
			scaled = lg / temp
			least_kept = torch.where(kept, scaled, torch.full_like(scaled, float("inf"))).argmin(dim=-1)
			ok_kept = _probe_kept(lib, lg, least_kept, temp, 0, top_p)
			has_removed = (~kept).any(dim=-1)
			top_removed = torch.where(~kept, scaled, torch.full_like(scaled, float("-inf"))).argmax(dim=-1)
			wrongly_kept = _probe_kept(lib, lg, top_removed, temp, 0, top_p) & has_removed
			for b in torch.nonzero(~ok_kept | wrongly_kept).flatten().tolist():
				idx = int(top_removed[b]) if bool(wrongly_kept[b]) else int(least_kept[b])
				cm = cum_mass_at(scaled[b].cpu(), idx)
				# the torch chain removes a token iff its cumulative mass <= 1 - top_p: a disagreement about it must sit ON that threshold
				ties.append((trial, top_p, b, abs(cm - (1 - top_p))))
			checked += B
	assert checked == 24 * 8 * 16
	assert all(t[3] < 2e-6 for t in ties), ties                      # only ties within f32 rounding of the threshold
	assert len(ties) <= checked // 100, (len(ties), checked)
	print(f"\n[top-p] {len(ties)} boundary ties in {checked} rows: {ties}")


def test_top_k_treats_the_two_zeros_as_equal():
	"""`scores < kth` compares values: with a k-th largest score of +0.0 the -0.0 entries stay, as in torch (the radix key of -0.0 sorts below
	+0.0 as a bit pattern; the kernel canonicalises it)"""
	from tortoise_tts_amd import _lib
	lib = _lib.load()
	B, V = 4, 8194
	lg = torch.full((B, V), -5.0, device=DEV)
	lg[:, :10] = 3.0
	lg[:, 100:120] = 0.0
	lg[:, 200:220] = -0.0
	assert bool(torch.signbit(lg[:, 200]).all())
	kth = torch.topk(lg, 15)[0][..., -1, None]
	assert bool((kth == 0).all()) and not bool((lg[:, 200:220] < kth).any())               # torch keeps them
	for probe in (200, 219, 100, 5):
		assert bool(_probe_kept(lib, lg, torch.full((B,), probe, device=DEV), 1.0, 15, 1.0).all()), probe
	assert not bool(_probe_kept(lib, lg, torch.full((B,), 300, device=DEV), 1.0, 15, 1.0).any())      # -5: below the k-th largest


def test_sampling_kernel_rejects_what_it_cannot_do():
	from tortoise_tts_amd import _lib
	lib = _lib.load()
	a = _lib.SampleArgs()
	assert lib.ttk_sample_step_warped(_lib.C.byref(a), None) != 0 and b"ttk_sample_step_warped" in lib.ttk_last_error()
	x = torch.zeros(1, 20000, device=DEV)
	i64 = torch.zeros(4, dtype=torch.long, device=DEV)
	a.scores, a.ld, a.B, a.V, a.q, a.ldq, a.temperature, a.top_k = x.data_ptr(), 20000, 1, 20000, x.data_ptr(), 20000, 1.0, 5
	a.unfinished = a.tok = a.ids = a.col = i64.data_ptr()
	assert lib.ttk_sample_step_warped(_lib.C.byref(a), None) != 0 and b"9216" in lib.ttk_last_error()   # wide rows: no in-kernel top-k
	a.top_k, a.V, a.ld, a.ldq, a.repetition_penalty = 0, 100, 100, 100, 1.5
	assert lib.ttk_sample_step_warped(_lib.C.byref(a), None) != 0 and b"history" in lib.ttk_last_error()


@pytest.mark.parametrize("bias", [2.0, 4.5, 9.0])
def test_early_stop_without_per_token_sync_keeps_ids_and_rng_stream(lib, bias):
	"""Graph mode looks at the all-finished flag a few tokens late; the tokens generated past the true end must be cut off and the
	generator offset they consumed handed back, so that ids AND the next draw (the diffusion noise, inference.py:404) equal the
	reference loop's, which tests `unfinished_sequences.max() == 0` after every token."""
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 11)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][cfg.stop_mel_token] = bias          # rows stop after a few to a few dozen tokens, at different steps
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	text = torch.randint(1, 255, (1, 7), generator=torch.Generator().manual_seed(5))
	cond = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(6))
	kw = dict(num_return_sequences=4, max_generate_length=60, temperature=0.8, top_k=0)
	with torch.inference_mode():
		ref = O.inference_speech(O.AROracle(sd, cfg), cond, text, sample_device="cuda", **kw)
		after_ref = torch.rand(8, device=DEV)
		outs = []
		for use_graph in (False, True):
			model = UnifiedVoice(sd, cfg, dtype="f32", device=DEV, max_batch=4, max_ctx=96, use_graph=use_graph)
			for _ in range(2):                       # second call replays the cached graph from the start
				got = model.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, **kw)
				after = torch.rand(8, device=DEV)
				assert got.shape == ref.shape and torch.equal(got.cpu(), ref), (use_graph, got.shape, ref.shape)
				assert torch.equal(after, after_ref), use_graph


@pytest.mark.parametrize("mass,temp,top_k,pen", [(0.9, 0.8, 0, 1.0), (0.5, 1.0, 0, 1.0), (0.2, 1.2, 16, 2.0), (0.99, 0.7, 0, 1.0)])
def test_typical_sampling_inside_the_kernel_equals_the_torch_op_chain(mass, temp, top_k, pen):
	"""TypicalLogitsWarper (unified_voice.py:47-75) inside ttk_sample_step_warped (round 3), where the reference's custom logits_processor runs -- after
	the repetition penalty and suppress_tokens, before temperature / top-k: sampled ids, history and the generator stream equal the torch-op chain
	(tortoise_tts_amd.sampling.LogitsPipeline + torch.multinomial), token by token with a growing history; the kept set is probed at its boundary."""
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.sampling import LogitsPipeline
	lib = _lib.load()
	B, V, stop, steps, off = 16, 8194, 8193, 8, 2
	g = torch.Generator().manual_seed(int(mass * 100) + top_k)
	logits = [(torch.randn((B, V), generator=g) * 3).to(DEV) for _ in range(steps)]
	for lg in logits:
		lg[:, :40] += 4.0
	sup = [7, 8193]
	pipe = LogitsPipeline(temperature=temp, top_k=top_k, repetition_penalty=pen, suppress_tokens=sup, typical_mass=mass, vocab=V, device=DEV)
	torch.manual_seed(5); torch.cuda.manual_seed_all(5)
	input_ids = torch.ones((B, off), dtype=torch.long, device=DEV)
	input_ids[:, -1] = 8192
	ref = []
	for lg in logits:
		nxt = torch.multinomial(torch.softmax(pipe(input_ids, lg), dim=-1), num_samples=1).squeeze(1)
		input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
		ref.append(nxt)
	ref = torch.stack(ref, 1)
	after_ref = torch.rand(4, device=DEV)
	torch.manual_seed(5); torch.cuda.manual_seed_all(5)
	unf = torch.ones(B, dtype=torch.long, device=DEV)
	tok = torch.empty(B, dtype=torch.long, device=DEV)
	ids = torch.full((B, steps), -1, dtype=torch.long, device=DEV)
	col = torch.zeros(B, dtype=torch.long, device=DEV)
	hist = torch.ones((B, off + steps), dtype=torch.long, device=DEV)
	hist[:, off - 1] = 8192
	mask = torch.zeros(V, dtype=torch.bool, device=DEV)
	mask[sup] = True
	q = torch.empty((B, V), device=DEV)
	a = _lib.SampleArgs()
	a.ld, a.B, a.V, a.q, a.ldq = V, B, V, q.data_ptr(), V
	a.suppress, a.temperature, a.top_k, a.top_p, a.repetition_penalty, a.typical_mass = mask.data_ptr(), temp, top_k, 1.0, pen, mass
	a.stop_token, a.unfinished, a.tok, a.ids, a.ids_ld, a.ids_cols, a.col = stop, unf.data_ptr(), tok.data_ptr(), ids.data_ptr(), steps, steps, col.data_ptr()
	a.history, a.hist_ld, a.hist_off = hist.data_ptr(), hist.stride(0), off
	for lg in logits:
		q.exponential_(1)
		a.scores = lg.data_ptr()
		_lib.check(lib.ttk_sample_step_warped(_lib.C.byref(a), _lib.stream_ptr()), "ttk_sample_step_warped")
	torch.cuda.synchronize()
	after = torch.rand(4, device=DEV)
	assert torch.equal(ids, ref), (ids != ref).sum().item()
	assert torch.equal(hist[:, off:], ref) and torch.equal(after, after_ref)
	# the kept set at its boundary, without history (penalty off): the least typical token torch keeps is kept, the most typical one it drops is dropped
	if pen == 1.0 and top_k == 0:
		lg = logits[0]
		sc = LogitsPipeline(temperature=temp, suppress_tokens=sup, typical_mass=mass, vocab=V, device=DEV)(None, lg)
		kept = torch.isfinite(sc)
		lp = torch.log_softmax(torch.where(mask, -float("inf"), lg), dim=-1)
		H = -(lp * lp.exp()).nansum(-1, keepdim=True)
		dist = (-lp - H).abs()
		far_kept = torch.where(kept, dist, torch.full_like(dist, -1.0)).argmax(dim=-1)
		near_drop = torch.where(~kept & ~mask, dist, torch.full_like(dist, float("inf"))).argmin(dim=-1)

		def probe(idx):
			qq = torch.ones((B, V), device=DEV)
			qq[torch.arange(B, device=DEV), idx] = 1e-30
			a2 = _lib.SampleArgs()
			u2, t2, i2, c2 = torch.ones(B, dtype=torch.long, device=DEV), torch.empty(B, dtype=torch.long, device=DEV), torch.full((B, 1), -1, dtype=torch.long, device=DEV), torch.zeros(B, dtype=torch.long, device=DEV)
			a2.scores, a2.ld, a2.B, a2.V, a2.q, a2.ldq = lg.data_ptr(), V, B, V, qq.data_ptr(), V
			a2.suppress, a2.temperature, a2.top_k, a2.top_p, a2.repetition_penalty, a2.typical_mass = mask.data_ptr(), temp, 0, 1.0, 1.0, mass
			a2.stop_token, a2.unfinished, a2.tok, a2.ids, a2.ids_ld, a2.ids_cols, a2.col = V + 5, u2.data_ptr(), t2.data_ptr(), i2.data_ptr(), 1, 1, c2.data_ptr()
			_lib.check(lib.ttk_sample_step_warped(_lib.C.byref(a2), _lib.stream_ptr()), "ttk_sample_step_warped")
			torch.cuda.synchronize()
			return t2 == idx
		assert bool(probe(far_kept).all()) and not bool(probe(near_drop).any())
