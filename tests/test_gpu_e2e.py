"""The whole hot-path section of `TTS.inference` (inference.py:331-413) through `TTSHotPath`, stage by stage against the oracle, and
the pipelined multi-line variant against the sequential one.  GPU only; calls go through the C ABI."""
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def small():
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.diffusion import DiffusionTTS
	from tortoise_tts_amd.inference import TTSHotPath
	asd = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 31)
	dsd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32)
	ar = UnifiedVoice(asd, W.AR_SMALL, dtype="f32", device=DEV, max_batch=8, max_ctx=128)
	df = DiffusionTTS(dsd, W.DIFF_SMALL, dtype="f32", device=DEV)
	return TTSHotPath(ar, df), O.AROracle(asd, W.AR_SMALL), O.DiffusionOracle(dsd, W.DIFF_SMALL)


def _inputs(seed, Tt):
	g = torch.Generator().manual_seed(seed)
	return (torch.randint(1, 255, (1, Tt), generator=g), torch.randn(1, W.AR_SMALL.model_dim, generator=g),
			torch.randn(1, 2 * W.DIFF_SMALL.model_channels, generator=g))


@pytest.mark.parametrize("candidates,steps,Tt", [(4, 5, 9), (1, 3, 4)])
def test_inference_matches_oracle_stage_by_stage(small, candidates, steps, Tt):
	tts, aro, dor = small
	text, al, dl = _inputs(100 + candidates, Tt)
	max_ar = 24
	with torch.inference_mode():
		mels, seconds, aux = tts.inference(text, al.to(DEV), dl.to(DEV), max_ar_steps=max_ar, max_diffusion_steps=steps, candidates=candidates,
										   suppress_tokens=[W.AR_SMALL.stop_mel_token], return_all=True)
		# oracle, same stages; sampling on the device so both consume the same Philox stream (generate reseeds to 0)
		ref_ids = O.inference_speech(aro, al, text, num_return_sequences=candidates, max_generate_length=max_ar, temperature=0.8, top_k=0,
									 sample_device="cuda", suppress_tokens=[W.AR_SMALL.stop_mel_token])
		noise_ref = torch.randn((1, 100, aux["noise"].shape[-1]), device=DEV)       # the draw TTS.inference makes next (inference.py:404)
		ref_codes = O.fix_stop_tokens(ref_ids, W.AR_SMALL.stop_mel_token)
		assert torch.equal(aux["codes"].cpu(), ref_codes)                           # integer ids: bit-exact
		ref_lat = aro.forward_latents(al.repeat(candidates, 1), text.repeat(candidates, 1), ref_codes)
		ref_lat = O.trim_calm_tokens(ref_codes, ref_lat)[:1]
		assert aux["latents"].shape == ref_lat.shape and (aux["latents"].cpu() - ref_lat).abs().max() < 1e-3
		T = O.mel_frames_for(ref_lat.shape[1])
		assert mels.shape == (1, 100, T) and seconds == T * 256 / 24000
		assert torch.equal(aux["noise"], noise_ref)
		ref_E = dor.timestep_independent(ref_lat, dl, T)
		assert (aux["E"].cpu() - ref_E).abs().max() < 1e-3
		ref_mel = O.SpacedSchedule(steps=steps, cond_free=True).sample_loop(dor, noise_ref.cpu(), ref_E, sampler="ddim")
		ref_out = O.denormalize_tacotron_mel(ref_mel)[:, :, :T]
		assert (mels.cpu() - ref_out).abs().max() < 2e-2                            # log-mel units, range [-11.5, 2.3]; f32 mode
		assert (aux["mel"].cpu() - ref_mel).abs().max() < 3e-3                      # normalised mel in [-1, 1]


def test_stop_token_tail_and_calm_trim_through_the_pipeline(small):
	"""rows that stop early get the 83-fill and the 45,45,248 tail (inference.py:353-366); a long calm run trims the latents (:381-389)."""
	from tortoise_tts_amd.inference import fix_stop_tokens, trim_calm_tokens
	stop = W.AR_SMALL.stop_mel_token
	codes = torch.randint(0, 8000, (3, 20), device=DEV)
	codes[0, 7:] = stop
	codes[2, 15:] = stop
	fixed = fix_stop_tokens(codes, stop)
	assert torch.equal(fixed.cpu(), O.fix_stop_tokens(codes.cpu(), stop))
	assert fixed[0, 7:17].eq(83).all() and fixed[0, -3:].tolist() == [45, 45, 248] and torch.equal(fixed[1, :-3], codes[1, :-3])
	lat = torch.randn(3, 20, 8, device=DEV)
	assert trim_calm_tokens(fixed, lat).shape[1] == 7 + 8 and torch.equal(trim_calm_tokens(fixed, lat).cpu(), O.trim_calm_tokens(fixed.cpu(), lat.cpu()))
	assert trim_calm_tokens(fixed[1:], lat[1:]).shape[1] == 20
	# no stop token anywhere: rows are left as they are (the reference raises on the empty min())
	assert torch.equal(fix_stop_tokens(codes[1:2], stop), codes[1:2])


def test_pipelined_lines_equal_sequential_calls(small):
	tts, _, _ = small
	lines = [_inputs(200 + i, Tt)[0] for i, Tt in enumerate((5, 11, 3))]
	_, al, dl = _inputs(300, 4)
	kw = dict(max_ar_steps=16, max_diffusion_steps=4, candidates=2, suppress_tokens=[W.AR_SMALL.stop_mel_token])
	with torch.inference_mode():
		seq = [tts.inference(t, al.to(DEV), dl.to(DEV), **kw) for t in lines]
		pipe = tts.inference_lines(lines, al.to(DEV), dl.to(DEV), **kw)
	torch.cuda.synchronize()
	assert len(pipe) == len(seq)
	for (m0, s0), (m1, s1, _codes) in zip(seq, pipe):
		assert s0 == s1 and torch.equal(m0, m1)           # same kernels, same inputs, same RNG draws => identical bits


def test_pipelined_lines_cold_with_growing_lengths_and_default_warpers():
	"""a COLD `inference_lines` (fresh handles: every workspace still has to grow, the token-step graph is captured inside the
	call) over lines whose text and mel lengths grow, with the CLI's default warpers (top-k 16, __main__.py:20) and a live stop token.
	The worker thread's hipMalloc / hipFree must not break the capture on the main thread (thread-local capture mode; one graph serves
	every text length), a failure in the worker must surface here, and the results must equal the sequential calls."""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.diffusion import DiffusionTTS
	from tortoise_tts_amd.inference import TTSHotPath
	asd = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 31)
	asd["mel_head.bias"] = asd["mel_head.bias"].clone()
	asd["mel_head.bias"][W.AR_SMALL.stop_mel_token] = 3.0               # rows end on their own, at different lengths per line
	dsd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32)

	def fresh():
		return TTSHotPath(UnifiedVoice(asd, W.AR_SMALL, dtype="f32", device=DEV, max_batch=4, max_ctx=160),
						  DiffusionTTS(dsd, W.DIFF_SMALL, dtype="f32", device=DEV))
	lines = [_inputs(600 + i, Tt)[0] for i, Tt in enumerate((3, 9, 21, 40))]
	_, al, dl = _inputs(700, 4)
	kw = dict(max_ar_steps=60, max_diffusion_steps=3, candidates=3, top_k=16)
	with torch.inference_mode():
		pipe = fresh().inference_lines(lines, al.to(DEV), dl.to(DEV), **kw)          # raises if the worker's future holds an exception
		tts = fresh()
		seq = [tts.inference(t, al.to(DEV), dl.to(DEV), **kw) for t in lines]
	torch.cuda.synchronize()
	assert len({m.shape[-1] for m, _ in seq}) > 1                       # the lines really have different lengths
	for (m0, s0), (m1, s1, _codes) in zip(seq, pipe):
		assert s0 == s1 and torch.equal(m0, m1)
	assert len(tts.autoregressive._states) == 1                         # one generation state / captured step for all four text lengths


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_candidate_shards_are_the_rows_of_the_unsharded_run(dtype):
	"""RNG contract of tortoise_tts_amd/dist.py: every shard of a candidate-sharded `inference_speech` equals, bit for bit, its rows of the
	single-GPU call for all candidates (stop token live, so shards end at different steps and are padded), and after `align_rng` the
	generator stands where the unsharded loop leaves it -- so the winner's diffusion noise is the same too."""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.inference import HotPathStages, TTSHotPath
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 31)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][cfg.stop_mel_token] += 5.0
	ar = UnifiedVoice(sd, cfg, dtype=dtype, device=DEV, max_batch=8, max_ctx=128)
	text, al, _ = _inputs(800, 6)
	C = 7
	kw = dict(do_sample=True, temperature=0.8, top_k=16, max_generate_length=50)
	with torch.inference_mode():
		full = ar.inference_speech(al.to(DEV), text.to(DEV), num_return_sequences=C, **kw)
		after_full = torch.rand(5, device=DEV)
		L = full.shape[1]
		for world in (2, 3):
			from tortoise_tts_amd.dist import candidate_shard
			for r in range(world):
				lo, hi = candidate_shard(C, r, world)
				part = ar.inference_speech(al.to(DEV), text.to(DEV), num_return_sequences=C, candidate_shard=(lo, hi), **kw)
				assert part.shape[0] == hi - lo and part.shape[1] <= L
				assert torch.equal(part, full[lo:hi, :part.shape[1]]), (world, r)
				assert bool((full[lo:hi, part.shape[1]:] == cfg.stop_mel_token).all())       # what the gather pads with
				st = HotPathStages(TTSHotPath(ar, None), text, al.to(DEV), None, ar_temp=0.8, top_k=16, max_ar_steps=50)
				st.align_rng(L)
				assert torch.equal(torch.rand(5, device=DEV), after_full), (world, r)
		with pytest.raises(ValueError):
			ar.inference_speech(al.to(DEV), text.to(DEV), num_return_sequences=C, candidate_shard=(3, 3), **kw)


def test_inference_sharded_equals_inference_on_one_rank(small):
	"""`TTSHotPath.inference_sharded` through a real (1-rank, RCCL) process group: same mel, ids and candidate choice as `inference`."""
	import torch.distributed as dist
	import clvp_oracle  # noqa: F401  (only to fail early if the oracle tree is missing)
	from tortoise_tts_amd.clvp import CLVP
	from tortoise_tts_amd.inference import TTSHotPath
	tts, _, _ = small
	ccfg = W.CLVPConfig(dim=128, depth=2, heads=2, num_speech_tokens=8194)
	full = TTSHotPath(tts.autoregressive, tts.diffusion, clvp=CLVP(W.synth_state_dict(W.clvp_shapes(ccfg), 34), ccfg, dtype="f32", device=DEV))
	text, al, dl = _inputs(900, 8)
	kw = dict(max_ar_steps=14, max_diffusion_steps=3, candidates=5, top_k=16, suppress_tokens=[W.AR_SMALL.stop_mel_token])
	import os, socket
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
	dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV))
	try:
		with torch.inference_mode():
			m0, s0, a0 = full.inference(text, al.to(DEV), dl.to(DEV), return_all=True, **kw)
			m1, s1, a1 = full.inference_sharded(text, al.to(DEV), dl.to(DEV), return_all=True, **kw)
		torch.cuda.synchronize()
	finally:
		dist.destroy_process_group()
	assert s0 == s1 and torch.equal(m0, m1) and torch.equal(a0["codes"], a1["codes"]) and a0["best"] == a1["best"]
	assert torch.equal(a0["scores"], a1["scores"])


def test_sharded_lines_diffused_together_equal_the_per_line_calls(small):
	"""`TTSHotPath.inference_sharded_lines` (configs[3]: the lines of one text, their winners diffused as ONE ragged DDIM batch) through a 1-rank
	RCCL group: every line's mel, ids, scores and choice equal its own `inference_sharded` call -- lines of different length, stop token live so
	the mel lengths differ too"""
	import torch.distributed as dist
	from tortoise_tts_amd.clvp import CLVP
	from tortoise_tts_amd.inference import TTSHotPath
	tts, _, _ = small
	ccfg = W.CLVPConfig(dim=128, depth=2, heads=2, num_speech_tokens=8194)
	full = TTSHotPath(tts.autoregressive, tts.diffusion, clvp=CLVP(W.synth_state_dict(W.clvp_shapes(ccfg), 34), ccfg, dtype="f32", device=DEV))
	g = torch.Generator().manual_seed(901)
	lines = [torch.randint(1, 255, (1, n), generator=g) for n in (8, 5, 11)]
	_, al, dl = _inputs(900, 8)
	kws = [dict(max_ar_steps=m, max_diffusion_steps=3, candidates=5, top_k=16, suppress_tokens=[W.AR_SMALL.stop_mel_token]) for m in (14,)]
	import os, socket
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
	dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(DEV))
	try:
		with torch.inference_mode():
			single = [full.inference_sharded(t, al.to(DEV), dl.to(DEV), return_all=True, **kws[0]) for t in lines]
			marks = []
			batch = full.inference_sharded_lines(lines, al.to(DEV), dl.to(DEV), return_all=True, phase_marks=marks, **kws[0])
		torch.cuda.synchronize()
	finally:
		dist.destroy_process_group()
	assert len(batch) == 3
	for (m0, s0, a0), (m1, s1, a1) in zip(single, batch):
		assert s0 == s1 and torch.equal(m0, m1) and torch.equal(a0["codes"], a1["codes"]) and a0["best"] == a1["best"] and torch.equal(a0["scores"], a1["scores"])
	# phase marks: every line has its sampling and latent-pass marks, the shared diffusion ends the last line's list
	# ("_before_ddim" brackets what lies between a line's latent pass and the shared diffusion: bench.phase_roofline keeps "_" marks out of its phases)
	assert marks[2][-1][2] == 3                                      # the shared diffusion's mark says how many lines it served
	assert [[m[0] for m in lm] for lm in marks] == [["start", "ar_decode", "latent_pass"]] * 2 + [["start", "ar_decode", "latent_pass", "_before_ddim", "ddim"]]


def test_tokens_to_waveform_with_the_vocoder(small):
	"""text tokens + latents -> mel (hot path) -> waveform (BigVGAN on libttk), against the same chain through the two oracles"""
	import bigvgan_oracle as BO
	from tortoise_tts_amd.inference import TTSHotPath
	from tortoise_tts_amd.vocoder import BigVGAN
	tts, aro, dor = small
	vsd = W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), 33)
	full = TTSHotPath(tts.autoregressive, tts.diffusion, BigVGAN(vsd, W.VOC_SMALL, dtype="f32", device=DEV))
	text, al, dl = _inputs(400, 6)
	kw = dict(max_ar_steps=12, max_diffusion_steps=3, candidates=2, suppress_tokens=[W.AR_SMALL.stop_mel_token])
	with torch.inference_mode():
		wav, sr = full.inference_to_wav(text, al.to(DEV), dl.to(DEV), **kw)
		mels, _ = full.inference(text, al.to(DEV), dl.to(DEV), **kw)
		ref = BO.BigVGANOracle(vsd, W.VOC_SMALL).inference(mels.cpu())
	T = O.mel_frames_for(12)
	assert sr == 24000 and wav.shape == (1, 1, T * W.VOC_SMALL.hop_size) and (wav.cpu() - ref).abs().max() < 1e-4
	with pytest.raises(ValueError):
		tts.inference_to_wav(text, al.to(DEV), dl.to(DEV), **kw)


def test_clvp_picks_the_candidate_that_is_diffused(small):
	"""with a CLVP model the best-scoring candidate's latents go to the diffusion (trimmed by its own codes); scores equal the oracle's"""
	import clvp_oracle as CO
	from tortoise_tts_amd.clvp import CLVP
	from tortoise_tts_amd.inference import TTSHotPath, trim_calm_tokens
	tts, aro, dor = small
	ccfg = W.CLVPConfig(dim=128, depth=2, heads=2, num_speech_tokens=8194)      # table wide enough for any sampled mel id
	csd = W.synth_state_dict(W.clvp_shapes(ccfg), 34)
	full = TTSHotPath(tts.autoregressive, tts.diffusion, clvp=CLVP(csd, ccfg, dtype="f32", device=DEV))
	text, al, dl = _inputs(500, 7)
	kw = dict(max_ar_steps=14, max_diffusion_steps=2, candidates=6, suppress_tokens=[W.AR_SMALL.stop_mel_token])
	with torch.inference_mode():
		mels, _, aux = full.inference(text, al.to(DEV), dl.to(DEV), return_all=True, **kw)
		base_mels, _, base = tts.inference(text, al.to(DEV), dl.to(DEV), return_all=True, **kw)
		ref_scores = CO.CLVPOracle(csd, ccfg).forward(text.repeat(6, 1), aux["codes"].cpu())
	assert torch.equal(aux["codes"], base["codes"]) and base["best"] == 0 and base["scores"] is None
	assert (aux["scores"].cpu() - ref_scores).abs().max() < 1e-4 and aux["best"] == int(ref_scores.argmax())
	lat_all = tts.autoregressive.forward(al.to(DEV).expand(6, -1), text.to(DEV).expand(6, -1), torch.tensor([7] * 6), aux["codes"], torch.tensor([14 * 1024] * 6),
										 return_latent=True, clip_inputs=False)
	b = aux["best"]
	assert torch.equal(aux["latents"], trim_calm_tokens(aux["codes"][b:b + 1], lat_all[b:b + 1]))
	if b != 0:
		assert not torch.equal(mels, base_mels)


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_latents_for_the_winner_only_gives_the_same_bits(dtype):
	"""the k = 1 variant of SURVEY.md 8d row 2: choosing the candidate first and running the dense latent pass on its row alone returns the
	mel of the all-candidates pass bit for bit (rows of the pass are independent), with and without a scorer"""
	import clvp_oracle  # noqa: F401  (oracle/ on the path: fixtures of this module)
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.clvp import CLVP
	from tortoise_tts_amd.diffusion import DiffusionTTS
	from tortoise_tts_amd.inference import TTSHotPath
	ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 31), W.AR_SMALL, dtype=dtype, device=DEV, max_batch=8, max_ctx=128)
	df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32), W.DIFF_SMALL, dtype=dtype, device=DEV)
	ccfg = W.CLVPConfig(dim=128, depth=2, heads=2, num_speech_tokens=8194)
	clvp = CLVP(W.synth_state_dict(W.clvp_shapes(ccfg), 34), ccfg, dtype="f32", device=DEV)
	text, al, dl = _inputs(900, 8)
	kw = dict(max_ar_steps=18, max_diffusion_steps=3, candidates=7, suppress_tokens=[W.AR_SMALL.stop_mel_token], return_all=True)
	for tts in (TTSHotPath(ar, df), TTSHotPath(ar, df, clvp=clvp)):
		with torch.inference_mode():
			m_all, s_all, a_all = tts.inference(text, al.to(DEV), dl.to(DEV), **kw)
			m_one, s_one, a_one = tts.inference(text, al.to(DEV), dl.to(DEV), latents_for="winner", **kw)
		assert s_all == s_one and a_all["best"] == a_one["best"] and torch.equal(a_all["codes"], a_one["codes"])
		assert torch.equal(a_all["latents"], a_one["latents"]) and torch.equal(m_all, m_one)
	with pytest.raises(ValueError):
		TTSHotPath(ar, df).inference(text, al.to(DEV), dl.to(DEV), latents_for="some", **kw)


def test_pipelined_lines_with_a_scorer_equal_sequential_calls(small):
	"""`inference_lines` (lines sampled as one decode batch, diffusion pipelined) picks each line's candidate with CLVP as `inference` does"""
	from tortoise_tts_amd.clvp import CLVP
	from tortoise_tts_amd.inference import TTSHotPath
	tts0, _, _ = small
	ccfg = W.CLVPConfig(dim=128, depth=2, heads=2, num_speech_tokens=8194)
	tts = TTSHotPath(tts0.autoregressive, tts0.diffusion, clvp=CLVP(W.synth_state_dict(W.clvp_shapes(ccfg), 34), ccfg, dtype="f32", device=DEV))
	lines = [_inputs(400 + i, Tt)[0] for i, Tt in enumerate((6, 13, 4))]
	_, al, dl = _inputs(450, 4)
	kw = dict(max_ar_steps=14, max_diffusion_steps=3, candidates=2, suppress_tokens=[W.AR_SMALL.stop_mel_token])
	with torch.inference_mode():
		seq = [tts.inference(t, al.to(DEV), dl.to(DEV), return_all=True, **kw) for t in lines]
		pipe = tts.inference_lines(lines, al.to(DEV), dl.to(DEV), **kw)
	torch.cuda.synchronize()
	for (m0, s0, _a), (m1, s1, _c) in zip(seq, pipe):
		assert s0 == s1 and torch.equal(m0, m1)
