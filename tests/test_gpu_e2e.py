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
