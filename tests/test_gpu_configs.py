"""BASELINE.json's other configs as parity cases (only configs[1] is a bench line).

  configs[0]  one short sentence, 1 candidate, 4 diffusion steps, fp32, FULL-SIZE models: the whole hot path against the oracle.
  configs[3]  the per-GPU shard of the long-form case -- 32 candidates, 256 text tokens, 500 mel tokens (ctx 760), T = 2176 frames,
              200 DDIM steps: sizes the oracle can only follow on the small models, so ids (bit-exact) and mel (tolerance) are checked
              there, each stressed dimension at its full value.
GPU only; calls go through the C ABI."""
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
	return torch.Generator().manual_seed(seed)


def test_config0_full_size_fp32_whole_path_vs_oracle():
	"""24 text tokens, 1 candidate, 48 mel tokens with the stop token suppressed (fixed M), T = 208 frames (2.22 s), 4 DDIM steps."""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.diffusion import DiffusionTTS
	from tortoise_tts_amd.inference import TTSHotPath
	asd = W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0)
	dsd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 1)
	tts = TTSHotPath(UnifiedVoice(asd, W.AR_FULL, dtype="f32", device=DEV, max_batch=1, max_ctx=24 + 4 + 48 + 8),
					 DiffusionTTS(dsd, W.DIFF_FULL, dtype="f32", device=DEV))
	text = torch.randint(1, 255, (1, 24), generator=gen(1234))
	al, dl = torch.randn(1, 1024, generator=gen(1235)), torch.randn(1, 2048, generator=gen(1236))
	stop = W.AR_FULL.stop_mel_token
	with torch.inference_mode():
		mels, seconds, aux = tts.inference(text, al.to(DEV), dl.to(DEV), max_ar_steps=48, max_diffusion_steps=4, candidates=1,
										   suppress_tokens=[stop], return_all=True)
		aro, dor = O.AROracle(asd, W.AR_FULL), O.DiffusionOracle(dsd, W.DIFF_FULL)
		ref_ids = O.inference_speech(aro, al, text, num_return_sequences=1, max_generate_length=48, temperature=0.8, top_k=0,
									 sample_device="cuda", suppress_tokens=[stop])
		noise_ref = torch.randn((1, 100, 208), device=DEV)
		assert torch.equal(aux["codes"].cpu(), O.fix_stop_tokens(ref_ids, stop))                    # integer ids: bit-exact
		ref_lat = O.trim_calm_tokens(ref_ids, aro.forward_latents(al, text, ref_ids))[:1]
		assert aux["latents"].shape == ref_lat.shape == (1, 48, 1024)
		assert (aux["latents"].cpu() - ref_lat).abs().max() < 2e-3
		T = O.mel_frames_for(48)
		assert T == 208 and mels.shape == (1, 100, T) and abs(seconds - 2.2187) < 1e-3 and torch.equal(aux["noise"], noise_ref)
		ref_E = dor.timestep_independent(ref_lat, dl, T)
		assert (aux["E"].cpu() - ref_E).abs().max() < 2e-3
		ref_mel = O.SpacedSchedule(steps=4, cond_free=True).sample_loop(dor, noise_ref.cpu(), ref_E, sampler="ddim")
		assert (aux["mel"].cpu() - ref_mel).abs().max() < 5e-3                                      # normalised mel in [-1, 1], fp32 mode
		assert (mels.cpu() - O.denormalize_tacotron_mel(ref_mel)[:, :, :T]).abs().max() < 4e-2     # log-mel units, range [-11.5, 2.3]


def test_config3_shard_lengths_ids_bit_exact_small_model():
	"""32 candidates x 500 mel tokens after a 256-token text line (cache length up to 760), stop token live: rows finish on their own."""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 41)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][cfg.stop_mel_token] = -2.0          # rare stops: most rows run the full 500 tokens, a few end early
	model = UnifiedVoice(sd, cfg, dtype="f32", device=DEV, max_batch=32, max_ctx=256 + 4 + 500 + 8)
	text = torch.randint(1, 255, (1, 256), generator=gen(7))
	cond = torch.randn(1, cfg.model_dim, generator=gen(8))
	kw = dict(num_return_sequences=32, max_generate_length=500, temperature=0.8, top_k=0, top_p=1.0, repetition_penalty=1.0)
	with torch.inference_mode():
		ref = O.inference_speech(O.AROracle(sd, cfg), cond, text, sample_device="cuda", **kw)
		after_ref = torch.rand(4, device=DEV)
		got = model.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, **kw)
		after = torch.rand(4, device=DEV)
	assert got.shape == ref.shape == (32, 500) and torch.equal(got.cpu(), ref) and torch.equal(after, after_ref)
	lat = model.forward(cond.to(DEV).expand(32, -1), text.to(DEV).expand(32, -1), torch.tensor([256] * 32), got, torch.tensor([500 * 1024] * 32),
						return_latent=True, clip_inputs=False)
	with torch.inference_mode():
		ref_lat = O.AROracle(sd, cfg).forward_latents(cond.repeat(2, 1), text.repeat(2, 1), ref[:2])
	assert lat.shape == (32, 500, cfg.model_dim) and (lat[:2].cpu() - ref_lat).abs().max() < 1e-3


def test_config3_long_mel_and_200_steps_small_model():
	"""T = 2176 frames (500 mel tokens; the relative-position bias saturates 17x over) for 6 steps, and the 200-step schedule at a short T."""
	from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
	cfg = W.DIFF_SMALL
	sd = W.synth_state_dict(W.diffusion_shapes(cfg), 42)
	model, dor = DiffusionTTS(sd, cfg, dtype="f32", device=DEV), O.DiffusionOracle(sd, cfg)
	dcond = torch.randn(1, 2 * cfg.model_channels, generator=gen(3))
	for M, steps, tol in ((500, 6, 3e-3), (31, 200, 2e-2)):
		T = O.mel_frames_for(M)
		lat = torch.randn(1, M, cfg.in_latent_channels, generator=gen(4))
		noise = torch.randn(1, 100, T, generator=gen(5))
		with torch.inference_mode():
			E = model.timestep_independent(lat.to(DEV), dcond.to(DEV), T, False)
			ref_E = dor.timestep_independent(lat, dcond, T)
			assert (E.cpu() - ref_E).abs().max() < 1e-3
			mel = get_diffuser(steps=steps, cond_free=True).sample_loop(model, (1, 100, T), sampler="ddim", noise=noise.to(DEV),
																		model_kwargs={"precomputed_aligned_embeddings": E})
			ref = O.SpacedSchedule(steps=steps, cond_free=True).sample_loop(dor, noise, ref_E, sampler="ddim")
		assert mel.shape == (1, 100, T) and (mel.cpu() - ref).abs().max() < tol, (M, steps, float((mel.cpu() - ref).abs().max()))
