"""BigVGAN vocoder on libttk against the reference's waveform (tests/golden/vocoder_small.npz) and the oracle.
GPU only; calls go through the C ABI (`ttk_voc_*`)."""
import numpy as np
import pytest
import torch

import bigvgan_oracle as BO
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
	return torch.from_numpy(np.asarray(a))


def maxerr(a, b):
	return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


def test_small_f32_equals_reference_waveform(golden):
	from tortoise_tts_amd.vocoder import BigVGAN
	g = golden("vocoder_small")
	sd = W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), int(g["seed"]))
	voc = BigVGAN(sd, W.VOC_SMALL, dtype="f32", device=DEV)
	audio = voc.inference(t(g["mel"]).to(DEV))
	assert audio.shape == g["audio"].shape and audio.dtype == torch.float32
	assert maxerr(audio, g["audio"]) < 1e-4                       # waveform in [-1, 1], fp32 mode, vs the REFERENCE class


@pytest.mark.parametrize("B,T", [(1, 1), (1, 2), (3, 5), (2, 64)])
def test_small_f32_edge_lengths_vs_oracle(B, T):
	"""sequences shorter than the filters' reach (every replicate / zero padding branch) and a batch"""
	from tortoise_tts_amd.vocoder import BigVGAN
	sd = W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), 52)
	voc = BigVGAN(sd, W.VOC_SMALL, dtype="f32", device=DEV)
	mel = torch.randn(B, 100, T, generator=torch.Generator().manual_seed(T)) * 2 - 5
	with torch.inference_mode():
		ref = BO.BigVGANOracle(sd, W.VOC_SMALL).inference(mel)
	audio = voc.inference(mel.to(DEV))
	assert audio.shape == ref.shape == (B, 1, T * W.VOC_SMALL.hop_size) and maxerr(audio, ref) < 1e-4


def test_small_bf16_tolerance_and_weight_norm_input(golden):
	"""bf16 mode within a stated distance of the reference waveform; the handle also takes the checkpoint's weight_g / weight_v form"""
	from tortoise_tts_amd.vocoder import BigVGAN
	g = golden("vocoder_small")
	sd = W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), int(g["seed"]))
	voc = BigVGAN(sd, W.VOC_SMALL, dtype="bf16", device=DEV)
	audio = voc.inference(t(g["mel"]).to(DEV))
	ref = t(g["audio"])
	rel = ((audio.cpu().double() - ref.double()).norm() / ref.double().norm()).item()
	# synthetic weights drive the output conv far into tanh's steep region, so single samples move more than the signal as a whole
	assert rel < 4e-2 and maxerr(audio, ref) < 0.25, (rel, maxerr(audio, ref))
	# same weights expressed as weight_g / weight_v (v scaled arbitrarily, g carrying the norm)
	wn = {}
	for k, v in sd.items():
		if k.endswith(".weight") and v.dim() == 3:
			norm = v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1)
			wn[k + "_v"] = v * 3.0
			wn[k + "_g"] = norm
		else:
			wn[k] = v
	wn["activation_post.upsample.filter"] = torch.zeros(1, 1, 12)       # constant buffers of the checkpoint are ignored
	voc2 = BigVGAN(wn, W.VOC_SMALL, dtype="f32", device=DEV)
	assert maxerr(voc2.inference(t(g["mel"]).to(DEV)), g["audio"]) < 1e-4


def test_full_size_fp32_short_clip_vs_oracle_and_bf16_utterance():
	"""the published 112 M-parameter configuration: fp32 against the oracle on a clip the CPU can follow (8 frames -> 2048 samples), then a
	configs[1]-length mel (1088 frames, 11.6 s) in bf16: shape, range, determinism."""
	from tortoise_tts_amd.vocoder import BigVGAN
	cfg = W.VOC_FULL
	assert W.n_params(W.vocoder_shapes(cfg)) == 112_414_513 and cfg.hop_size == 256
	sd = W.synth_state_dict(W.vocoder_shapes(cfg), 53)
	mel = torch.randn(1, 100, 8, generator=torch.Generator().manual_seed(9)) * 2 - 5
	with torch.inference_mode():
		ref = BO.BigVGANOracle(sd, cfg).inference(mel)
	voc = BigVGAN(sd, cfg, dtype="f32", device=DEV)
	audio = voc.inference(mel.to(DEV))
	assert audio.shape == ref.shape == (1, 1, 2048) and maxerr(audio, ref) < 2e-4
	del voc
	vb = BigVGAN(sd, cfg, dtype="bf16", device=DEV)
	ab = vb.inference(mel.to(DEV)).cpu()
	assert ((ab.double() - ref.double()).norm() / ref.double().norm()).item() < 5e-2
	long_mel = (torch.randn(1, 100, 1088, generator=torch.Generator().manual_seed(10)) * 2 - 5).to(DEV)
	a = vb.inference(long_mel)
	b = vb.inference(long_mel)
	assert a.shape == (1, 1, 1088 * 256) and torch.isfinite(a).all() and float(a.abs().max()) <= 1.0 and torch.equal(a, b)
	with pytest.raises(Exception, match="mel must be"):
		vb.inference(torch.zeros(1, 80, 4))
