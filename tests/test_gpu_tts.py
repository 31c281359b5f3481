"""Text + reference clip -> waveform through `tortoise_tts_amd.TTS` (the reference's `TTS.inference`, inference.py:142-425, BigVGAN branch):
the conditioning chain (resample, mel front-ends, encoders) against the oracles, and the assembled call against its own parts.
GPU only; every device stage goes through the C ABI."""
import math
import random

import numpy as np
import pytest
import torch

import cond_oracle as CO
import mel_oracle as MO
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def speechlike(seed, n, sr):
	g = torch.Generator().manual_seed(seed)
	t = torch.arange(n) / sr
	y = 0.3 * torch.sin(2 * math.pi * 180 * t) + 0.15 * torch.sin(2 * math.pi * 1250 * t + 1.0) + 0.05 * torch.sin(2 * math.pi * 4100 * t) + 0.02 * torch.randn(n, generator=g)
	return (y * (0.5 + 0.5 * torch.sin(2 * math.pi * 3 * t)))[None]


@pytest.mark.parametrize("orig,new,n", [(22050, 24000, 30001), (44100, 22050, 50000), (16000, 22050, 12345), (48000, 22050, 4097), (22050, 24000, 1)])
def test_resample_vs_oracle(orig, new, n):
	from tortoise_tts_amd.mel import resample
	y = speechlike(n, n, orig).repeat(2, 1) * torch.tensor([[1.0], [-0.5]])
	got = resample(y.to(DEV), orig, new, device=DEV).cpu()
	ref = MO.resample(y, orig, new)
	assert got.shape == ref.shape == (2, math.ceil(new * n / orig))
	# the kernel bank is built in f32 like torchaudio.functional.resample builds it for f32 audio; the oracle's is float64: 1e-5 absolute
	assert (got - ref).abs().max().item() < 1e-5 * max(1.0, ref.abs().max().item())
	assert resample(y, 22050, 22050) is y


@pytest.fixture(scope="module")
def parts(golden):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.conditioning import ConditioningEncoder, ContextualEmbedder
	from tortoise_tts_amd.diffusion import DiffusionTTS
	from tortoise_tts_amd.mel import TacotronSTFT, TorchMelSpectrogram
	from tortoise_tts_amd.tokenizer import VoiceBpeTokenizer
	from tortoise_tts_amd.tts import TTS
	from tortoise_tts_amd.vocoder import BigVGAN
	g = golden("tokenizer")
	tok = VoiceBpeTokenizer(vocab={str(t): i for i, t in enumerate(g["vocab"])}, merges=[str(m) for m in g["merges"]], special_tokens=[str(s) for s in g["special"]])
	sd = dict(ar=W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 31), df=W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32),
			  voc=W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), 33), arc=W.synth_state_dict(W.ar_conditioning_shapes(W.AR_SMALL), 35),
			  dfc=W.synth_state_dict(W.diffusion_conditioning_shapes(W.DIFF_SMALL), 36))
	norms = torch.rand(80, generator=torch.Generator().manual_seed(2)) * 3 + 1
	tts = TTS(UnifiedVoice(sd["ar"], W.AR_SMALL, dtype="f32", device=DEV, max_batch=8, max_ctx=128), DiffusionTTS(sd["df"], W.DIFF_SMALL, dtype="f32", device=DEV), tok,
			  vocoder=BigVGAN(sd["voc"], W.VOC_SMALL, dtype="f32", device=DEV),
			  conditioning_encoder=ConditioningEncoder(sd["arc"], W.AR_SMALL, dtype="f32", device=DEV), contextual_embedder=ContextualEmbedder(sd["dfc"], W.DIFF_SMALL, dtype="f32", device=DEV),
			  tms=TorchMelSpectrogram(mel_norms=norms, device=DEV), stft=TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000, device=DEV))
	return tts, sd, norms


def oracle_latents(sd, norms, wav22):
	"""emb/mel.py:50-109 through the oracles: 132300-sample AR clip, 102400-sample 24 kHz diffusion clip"""
	ar_wav = torch.nn.functional.pad(wav22, (0, 132300 - wav22.shape[-1])) if wav22.shape[-1] < 132300 else wav22
	ar_mel = MO.torch_mel_spectrogram(ar_wav, norms)
	w24 = MO.resample(wav22, 22050, 24000)
	w24 = torch.nn.functional.pad(w24, (0, 102400 - w24.shape[-1])) if w24.shape[-1] < 102400 else w24[..., :102400]
	df_mel = MO.tacotron_mel(w24)
	with torch.inference_mode():
		return (ar_mel, df_mel, CO.ar_get_conditioning(sd["arc"], ar_mel[:, None], W.AR_SMALL.heads),
				CO.diffusion_get_conditioning(sd["dfc"], df_mel[:, None], W.DIFF_SMALL.num_heads))


@pytest.mark.parametrize("sr,n", [(22050, 40000), (44100, 70000)])
def test_encode_audio_vs_oracle_chain(parts, sr, n):
	tts, sd, norms = parts
	wav = speechlike(7, n, sr)
	enc = tts.encode_audio(wav.to(DEV), sr)
	wav22 = MO.resample(wav, sr, 22050)
	ar_mel, df_mel, ar_lat, df_lat = oracle_latents(sd, norms, wav22)
	ar_c, df_c = enc["conds"]
	assert ar_c.shape == (1, 1, 80, 132300 // 256 + 1) and df_c.shape == (1, 1, 100, 102400 // 256 + 1)
	assert (ar_c[0].cpu() * norms[None, :, None] - ar_mel * norms[None, :, None]).exp().sub(1).abs().max() < 1.0   # finite, same scale
	assert (ar_c[0, 0].cpu() - ar_mel[0]).abs().median().item() < 1e-4 and (df_c[0, 0].cpu() - df_mel[0]).abs().median().item() < 1e-4
	a, d = enc["latent"]
	assert a.shape == (1, W.AR_SMALL.model_dim) and d.shape == (1, 2 * W.DIFF_SMALL.model_channels)
	assert (a.cpu() - ar_lat).abs().max().item() < 2e-3 * max(1.0, ar_lat.abs().max().item())
	assert (d.cpu() - df_lat).abs().max().item() < 2e-3 * max(1.0, df_lat.abs().max().item())
	assert enc["metadata"] == {"original_length": n, "sample_rate": sr, "duration": n / sr}
	assert tts.encode_audio(enc) is enc


def test_long_clip_is_cropped_with_the_given_rng(parts):
	from tortoise_tts_amd.mel import format_autoregressive_conditioning
	tts, sd, norms = parts
	wav = speechlike(8, 150000, 22050).to(DEV)
	m1 = format_autoregressive_conditioning(wav, tts.tms, rng=random.Random(3))
	start = random.Random(3).randint(0, 150000 - 132300)
	ref = MO.torch_mel_spectrogram(wav.cpu()[:, start:start + 132300], norms)
	assert m1.shape == (1, 80, 517) and (m1.cpu() - ref).abs().median().item() < 1e-4
	assert format_autoregressive_conditioning(wav, tts.tms, cond_length=0).shape == (1, 80, 150000 // 256 + 1)


def test_text_and_clip_to_waveform(parts, golden):
	"""two lines of text and a clip in, one waveform out; equal to tokenising, encoding and running the hot path by hand with the same seed"""
	tts, sd, norms = parts
	wav = speechlike(9, 30000, 22050)
	text = "Hello there, Mr. Fox.\nThe end!"
	kw = dict(max_ar_steps=10, max_diffusion_steps=3, candidates=2)
	out, sr = tts.inference(text, wav.to(DEV), seed=1234, **kw)
	enc = tts.encode_audio(wav.to(DEV), 22050)
	ids = [tts.encode_text(line) for line in text.split("\n")]
	assert ids[0].tolist() == tts.tokenizer.encode("hello there, mister fox.") and ids[0].dtype == torch.int64 and int(ids[0].max()) < 255
	from tortoise_tts_amd.tts import set_seed
	set_seed(1234)
	by_hand = [tts.hot.inference_to_wav(i.to(DEV)[None], enc["latent"][0], enc["latent"][1], **kw)[0] for i in ids]
	assert sr == 24000 and out.dim() == 3 and out.shape[:2] == (1, 1)
	assert out.shape[-1] == sum(w.shape[-1] for w in by_hand) and torch.equal(out, torch.concat(by_hand, dim=-1))
	assert torch.isfinite(out).all() and float(out.abs().max()) <= 1.0
	out2, _ = tts.inference(text, enc, seed=1234, **kw)                 # a precomputed voice dict instead of the clip
	assert torch.equal(out, out2)
	with pytest.raises(NotImplementedError):
		tts.inference(text, enc, vocoder_type="hifigan")
	with pytest.raises(ValueError, match="empty line"):
		tts.inference("", enc)
