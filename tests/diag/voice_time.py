"""Diagnostic: wall time of the voice front-end at full size (bf16 encoders, f32 front-ends) for one 6 s reference clip:
   python tests/diag/voice_time.py"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W, mel as M
from tortoise_tts_amd.conditioning import ConditioningEncoder, ContextualEmbedder
dev = "cuda:0"
enc = ConditioningEncoder(W.synth_state_dict(W.ar_conditioning_shapes(W.AR_FULL), 1), W.AR_FULL, dtype="bf16", device=dev)
ctx = ContextualEmbedder(W.synth_state_dict(W.diffusion_conditioning_shapes(W.DIFF_FULL), 2), W.DIFF_FULL, dtype="bf16", device=dev)
tms = M.TorchMelSpectrogram(mel_norms=torch.ones(80), device=dev)
stft = M.TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000, device=dev)
wav = (torch.randn(1, 6 * 44100, generator=torch.Generator().manual_seed(0)) * 0.1).to(dev)

def timed(name, fn, reps=20):
	fn(); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(reps):
		out = fn()
	torch.cuda.synchronize()
	print(f"{name:34s} {1e3 * (time.perf_counter() - t0) / reps:8.3f} ms", flush=True)
	return out

with torch.inference_mode():
	w22 = timed("resample 44.1k -> 22.05k (6 s)", lambda: M.resample(wav, 44100, 22050, device=dev))
	ar_mel = timed("AR mel (132300 samples)", lambda: M.format_autoregressive_conditioning(w22, tms))
	df_mel = timed("resample + diffusion mel (102400)", lambda: M.format_diffusion_conditioning(w22, stft))
	timed("ConditioningEncoder (517 frames)", lambda: enc.get_conditioning(ar_mel[:, None]))
	timed("ContextualEmbedder (401 frames)", lambda: ctx.get_conditioning(df_mel[:, None]))
	timed("encode() whole voice", lambda: M.encode(wav, 44100, tms=tms, stft=stft, conditioning_encoder=enc, contextual_embedder=ctx))
