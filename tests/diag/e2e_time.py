"""Diagnostic: text tokens + conditioning latents -> waveform at the benchmark's shape with every stage on libttk
(AR sampling, latent pass, CLVP scoring, DDIM, BigVGAN), bf16."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.autoregressive import UnifiedVoice
from tortoise_tts_amd.clvp import CLVP
from tortoise_tts_amd.diffusion import DiffusionTTS
from tortoise_tts_amd.inference import TTSHotPath
from tortoise_tts_amd.vocoder import BigVGAN
dev = "cuda:0"
ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL, dtype="bf16", device=dev, max_batch=16, max_ctx=64 + 4 + 250 + 8)
df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype="bf16", device=dev)
ccfg = W.CLVPConfig(num_speech_tokens=8194)       # synthetic AR weights can emit any mel id
cl = CLVP(W.synth_state_dict(W.clvp_shapes(ccfg), 0), ccfg, dtype="bf16", device=dev)
vo = BigVGAN(W.synth_state_dict(W.vocoder_shapes(W.VOC_FULL), 0), W.VOC_FULL, dtype="bf16", device=dev)
tts = TTSHotPath(ar, df, vocoder=vo, clvp=cl)
g = torch.Generator().manual_seed(1234)
text = torch.randint(1, 255, (1, 64), generator=g).to(dev)
al, dl = torch.randn(1, 1024, generator=g).to(dev), torch.randn(1, 2048, generator=g).to(dev)
kw = dict(max_ar_steps=250, max_diffusion_steps=80, ar_temp=0.8, candidates=16, suppress_tokens=[8193])
wav, sr = tts.inference_to_wav(text, al, dl, **kw); torch.cuda.synchronize()
ts = []
for _ in range(3):
	t0 = time.perf_counter(); wav, sr = tts.inference_to_wav(text, al, dl, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
sec = wav.shape[-1] / sr
print(f"tokens -> waveform: {1e3 * min(ts):.1f} ms for {sec:.2f} s of audio ({sec / min(ts):.1f} audio-s/wall-s), wav {tuple(wav.shape)} @ {sr} Hz", flush=True)
