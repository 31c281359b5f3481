"""Diagnostic (not a test): wall time of each phase of the benchmark step."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.autoregressive import UnifiedVoice
from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser

dev = "cuda:0"
dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL, dtype=dtype, device=dev, max_batch=16, max_ctx=64 + 4 + 250 + 8)
df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype=dtype, device=dev)
g = torch.Generator().manual_seed(1234)
text = torch.randint(1, 255, (1, 64), generator=g).to(dev)
cond = torch.randn(1, 1024, generator=g).to(dev)
dcond = torch.randn(1, 2048, generator=g).to(dev)


def timed(name, fn, n=3):
	fn(); torch.cuda.synchronize()
	t0 = time.perf_counter()
	for _ in range(n):
		out = fn()
	torch.cuda.synchronize()
	print(f"{name:28s} {1e3 * (time.perf_counter() - t0) / n:9.2f} ms", flush=True)
	return out


with torch.inference_mode():
	codes = timed("inference_speech (graph)", lambda: ar.inference_speech(cond, text, do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16,
																		 max_generate_length=250, suppress_tokens=[8193]))
	ar.use_graph = False
	timed("inference_speech (eager)", lambda: ar.inference_speech(cond, text, do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16,
																 max_generate_length=250, suppress_tokens=[8193]), n=1)
	ar.use_graph = True
	lat = timed("latent pass B=16", lambda: ar.forward(cond.expand(16, -1), text.expand(16, -1), torch.tensor([64] * 16), codes, torch.tensor([250 * 1024] * 16),
													  return_latent=True, clip_inputs=False))
	T = 250 * 4 * 24000 // 22050
	E = timed("timestep_independent", lambda: df.timestep_independent(lat[:1], dcond, T, False))
	noise = torch.randn(1, 100, T, device=dev)
	timed("ddim 80 steps", lambda: get_diffuser(80, True).sample_loop(df, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E}), n=2)
