"""Diagnostic (not a test): decode time per layer when all weights fit the 256 MiB Infinity Cache (8 layers) vs not (30)."""
import sys, os, time, dataclasses
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.autoregressive import UnifiedVoice
dev = "cuda:0"
for layers in (4, 8, 30):
	cfg = dataclasses.replace(W.AR_FULL, layers=layers)
	ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(cfg), 0), cfg, dtype="bf16", device=dev, max_batch=16, max_ctx=64 + 4 + 250 + 8)
	g = torch.Generator().manual_seed(1234)
	text = torch.randint(1, 255, (1, 64), generator=g).to(dev)
	cond = torch.randn(1, 1024, generator=g).to(dev)
	f = lambda: ar.inference_speech(cond, text, do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16, max_generate_length=250, suppress_tokens=[8193])
	with torch.inference_mode():
		f(); torch.cuda.synchronize()
		t0 = time.perf_counter(); f(); torch.cuda.synchronize()
		dt = time.perf_counter() - t0
	print(f"layers={layers:3d}  {dt * 1e3:8.2f} ms  -> {dt / 250 * 1e6:8.1f} us/token", flush=True)
	del ar
