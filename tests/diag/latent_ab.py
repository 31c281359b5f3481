"""Diagnostic: wall time of the dense latent pass (forward(return_latent=True), 16 candidates x 250 codes, 64 text tokens, bf16) for A/B runs on ONE box:
   python tests/diag/latent_ab.py [reps]          (env TTK_GEMM_TILE_WIDE / TTK_LIB ... select the variant)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.autoregressive import UnifiedVoice
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL, dtype="bf16", device=dev, max_batch=16, max_ctx=64 + 4 + 250 + 8)
g = torch.Generator().manual_seed(1234)
text = torch.randint(1, 255, (1, 64), generator=g).to(dev)
cond = torch.randn(1, 1024, generator=g).to(dev)
codes = torch.randint(0, 8192, (16, 250), generator=g).to(dev)
run = lambda: ar.forward(cond.expand(16, -1), text.expand(16, -1), torch.tensor([64] * 16), codes, torch.tensor([250 * 1024] * 16), return_latent=True, clip_inputs=False)
with torch.inference_mode():
	run(); run(); torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TTK_")) + f": min {min(ts):.3f} ms  median {sorted(ts)[len(ts) // 2]:.3f} ms", flush=True)
