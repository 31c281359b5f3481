"""Diagnostic: wall time of the 80-step DDIM loop at configs[1] size (bf16, T=1088), repeated, for A/B runs on ONE box:
   python tests/diag/ddim_ab.py [reps]          (env TTK_LIB / TTK_GEMM_TILE / ... select the variant)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype=os.environ.get("TTK_DDIM_DTYPE", "bf16"), device=dev)
g = torch.Generator().manual_seed(1)
T = int(os.environ.get("TTK_AB_T", "1088"))      # frames (TTK_AB_T=2176: a configs[3] line)
E = torch.randn(1, 1024, T, generator=g).to(dev)
noise = torch.randn(1, 100, T, generator=g).to(dev)
STEPS = int(os.environ.get("TTK_AB_STEPS", "80"))
run = lambda: get_diffuser(STEPS, True).sample_loop(df, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E})
with torch.inference_mode():
	run(); torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(f"{os.environ.get('TTK_LIB', 'libttk.so').split('/')[-1]} " + " ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TTK_") and k != "TTK_LIB") + f": min {min(ts):.2f} ms  median {sorted(ts)[len(ts) // 2]:.2f} ms", flush=True)
