"""Diagnostic driver for rocprofv3 --pmc: a few DDIM steps at configs[1] size (bf16)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
dev = "cuda:0"
df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype="bf16", device=dev)
g = torch.Generator().manual_seed(1)
T = 1088
E = torch.randn(1, 1024, T, generator=g).to(dev)
noise = torch.randn(1, 100, T, generator=g).to(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with torch.inference_mode():
	get_diffuser(steps, True).sample_loop(df, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E})
torch.cuda.synchronize()
print("done")
