"""Diagnostic: digest of the diffusion network's outputs under the library TTK_LIB selects -- one evaluation and a 3-step DDIM loop per (dtype, T) -- so that two builds
of libttk can be compared bit for bit on ONE box:   TTK_LIB=a.so python tests/diag/lib_bits.py > a.txt;  TTK_LIB=b.so python tests/diag/lib_bits.py > b.txt;  diff a.txt b.txt"""
import hashlib, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
dev = "cuda:0"
sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 1)
cases = [("bf16", 1088), ("bf16", 1000), ("f16", 1088), ("bf16", 320), ("bf16", 1216), ("bf16", 2176), ("f32", 320)]
if len(sys.argv) > 1: cases = [(c.split(":")[0], int(c.split(":")[1])) for c in sys.argv[1:]]
for dtype, T in cases:
	m = DiffusionTTS(sd, W.DIFF_FULL, dtype=dtype, device=dev)
	noise = torch.randn(1, 100, T, generator=torch.Generator().manual_seed(3)).to(dev)
	E = torch.randn(1, 1024, T, generator=torch.Generator().manual_seed(4)).to(dev)
	with torch.inference_mode():
		y = m(noise, torch.tensor([900], device=dev), precomputed_aligned_embeddings=E)
		mel = get_diffuser(steps=3, cond_free=True).sample_loop(m, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E})
	torch.cuda.synchronize()
	h = lambda t: hashlib.sha256(t.float().cpu().numpy().tobytes()).hexdigest()[:16]
	print(f"{dtype} T={T}: eval {h(y)} ddim3 {h(mel)} finite {bool(torch.isfinite(mel).all())}", flush=True)
	del m
