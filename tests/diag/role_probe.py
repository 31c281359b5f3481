"""Diagnostic: which GEMM role (TTK_GEMM_ROLE bit mask) changes the bits of one network evaluation against the generic kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS
sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 1)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1088
g = torch.Generator().manual_seed(3)
x = torch.randn(1, 100, T, generator=g).cuda(); E = torch.randn(1, 1024, T, generator=g).cuda(); t = torch.tensor([900]).cuda()
ref = None
for mask in (0, 2, 4, 8, 16, 30):
	os.environ["TTK_GEMM_ROLE"] = str(mask)
	m = DiffusionTTS(sd, W.DIFF_FULL, dtype="bf16", device="cuda:0")
	with torch.inference_mode():
		y = m(x, t, precomputed_aligned_embeddings=E)
		y2 = m(x, t, precomputed_aligned_embeddings=E)
	torch.cuda.synchronize()
	if ref is None: ref = y
	print(f"mask {mask:2d}: repeatable {torch.equal(y, y2)}  equal to generic {torch.equal(y, ref)}  max|diff| {(y - ref).abs().max().item():.3e}", flush=True)
	del m
