"""Diagnostic: wall time of CLVP scoring at the benchmark's shape (64 text tokens, 16 candidates x 250 codes), full-size model."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.clvp import CLVP
dev = "cuda:0"
sd = W.synth_state_dict(W.clvp_shapes(W.CLVP_FULL), 0)
g = torch.Generator().manual_seed(1)
text = torch.randint(1, 255, (1, 64), generator=g).to(dev)
codes = torch.randint(0, 8192, (16, 250), generator=g).to(dev)
for dt in (sys.argv[1:] or ["bf16"]):
	m = CLVP(sd, W.CLVP_FULL, dtype=dt, device=dev)
	m(text, codes); torch.cuda.synchronize()
	ts = []
	for _ in range(5):
		t0 = time.perf_counter(); m(text, codes); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
	print(f"CLVP {dt}: {min(ts):.2f} ms per utterance (16 candidates)", flush=True)
	del m
