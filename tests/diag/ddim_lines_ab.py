"""Diagnostic: wall time of a ragged DDIM batch of several lines (default 4 lines of T = 1088, bf16, 20 steps) for A/B runs on ONE box:
   python tests/diag/ddim_lines_ab.py [reps]          (env TTK_AB_LINES, TTK_AB_T, TTK_AB_STEPS, TTK_GEMM_MIXED ... select the variant)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L, T, STEPS = int(os.environ.get("TTK_AB_LINES", "4")), int(os.environ.get("TTK_AB_T", "1088")), int(os.environ.get("TTK_AB_STEPS", "20"))
df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype="bf16", device=dev)
g = torch.Generator().manual_seed(1)
Es = [torch.randn(1, 1024, T, generator=g).to(dev) for _ in range(L)]
noises = [torch.randn(1, 100, T, generator=g).to(dev) for _ in range(L)]
run = lambda: get_diffuser(STEPS, True).sample_loop_lines(df, noises, Es)
with torch.inference_mode():
	run(); torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TTK_")) + f": {L} lines x T {T}, {STEPS} steps: min {min(ts):.2f} ms  median {sorted(ts)[len(ts) // 2]:.2f} ms", flush=True)
