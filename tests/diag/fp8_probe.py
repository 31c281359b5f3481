"""Diagnostic: distances between the fp8 / fp8w / bf16 diffusion modes and the oracle with and without fp8 operand rounding (small config)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import tortoise_oracle as O
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS
from test_gpu_fp8 import fp8_round, relerr, DIFF_FP8_KEYS
DEV = "cuda:0"
cfg = W.DIFF_SMALL
g = np.load(os.path.join(ROOT, "tests", "golden", "diff_small.npz"))
sd = W.synth_state_dict(W.diffusion_shapes(cfg), int(g["seed"]))
sd_r = {k: (fp8_round(v)[0] if k.endswith(DIFF_FP8_KEYS) else v) for k, v in sd.items()}
x, t, E = torch.from_numpy(g["x"]), torch.from_numpy(g["t"]), torch.from_numpy(g["E"])
run = lambda m: m(x.to(DEV), t.to(DEV), precomputed_aligned_embeddings=E.to(DEV)).cpu()
y8, yw, yb = (run(DiffusionTTS(sd, cfg, dtype=d, device=DEV)) for d in ("fp8", "fp8w", "bf16"))
ybr = run(DiffusionTTS(sd_r, cfg, dtype="bf16", device=DEV))
with torch.inference_mode():
	plain = O.DiffusionOracle(sd, cfg).forward(x, t, E)
	plain_r = O.DiffusionOracle(sd_r, cfg).forward(x, t, E)
	O.BLOCK_OPERAND_ROUNDING = O.fp8_e4m3_round
	emu = O.DiffusionOracle(sd_r, cfg).forward(x, t, E)
	O.BLOCK_OPERAND_ROUNDING = None
for name, a, b in (("bf16 vs oracle", yb, plain), ("fp8w vs oracle(rounded w)", yw, plain_r), ("fp8 vs oracle(rounded w + act)", y8, emu),
				   ("fp8 vs oracle(rounded w)", y8, plain_r), ("oracle(rounded w + act) vs oracle(rounded w)", emu, plain_r), ("fp8 vs fp8w", y8, yw),
				   ("fp8 vs plain oracle", y8, plain), ("fp8w vs plain oracle", yw, plain)):
	print(f"{name:48s} {relerr(a, b):.4f}")
