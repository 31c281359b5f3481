"""Diagnostic: wall time of the 250-token AR loop (configs[1] shape), repeated, for A/B runs on ONE box:
   python tests/diag/ar_ab.py [reps]          (env TTK_LIB / TTK_AR_NARROW / ... select the variant)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.autoregressive import UnifiedVoice
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(os.environ.get("TTK_AB_B", "16"))
NTEXT, NNEW = int(os.environ.get("TTK_AB_TEXT", "64")), int(os.environ.get("TTK_AB_NEW", "250"))      # TTK_AB_B=32 TTK_AB_TEXT=256 TTK_AB_NEW=500: a configs[3] shard
ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL, dtype=os.environ.get("TTK_AB_DTYPE", "bf16"), device=dev, max_batch=B, max_ctx=NTEXT + 4 + NNEW + 8)
g = torch.Generator().manual_seed(1234)
text = torch.randint(1, 255, (1, NTEXT), generator=g).to(dev)
cond = torch.randn(1, 1024, generator=g).to(dev)
run = lambda: ar.inference_speech(cond, text, do_sample=True, temperature=0.8, top_k=0, num_return_sequences=B, max_generate_length=NNEW, suppress_tokens=[8193])
with torch.inference_mode():
	run(); run(); torch.cuda.synchronize()
	ts = []
	for _ in range(reps):
		t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
print(f"{os.environ.get('TTK_LIB', 'libttk.so').split('/')[-1]} " + " ".join(f"{k[4:]}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TTK_") and k != "TTK_LIB") + f": min {min(ts):.2f} ms  median {sorted(ts)[len(ts) // 2]:.2f} ms", flush=True)
