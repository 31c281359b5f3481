"""Diagnostic driver for rocprofv3 --pmc: prefill + a few KV-cached decode tokens at configs[1] size (bf16, B=16)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.autoregressive import UnifiedVoice
dev = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL, dtype="bf16", device=dev, max_batch=16, max_ctx=64 + 4 + 250 + 8, use_graph=False)
g = torch.Generator().manual_seed(1234)
text = torch.randint(1, 255, (1, 64), generator=g).to(dev)
cond = torch.randn(1, 1024, generator=g).to(dev)
with torch.inference_mode():
	ar.inference_speech(cond, text, do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16, max_generate_length=n, suppress_tokens=[8193])
torch.cuda.synchronize()
print("done")
