"""Diagnostic: K utterances through inference (sequential) vs inference_lines (AR of line i+1 overlapping DDIM of line i)."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.autoregressive import UnifiedVoice
from tortoise_tts_amd.diffusion import DiffusionTTS
from tortoise_tts_amd.inference import TTSHotPath
dev = "cuda:0"
ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL, dtype="bf16", device=dev, max_batch=16, max_ctx=64 + 4 + 250 + 8)
df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype="bf16", device=dev)
tts = TTSHotPath(ar, df)
g = torch.Generator().manual_seed(1234)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 4
lines = [torch.randint(1, 255, (1, 64), generator=g).to(dev) for _ in range(K)]
al = torch.randn(1, 1024, generator=g).to(dev); dl = torch.randn(1, 2048, generator=g).to(dev)
kw = dict(max_ar_steps=250, max_diffusion_steps=80, ar_temp=0.8, candidates=16, suppress_tokens=[8193])
seq = [tts.inference(l, al, dl, **kw) for l in lines[:1]]; torch.cuda.synchronize()
t0 = time.perf_counter(); seq = [tts.inference(l, al, dl, **kw) for l in lines]; torch.cuda.synchronize(); t_seq = time.perf_counter() - t0
pipe = tts.inference_lines(lines[:1], al, dl, **kw); torch.cuda.synchronize()
t0 = time.perf_counter(); pipe = tts.inference_lines(lines, al, dl, **kw); torch.cuda.synchronize(); t_pipe = time.perf_counter() - t0
same = all(torch.equal(a[0], b[0]) for a, b in zip(seq, pipe))
print(f"K={K} sequential {1e3 * t_seq / K:.1f} ms/utt   pipelined {1e3 * t_pipe / K:.1f} ms/utt   identical mels: {same}", flush=True)
