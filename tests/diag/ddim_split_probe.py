"""Diagnostic: would running the conditioned and the conditioning-free evaluation as two side-by-side chains (b = 1 each, half the
tiles per GEMM) beat the batched 2b chain?  Emulated with two handles, two host threads, two streams, each running an 80-step loop
WITHOUT cond-free guidance (one b = 1 evaluation per step)."""
import os, sys, time, threading
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
dev = "cuda:0"
sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0)
dfs = [DiffusionTTS(sd, W.DIFF_FULL, dtype="bf16", device=dev) for _ in range(2)]
g = torch.Generator().manual_seed(1)
T = 1088
E = torch.randn(1, 1024, T, generator=g).to(dev)
noise = torch.randn(1, 100, T, generator=g).to(dev)

def loop(df, cond_free, stream=None):
	with torch.inference_mode(), torch.cuda.stream(stream or torch.cuda.current_stream()):
		return get_diffuser(80, cond_free).sample_loop(df, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E}, consume_rng=False)

def timed(fn, n=3):
	fn(); torch.cuda.synchronize()
	ts = []
	for _ in range(n):
		t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
	return min(ts)

print(f"batched 2b chain (cond-free on):      {timed(lambda: loop(dfs[0], True)):.1f} ms", flush=True)
print(f"one b=1 chain alone (cond-free off):  {timed(lambda: loop(dfs[0], False)):.1f} ms", flush=True)
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
def both():
	th = [threading.Thread(target=loop, args=(dfs[i], False, streams[i])) for i in range(2)]
	for t in th: t.start()
	for t in th: t.join()
print(f"two b=1 chains side by side:          {timed(both):.1f} ms", flush=True)
