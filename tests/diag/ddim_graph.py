"""Diagnostic: the 80-step DDIM loop at configs[1] size (bf16, T=1088) launched eagerly (one C call enqueues ~10k launches over two
streams) against the SAME call captured once into a HIP graph and replayed.
   python tests/diag/ddim_graph.py [reps]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import weights as W
from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
dev = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL, dtype="bf16", device=dev)
g = torch.Generator().manual_seed(1)
T = 1088
E = torch.randn(1, 1024, T, generator=g).to(dev)
noise = torch.randn(1, 100, T, generator=g).to(dev)
d = get_diffuser(80, True)
run = lambda: d.sample_loop(df, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E}, consume_rng=False)
def timed(fn):
	ts = []
	for _ in range(reps):
		torch.cuda.synchronize(); t0 = time.perf_counter(); out = fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t0))
	return min(ts), out
with torch.inference_mode():
	ref = run(); run(); torch.cuda.synchronize()
	t_eager, _ = timed(run)
	print(f"eager: {t_eager:.2f} ms", flush=True)
	try:
		gr = torch.cuda.CUDAGraph()
		s = torch.cuda.Stream()
		with torch.cuda.stream(s):
			with torch.cuda.graph(gr, stream=s, capture_error_mode="thread_local"):
				out = run()
		torch.cuda.synchronize()
		t_graph, _ = timed(gr.replay)
		print(f"graph replay: {t_graph:.2f} ms   identical: {bool(torch.equal(out, ref))}", flush=True)
	except Exception as e:
		print("capture failed:", repr(e)[:400], flush=True)
