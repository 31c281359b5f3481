#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

metric   audio-sec/wall-sec (RTF^-1): seconds of 24 kHz audio whose mel spectrogram the hot path produces per wall second
workload configs[1]: one utterance per GPU, 64 text tokens, 16 AR candidates x 250 mel tokens (stop token suppressed so the
         length is fixed), latent pass on the 16 candidates, 80 DDIM steps with conditioning-free guidance at T = 1088
         frames (11.6 s of audio), bf16 weights/MFMA operands with f32 accumulation.  Full-size synthetic weights (no
         checkpoints exist offline), inputs resident in HBM before the timed region.
step     one utterance through the whole hot path (tortoise_tts_amd/inference.py: inference.py:331-413 of the reference).
N > 1    weak scaling: every rank runs its own utterance (BASELINE configs[2]); the only exchange is one RCCL all-gather of
         the sampled candidate ids per utterance (the hand-off to candidate scoring); value = N * audio / max-rank time.
Also reported: `roofline` of the dominant kernel (HIP events inside libttk, see ttk_prof_*) and `cpu_baseline` (the CPU
oracle on a bounded sample of the same workload, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TEXT_TOKENS, CANDIDATES, MEL_TOKENS, DDIM_STEPS = 64, 16, 250, 80


def parse():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "fp8w", "fp8"],
					help="BASELINE config 5 -- fp8w: bf16 arithmetic, block GEMM weights in fp8-e4m3; fp8: also fp8 activations into the diffusion block GEMMs (fp8 MFMA)")
	ap.add_argument("--with-vocoder", action="store_true", help="BASELINE config 5's tail: the BigVGAN vocoder (bf16) inside the step; off for the headline metric")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-roofline", action="store_true")
	ap.add_argument("--small", action="store_true", help="tiny models (plumbing check only; the number is NOT the metric)")
	return ap.parse_args()


def cpu_baseline(seed):
	"""The CPU oracle (oracle/tortoise_oracle.py, kind 'port') on a bounded sample of the same workload, on this box's host
	cores: 16 KV-cached decode steps at B=16 after a prefill (scaled to 250), and 4 DDIM steps (cond + cond-free evaluation)
	at T = 544 frames, scaled by the network's flop model F(T) to T = 1088 and to 80 steps."""
	sys.path.insert(0, os.path.join(ROOT, "oracle"))
	import tortoise_oracle as O
	from tortoise_tts_amd import weights as W
	# the GPU box gives a 1-GPU job a 16-core share whatever os.cpu_count() says; oversubscribing it is pathologically slow
	try:
		cores = len(os.sched_getaffinity(0))
	except AttributeError:
		cores = os.cpu_count() or 1
	cores = max(1, min(cores, 16))
	torch.set_num_threads(cores)
	g = torch.Generator().manual_seed(seed)
	with torch.inference_mode():
		ar = O.AROracle(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL)
		text = torch.randint(1, 255, (1, TEXT_TOKENS), generator=g)
		cond = torch.randn(1, 1024, generator=g)
		t0 = time.perf_counter()
		logits, past, _ = ar.prefill(ar.prefix_embeddings(cond, text), CANDIDATES)
		t_prefill = time.perf_counter() - t0
		tok = torch.randint(0, 8192, (CANDIDATES,), generator=g)
		n_dec = 16
		t0 = time.perf_counter()
		for k in range(1, n_dec + 1):
			_, past, _ = ar.decode(tok, k, past)
		t_dec = (time.perf_counter() - t0) / n_dec
		del ar, past
		d = O.DiffusionOracle(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL)
		Ts, n_st = 544, 4
		x = torch.randn(1, 100, Ts, generator=g)
		E = torch.randn(1, 1024, Ts, generator=g)
		sched = O.SpacedSchedule(steps=DDIM_STEPS)
		t0 = time.perf_counter()
		for i in range(n_st):
			x = sched.ddim_step(d, x, DDIM_STEPS - 1 - i, E)
		t_step = (time.perf_counter() - t0) / n_st

	def F(T):   # flop per evaluation, SURVEY.md section 8d
		return 236 * 1024 ** 2 * T + 52 * 1024 * T * T + 1_843_200 * T
	T = MEL_TOKENS * 4 * 24000 // 22050
	est = t_prefill + MEL_TOKENS * t_dec + DDIM_STEPS * t_step * F(T) / F(Ts)
	audio = T * 256 / 24000
	return {"value": audio / est, "unit": "audio-sec/wall-sec", "cores": cores, "kind": "port",
			"sample": f"prefill + {n_dec} decode steps at B=16 (scaled to 250) + {n_st} DDIM steps at T={Ts} scaled by F(T) to T={T} and to 80 steps; "
					  f"measured {t_prefill:.2f}s + {t_dec * 1e3:.0f} ms/decode-step + {t_step:.2f} s/DDIM-step; latent pass not included"}


def log(msg):
	print(f"[bench] {msg}", file=sys.stderr, flush=True)


def main():
	a = parse()
	rank = int(os.environ.get("RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	local = int(os.environ.get("LOCAL_RANK", "0"))
	if world != a.gpus:
		if world == 1 and a.gpus > 1:
			raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
	# TTK_BENCH_REHEARSAL=1: every rank on cuda:0 with the gloo backend -- lets the N>1 control flow run on a 1-GPU box
	# (RCCL refuses two ranks on one device).  Never used by the driver; the number it prints is not a scaling result.
	rehearsal = os.environ.get("TTK_BENCH_REHEARSAL") == "1"
	if rehearsal:
		local = 0
	torch.cuda.set_device(local)
	dev = f"cuda:{local}"
	import torch.distributed as dist
	if world > 1:
		os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
		if rehearsal:
			dist.init_process_group("gloo")
		else:
			dist.init_process_group("nccl", device_id=torch.device(dev))

	from tortoise_tts_amd import _lib, weights as W
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	from tortoise_tts_amd.diffusion import DiffusionTTS
	from tortoise_tts_amd.inference import TTSHotPath
	_lib.load()
	ar_cfg, df_cfg = (W.AR_SMALL, W.DIFF_SMALL) if a.small else (W.AR_FULL, W.DIFF_FULL)
	ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(ar_cfg), 0), ar_cfg, dtype=a.dtype, device=dev, max_batch=CANDIDATES,
					  max_ctx=TEXT_TOKENS + 4 + MEL_TOKENS + 8)
	df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(df_cfg), 0), df_cfg, dtype=a.dtype, device=dev)
	voc = None
	if a.with_vocoder:
		from tortoise_tts_amd.vocoder import BigVGAN
		vcfg = W.VOC_SMALL if a.small else W.VOC_FULL
		voc = BigVGAN(W.synth_state_dict(W.vocoder_shapes(vcfg), 0), vcfg, dtype="f32" if a.dtype == "f32" else "bf16", device=dev)
	tts = TTSHotPath(ar, df, vocoder=voc)
	g = torch.Generator().manual_seed(1234 + rank)
	text = torch.randint(1, 255, (1, TEXT_TOKENS), generator=g).to(dev)
	ar_lat = torch.randn(1, ar_cfg.model_dim, generator=g).to(dev)
	df_lat = torch.randn(1, 2 * df_cfg.model_channels, generator=g).to(dev)
	kw = dict(max_ar_steps=MEL_TOKENS, max_diffusion_steps=DDIM_STEPS, ar_temp=0.8, candidates=CANDIDATES,
			  suppress_tokens=[ar_cfg.stop_mel_token], return_all=True)
	gdev = "cpu" if rehearsal else dev
	gathered = [torch.empty((CANDIDATES, MEL_TOKENS), dtype=torch.long, device=gdev) for _ in range(world)] if world > 1 else None

	def step(exchange=True):
		mels, seconds, aux = tts.inference(text, ar_lat, df_lat, **kw)
		if voc is not None:
			voc.inference(mels)
		if world > 1 and exchange:   # hand the candidate ids to the scoring rank (RCCL all-gather over xGMI, 32 KB per rank)
			dist.all_gather(gathered, aux["codes"].contiguous().to(gdev))
		return seconds

	def fence():
		if world > 1:
			dist.barrier()
		torch.cuda.synchronize()

	log("models built; warmup")
	for _ in range(a.warmup):
		step()
	fence()
	log("timed region")
	t0 = time.perf_counter()
	audio = 0.0
	for _ in range(a.steps):
		audio += step()
	fence()
	dt = time.perf_counter() - t0
	tmax = torch.tensor([dt], dtype=torch.float64, device=gdev)
	if world > 1:
		dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
	dt = float(tmax.item())

	log(f"timed {a.steps} steps in {dt:.3f}s")
	roof = None
	if rank == 0 and not a.no_roofline:
		from tortoise_tts_amd import profiling
		roof = profiling.dominant_kernel_roofline(lambda: step(exchange=False), ar, df)   # rank 0 alone: no collective in here
	# informational, never `value`: a stream of utterances with line i's diffusion overlapped with line i+1's sampling
	# (TTSHotPath.inference_lines; identical results).  `value` above stays the one-utterance-at-a-time figure of configs[1].
	piped = None
	if rank == 0 and world == 1 and not a.no_roofline:
		n_lines = 4
		lkw = {k: v for k, v in kw.items() if k != "return_all"}
		tts.inference_lines([text] * 2, ar_lat, df_lat, **lkw)
		torch.cuda.synchronize()
		t1 = time.perf_counter()
		res = tts.inference_lines([text] * n_lines, ar_lat, df_lat, **lkw)
		torch.cuda.synchronize()
		dl = time.perf_counter() - t1
		piped = {"value": sum(r[1] for r in res) / dl, "unit": "audio-sec/wall-sec", "lines": n_lines, "ms_per_line": 1e3 * dl / n_lines,
				 "note": "software-pipelined stream of utterances; not the headline metric"}
	log("roofline pass done; cpu baseline")
	cpu = None
	if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.small:
		cpu = cpu_baseline(1234)

	if rank == 0:
		line = {
			"metric": "audio-sec/wall-sec (RTF^-1), 16 AR candidates x 80 DDIM steps", "value": world * audio / dt,
			"unit": "audio-sec/wall-sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
			"ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": a.dtype, "data": "synthetic",
			"config": {"workload": "configs[1]: 1 utterance/GPU, 64 text tokens, 16 AR candidates x 250 mel tokens (KV-cached decode), "
								   "latent pass on 16 candidates, 80 DDIM steps with cond-free guidance at T=1088 (11.6 s audio)",
					   "text_tokens": TEXT_TOKENS, "candidates": CANDIDATES, "mel_tokens": MEL_TOKENS, "ddim_steps": DDIM_STEPS,
					   "mel_frames": MEL_TOKENS * 4 * 24000 // 22050, "parallelism": f"utterances x{world}" if world > 1 else "single GPU",
					   "small_models": bool(a.small), "vocoder_in_step": bool(a.with_vocoder)},
			"roofline": roof, "cpu_baseline": cpu, "pipelined_lines": piped,
		}
		print(json.dumps(line), flush=True)
	if world > 1:
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
