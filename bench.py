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
Also reported: `roofline` of the dominant kernel (HIP events inside libttk, see ttk_prof_*) with `roofline.phases` (AR decode
against the HBM peak, latent pass and DDIM loop against the MFMA peak, from device events around the phases of one more
un-instrumented step and SURVEY.md section 8d's algorithmic work), and `cpu_baseline` (the CPU oracle on a bounded sample of the
same workload, rank 0, N = 1 only).

--shard candidates   BASELINE configs[3] instead (not the headline line): ONE long-form utterance at a time, 2 lines x 256 text
         tokens, 32 candidates per GPU (256 at N = 8) x 500 mel tokens, 200 DDIM steps at T = 2176; candidates sharded over the ranks
         (tortoise_tts_amd/dist.py: ids all-gathered over RCCL, scores all-gathered, the lines' diffusions spread over the ranks, mels broadcast).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TEXT_TOKENS, CANDIDATES, MEL_TOKENS, DDIM_STEPS = 64, 16, 250, 80


def parse():
	ap = argparse.ArgumentParser()
	ap.add_argument("--gpus", type=int, default=1)
	ap.add_argument("--steps", type=int, default=3)
	ap.add_argument("--warmup", type=int, default=1)
	ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16", "f32", "fp8w", "fp8"],
					help="BASELINE config 5 -- fp8w: bf16 arithmetic, block GEMM weights in fp8-e4m3; fp8: also fp8 activations into the diffusion block GEMMs (fp8 MFMA)")
	ap.add_argument("--with-vocoder", action="store_true", help="BASELINE config 5's tail: the BigVGAN vocoder (bf16) inside the step; off for the headline metric")
	ap.add_argument("--no-cpu-baseline", action="store_true")
	ap.add_argument("--no-roofline", action="store_true")
	ap.add_argument("--small", action="store_true", help="tiny models (plumbing check only; the number is NOT the metric)")
	ap.add_argument("--no-graph", action="store_true", help="eager token loop (counter passes under rocprofv3 --pmc, which crashes on captured graphs); never the timed configuration")
	ap.add_argument("--shard", default="utterances", choices=["utterances", "candidates"],
					help="utterances: configs[1]/[2], one utterance per GPU (the headline metric); candidates: configs[3], one utterance's candidates over the GPUs")
	return ap.parse_args()


def cpu_baseline(seed, n_text=TEXT_TOKENS, n_cand=CANDIDATES, n_mel=MEL_TOKENS, n_ddim=DDIM_STEPS, n_lines=1, light=False):
	"""The CPU oracle (oracle/tortoise_oracle.py, kind 'port') on a bounded sample of the same workload, on this box's host cores, as
	BASELINE.md section 3 lays out: phases timed separately with time.perf_counter after a warm-up, medians of repeated samples --
	prefill at B = n_cand (second run timed), KV-cached decode steps (median of 3 groups of 8 steps, scaled to n_mel), the latent
	pass on the candidates at its full length (one run), DDIM steps at the FULL T (1 warm-up + median of 8 steps, each a
	conditioned + a conditioning-free evaluation, scaled to n_ddim).  light (configs[3], where one DDIM step at T = 2176 is tens of seconds
	of CPU time): groups of 4 decode steps, the latent pass on 4 candidates scaled to n_cand, 1 warm-up + 2 DDIM steps."""
	sys.path.insert(0, os.path.join(ROOT, "oracle"))
	import statistics
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
	T = n_mel * 4 * 24000 // 22050
	grp, n_lat, n_steps = (4, min(4, n_cand), 2) if light else (8, n_cand, 8)
	with torch.inference_mode():
		ar = O.AROracle(W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0), W.AR_FULL)
		text = torch.randint(1, 255, (1, n_text), generator=g)
		cond = torch.randn(1, 1024, generator=g)
		prefix = ar.prefix_embeddings(cond, text)
		ar.prefill(prefix, n_cand)                                       # warm-up
		t0 = time.perf_counter()
		logits, past, _ = ar.prefill(prefix, n_cand)
		t_prefill = time.perf_counter() - t0
		tok = torch.randint(0, 8192, (n_cand,), generator=g)
		k = 0
		for _ in range(2):                                               # warm-up steps
			k += 1
			_, past, _ = ar.decode(tok, k, past)
		groups = []
		for _ in range(3):
			t0 = time.perf_counter()
			for _ in range(grp):
				k += 1
				_, past, _ = ar.decode(tok, k, past)
			groups.append((time.perf_counter() - t0) / grp)
		t_dec = statistics.median(groups)
		del past
		codes = torch.randint(0, 8192, (n_lat, n_mel), generator=g)
		t0 = time.perf_counter()
		ar.forward_latents(cond.repeat(n_lat, 1), text.repeat(n_lat, 1), codes)
		t_lat = (time.perf_counter() - t0) * n_cand / n_lat
		del ar
		d = O.DiffusionOracle(W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0), W.DIFF_FULL)
		x = torch.randn(1, 100, T, generator=g)
		E = torch.randn(1, 1024, T, generator=g)
		sched = O.SpacedSchedule(steps=n_ddim)
		x = sched.ddim_step(d, x, n_ddim - 1, E)                          # warm-up
		steps = []
		for i in range(n_steps):
			t0 = time.perf_counter()
			x = sched.ddim_step(d, x, n_ddim - 2 - i, E)
			steps.append(time.perf_counter() - t0)
		t_step = statistics.median(steps)
	est = n_lines * (t_prefill + n_mel * t_dec + t_lat + n_ddim * t_step)
	audio = n_lines * T * 256 / 24000
	return {"value": audio / est, "unit": "audio-sec/wall-sec", "cores": cores, "kind": "port",
			"torch": torch.__version__, "seconds_per_utterance": est,
			"sample": f"B={n_cand}: prefill (2nd run) {t_prefill:.2f} s + decode {t_dec * 1e3:.0f} ms/step (median of 3 groups of {grp}, scaled to {n_mel}) + "
					  f"latent pass on {n_lat} candidates x {n_mel} tokens" + (f" scaled to {n_cand}" if n_lat != n_cand else "") + f" {t_lat:.2f} s (one run) + DDIM {t_step:.2f} s/step at T={T} "
					  f"(cond + cond-free evaluation; 1 warm-up, median of {n_steps} steps, scaled to {n_ddim})" + (f"; x {n_lines} lines" if n_lines > 1 else "") + f"; fp32, {cores} threads"}


def phase_roofline(marks_per_line, dtype_name, n_text=TEXT_TOKENS, n_cand=CANDIDATES, n_mel=MEL_TOKENS, n_ddim=DDIM_STEPS):
	"""Per-phase roofline fractions (SURVEY.md section 8d / BASELINE.md section 4): AR decode is HBM-bound (weights streamed once per
	token + the KV cache read), the latent pass and the DDIM loop are MFMA-bound.  `marks_per_line`: one list of (name, event) pairs per text
	line, from TTSHotPath.inference(phase_marks=...) / inference_sharded(phase_marks=...) of one step run exactly as the timed ones (captured
	graph, no instrumentation); n_cand = candidates decoded on THIS GPU.  Peaks: 8 TB/s HBM; dense MFMA 2.5 PFLOP/s bf16 / f16, 157.3 TFLOP/s
	f32; dtype fp8: the DDIM loop is graded against the 5 PFLOP/s fp8 peak (its ResBlock / AttentionBlock projection GEMMs run the fp8 MFMA; QK^T, PV
	and the remaining convs run bf16 -- the stricter denominator is used for the whole phase), the latent pass against 2.5 (the AR handle's fp8 is
	fp8 WEIGHTS on the bf16 MFMA)."""
	ms = {}
	for marks in marks_per_line:
		for i in range(len(marks) - 1):
			ms[marks[i + 1][0]] = ms.get(marks[i + 1][0], 0.0) + marks[i][1].elapsed_time(marks[i + 1][1])
	L = len(marks_per_line)
	# candidate shards with several lines: the lines' diffusions are spread over the ranks (dist.assign_diffusers), so THIS rank's "ddim" interval covers the
	# lines it diffused (one ragged batch) -- possibly none; "_"-prefixed marks bracket time that belongs to no phase of the line that carries them
	L_ddim = sum((m[2] if len(m) > 2 else 1) for marks in marks_per_line for m in marks if m[0] == "ddim")
	e_w = {"bf16": 2, "f16": 2, "f32": 4, "fp8w": 1, "fp8": 1}[dtype_name]
	e_kv = 4 if dtype_name == "f32" else 2
	peak_f = 157.3e12 if dtype_name == "f32" else 2.5e15
	peak_ddim = 5.0e15 if dtype_name == "fp8" else peak_f
	P1 = n_text + 4                                    # prefix rows incl. start_mel
	# prefill: dense pass over P1 rows (once: the candidates share the prefix); decode step k (k = 1..M-1) reads a cache of P1 + k - 1 rows and writes one
	blocks, head = 377_886_720, 8_398_850 + 4_096
	ar_bytes = 0.0
	for k in range(1, n_mel):
		ctx = P1 + k
		ar_bytes += blocks * e_w + head * (2 if e_w == 1 else e_w) + n_cand * 30 * 2 * ctx * 1024 * e_kv + n_cand * 8194 * 4
	ar_bytes *= L
	prefill_flop = L * (2.0 * blocks * P1 * n_cand + 2.0 * P1 * P1 * 1024 * 30 * n_cand)
	S = n_text + n_mel + 5
	lat_flop = L * (2.0 * blocks * S * n_cand + 2.0 * S * S * 1024 * 30 * n_cand)
	T = n_mel * 4 * 24000 // 22050
	F = 236 * 1024 ** 2 * T + 52 * 1024 * T * T + 1_843_200 * T
	ddim_flop = L_ddim * n_ddim * 2.0 * F
	ar_floor = ar_bytes / 8.0e12 * 1e3 + prefill_flop / peak_f * 1e3
	def phase(bound, work_key, work, name, peak, unit_key, unit_scale, **extra):
		t = ms.get(name)
		if not t or not work:           # the phase did not run on this rank (a candidate shard whose lines are all diffused elsewhere)
			return {"bound": bound, work_key: work, "ms": t, "frac": None, "floor_ms": work / peak * 1e3, "note": "did not run on this rank", **extra}
		return {"bound": bound, work_key: work, "ms": t, unit_key: work / (t * 1e-3) / unit_scale, "frac": work / (t * 1e-3) / peak, "floor_ms": work / peak * 1e3, **extra}
	out = {
		"ar_decode": phase("hbm", "algorithmic_bytes", ar_bytes, "ar_decode", 8.0e12, "achieved_GBps", 1e9, prefill_flop=prefill_flop,
						   note=f"prefill + {n_mel} sampled tokens; bytes = {n_mel - 1} KV-cached steps (weights once per step + KV read + logits)"),
		"latent_pass": phase("mfma", "flop", lat_flop, "latent_pass", peak_f, "achieved_TFLOPs", 1e12),
		"ddim": phase("mfma", "flop", ddim_flop, "ddim", peak_ddim, "achieved_TFLOPs", 1e12, peak_TFLOPs=peak_ddim / 1e12, lines_diffused_here=L_ddim,
					  note=f"timestep-independent conditioning + {n_ddim} steps x (cond + cond-free evaluation); flop = {2 * n_ddim} F(T) per line diffused on this rank"),
	}
	out["ar_decode"]["floor_ms"] = ar_floor
	if L > 1:
		out["lines"] = L
	out["whole_step_floor_ms"] = out["ar_decode"]["floor_ms"] + out["latent_pass"]["floor_ms"] + out["ddim"]["floor_ms"]
	out["whole_step_ms"] = sum(v for k, v in ms.items() if not k.startswith("_"))
	out["whole_step_frac_of_floor"] = out["whole_step_floor_ms"] / out["whole_step_ms"] if out["whole_step_ms"] else None
	return out


def log(msg):
	print(f"[bench] {msg}", file=sys.stderr, flush=True)


def launcher_command(gpus, argv, port):
	"""the command `python bench.py --gpus N` turns itself into when it was started WITHOUT a torch.distributed environment: one child rank per
	GPU under torch.distributed.run (the same line the driver would use for N > 1), rendezvous on 127.0.0.1"""
	return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}", "--master-addr", "127.0.0.1",
			"--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(a):
	"""`python bench.py --gpus N` (N > 1) as the driver starts the bench for N = 1: this process never touches the GPU (no torch import, no
	HIP call) -- it starts N fresh ranks as CHILD processes (never an exec of a process that initialised the GPU), relays rank 0's JSON line
	and exits with the children's status."""
	with socket.socket() as sk:
		sk.bind(("127.0.0.1", 0))
		port = sk.getsockname()[1]
	env = dict(os.environ)
	env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
	env.setdefault("OMP_NUM_THREADS", "4")
	cmd = launcher_command(a.gpus, sys.argv[1:], port)
	log(f"no torch.distributed environment and --gpus {a.gpus}: starting {a.gpus} ranks: {' '.join(cmd)}")
	proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
	lines = []
	for line in proc.stdout:
		line = line.rstrip("\n")
		if line.startswith("{") and '"metric"' in line:
			lines.append(line)
		else:
			print(line, file=sys.stderr, flush=True)      # anything else the ranks print is not the result line
	rc = proc.wait()
	if rc == 0 and len(lines) != 1:
		log(f"expected ONE result line from rank 0, got {len(lines)}")
		rc = 1
	for line in lines[-1:]:
		print(line, flush=True)
	return rc


def launch_probe(a):
	"""TTK_BENCH_PROBE=1 (tests/test_host_logic.py, no GPU): the ranks only rendezvous (gloo), count themselves and rank 0 prints a line --
	the launcher branch, the environment hand-over and the relay of ONE line, exercised on a box without a GPU.  Not a measurement."""
	import torch.distributed as dist
	world = int(os.environ.get("WORLD_SIZE", "1"))
	if world > 1:
		dist.init_process_group("gloo")
	one = torch.ones(1)
	if world > 1:
		dist.all_reduce(one)
	if int(os.environ.get("RANK", "0")) == 0:
		print(json.dumps({"metric": "launcher probe (no measurement)", "value": None, "n_gpus": a.gpus, "n_ranks_seen": int(one.item()),
						  "world_size_env": world, "steps": a.steps, "warmup": a.warmup}), flush=True)
	if world > 1:
		dist.destroy_process_group()


def main():
	a = parse()
	if "WORLD_SIZE" not in os.environ and a.gpus > 1:
		sys.exit(self_launch(a))
	global torch
	import torch
	if os.environ.get("TTK_BENCH_PROBE") == "1":
		return launch_probe(a)
	rank = int(os.environ.get("RANK", "0"))
	world = int(os.environ.get("WORLD_SIZE", "1"))
	local = int(os.environ.get("LOCAL_RANK", "0"))
	if world != a.gpus:
		raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start one rank per GPU (python bench.py --gpus N does it by itself)")
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
	by_cand = a.shard == "candidates"
	# configs[3]: per-GPU shard of 32 candidates, 2 lines x 256 text tokens, 500 mel tokens (T = 2176), 200 DDIM steps
	n_text, n_cand, n_mel, n_ddim, n_lines = (256, 32, 500, 200, 2) if by_cand else (TEXT_TOKENS, CANDIDATES, MEL_TOKENS, DDIM_STEPS, 1)
	if by_cand and world == 1:
		os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
		os.environ.setdefault("MASTER_PORT", "29533")
		os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
		dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(dev))      # the sharded entry runs on a group of any size
	ar = UnifiedVoice(W.synth_state_dict(W.ar_shapes(ar_cfg), 0), ar_cfg, dtype=a.dtype, device=dev, max_batch=n_cand,
					  max_ctx=n_text + 4 + n_mel + 8)
	if a.no_graph:
		ar.use_graph = False
	df = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(df_cfg), 0), df_cfg, dtype=a.dtype, device=dev)
	voc = None
	if a.with_vocoder:
		from tortoise_tts_amd.vocoder import BigVGAN
		vcfg = W.VOC_SMALL if a.small else W.VOC_FULL
		voc = BigVGAN(W.synth_state_dict(W.vocoder_shapes(vcfg), 0), vcfg, dtype="f32" if a.dtype == "f32" else "bf16", device=dev)
	tts = TTSHotPath(ar, df, vocoder=voc)
	g = torch.Generator().manual_seed(1234 + (0 if by_cand else rank))     # candidate shards: every rank holds the SAME utterance
	lines = [torch.randint(1, 255, (1, n_text), generator=g).to(dev) for _ in range(n_lines)]
	text = lines[0]
	ar_lat = torch.randn(1, ar_cfg.model_dim, generator=g).to(dev)
	df_lat = torch.randn(1, 2 * df_cfg.model_channels, generator=g).to(dev)
	kw = dict(max_ar_steps=n_mel, max_diffusion_steps=n_ddim, ar_temp=0.8, candidates=n_cand,
			  suppress_tokens=[ar_cfg.stop_mel_token], return_all=True)
	gdev = "cpu" if rehearsal else dev
	gathered = [torch.empty((CANDIDATES, MEL_TOKENS), dtype=torch.long, device=gdev) for _ in range(world)] if world > 1 and not by_cand else None

	def step(exchange=True, marks=None):
		if by_cand:     # one long-form utterance: its lines sampled one after the other, each with its candidates sharded over the ranks; the lines' winners
			# are diffused as one ragged batch on their owner (TTSHotPath.inference_sharded_lines: each line's mel equals its own call's)
			skw = {k: v for k, v in kw.items() if k != "candidates"}
			res = tts.inference_sharded_lines(lines, ar_lat, df_lat, candidates=n_cand * world, phase_marks=marks, **skw)
			return sum(r[1] for r in res)
		mels, seconds, aux = tts.inference(text, ar_lat, df_lat, phase_marks=marks, **kw)
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
	if not a.no_roofline and (rank == 0 or by_cand):
		# configs[1]/[2]: rank 0 alone (no collective in these passes).  configs[3]: the sharded step IS collectives, so every rank runs the two
		# extra steps and rank 0 reports its own GPU's phases (it owns the diffused candidate: no scorer is attached, candidate 0 wins).
		from tortoise_tts_amd import profiling
		marks = []
		step(exchange=False, marks=marks)                  # one more step, run exactly as the timed ones, with events at the phase boundaries
		torch.cuda.synchronize()
		roof = profiling.dominant_kernel_roofline(lambda: step(exchange=False), ar, df)
		if not a.small and rank == 0:      # only rank 0 reports; a shard that diffused no line has no "ddim" interval (phase_roofline says so instead of raising)
			roof["phases"] = phase_roofline(marks, a.dtype, n_text, n_cand, n_mel, n_ddim) if by_cand else phase_roofline([marks], a.dtype)
	# informational, never `value`: the k = 1 variant SURVEY.md 8d row 2 asks to be reported beside the headline -- the candidate is chosen first and
	# only its row goes through the dense latent pass (same bits out; the reference runs all 16, inference.py:370-379, and so does `value`)
	k1 = None
	if rank == 0 and world == 1 and not a.no_roofline and not by_cand:
		kw1 = dict(kw, latents_for="winner")
		tts.inference(text, ar_lat, df_lat, **kw1)
		torch.cuda.synchronize()
		t1 = time.perf_counter()
		sec = sum(tts.inference(text, ar_lat, df_lat, **kw1)[1] for _ in range(2))
		torch.cuda.synchronize()
		d1 = time.perf_counter() - t1
		k1 = {"value": sec / d1, "unit": "audio-sec/wall-sec", "ms_per_step": 1e3 * d1 / 2, "note": "latent pass on the diffused candidate only; not the headline metric"}
	# informational, never `value`: a stream of utterances through TTSHotPath.inference_lines -- the sampling of 4 consecutive lines as ONE
	# decode batch (weights streamed once per token for all of them), line i's diffusion overlapped with later lines' sampling; results
	# identical to the per-line calls.  Built on its own handle (max_batch = 4 x candidates) AFTER the headline was measured; `value` above
	# stays the one-utterance-at-a-time figure of configs[1].
	piped = None
	if rank == 0 and world == 1 and not a.no_roofline and not by_cand:
		n_lines, per_batch = 8, 4
		lens = [64, 48, 88, 56, 72, 40, 80, 64][:n_lines]          # text lengths vary from line to line (mean 64): one captured token step serves them all
		gl = torch.Generator().manual_seed(4321)
		ltexts = [torch.randint(1, 255, (1, n), generator=gl).to(dev) for n in lens]
		ar4 = UnifiedVoice(W.synth_state_dict(W.ar_shapes(ar_cfg), 0), ar_cfg, dtype=a.dtype, device=dev, max_batch=per_batch * n_cand,
						   max_ctx=max(lens) + 4 + n_mel + 8)
		tts4 = TTSHotPath(ar4, df)
		lkw = {k: v for k, v in kw.items() if k != "return_all"}
		tts4.inference_lines(ltexts[:per_batch], ar_lat, df_lat, ar_batch_lines=per_batch, **lkw)
		torch.cuda.synchronize()
		t1 = time.perf_counter()
		res = tts4.inference_lines(ltexts, ar_lat, df_lat, ar_batch_lines=per_batch, **lkw)
		torch.cuda.synchronize()
		dl = time.perf_counter() - t1
		piped = {"value": sum(r[1] for r in res) / dl, "unit": "audio-sec/wall-sec", "lines": n_lines, "text_tokens": lens, "ar_batch_lines": per_batch, "ms_per_line": 1e3 * dl / n_lines,
				 "note": "stream of utterances: 4 lines sampled as one decode batch, diffusion pipelined under the next batch; not the headline metric"}
		del tts4, ar4
	# informational, never `value`: the loop real utterances run.  The headline suppresses the stop token (fixed length), so
	# HF's stopping test is compiled out of its loop (`can_stop` False).  Here the stop token is NOT suppressed but unreachable -- mel_head.bias[stop] = -1e9 in
	# a second handle's synthetic weights, so its probability is exactly 0, the length stays 250 and the ids must equal the headline's -- and the loop runs
	# with the stopping test live: one event per token, the pinned flag read LAG tokens late (autoregressive.py `_token_loop`).
	live = None
	if rank == 0 and world == 1 and not a.no_roofline and not by_cand:
		sd = W.synth_state_dict(W.ar_shapes(ar_cfg), 0)
		sd["mel_head.bias"] = sd["mel_head.bias"].clone()
		sd["mel_head.bias"][ar_cfg.stop_mel_token] = -1e9
		ar_l = UnifiedVoice(sd, ar_cfg, dtype=a.dtype, device=dev, max_batch=n_cand, max_ctx=n_text + 4 + n_mel + 8)
		del sd
		tts_l = TTSHotPath(ar_l, df, vocoder=voc)
		kwl = {k: v for k, v in kw.items() if k != "suppress_tokens"}
		ref_codes = tts.inference(text, ar_lat, df_lat, **kw)[2]["codes"]
		got_codes = tts_l.inference(text, ar_lat, df_lat, **kwl)[2]["codes"]      # (also the warm-up: captures this handle's token step)
		torch.cuda.synchronize()
		n_live = max(2, min(a.steps, 5))
		t1 = time.perf_counter()
		sec = sum(tts_l.inference(text, ar_lat, df_lat, **kwl)[1] for _ in range(n_live))
		torch.cuda.synchronize()
		dl = time.perf_counter() - t1
		live = {"value": sec / dl, "unit": "audio-sec/wall-sec", "ms_per_step": 1e3 * dl / n_live, "steps": n_live,
				"ids_equal_headline": bool(got_codes.shape == ref_codes.shape and torch.equal(got_codes, ref_codes)), "mel_tokens": int(got_codes.shape[1]),
				"note": "stop token live (not suppressed, unreachable through mel_head.bias[stop] = -1e9): HF's per-token stopping test runs as the lagged pinned flag; not the headline metric"}
		del tts_l, ar_l
	log("roofline pass done; cpu baseline")
	cpu = None
	if rank == 0 and world == 1 and not a.no_cpu_baseline and not a.small:
		cpu = cpu_baseline(1234, n_text, n_cand, n_mel, n_ddim, n_lines, light=True) if by_cand else cpu_baseline(1234)

	if rank == 0:
		cfg_line = {"workload": "configs[1]: 1 utterance/GPU, 64 text tokens, 16 AR candidates x 250 mel tokens (KV-cached decode), "
								"latent pass on 16 candidates, 80 DDIM steps with cond-free guidance at T=1088 (11.6 s audio)",
					"text_tokens": TEXT_TOKENS, "candidates": CANDIDATES, "mel_tokens": MEL_TOKENS, "ddim_steps": DDIM_STEPS,
					"mel_frames": MEL_TOKENS * 4 * 24000 // 22050, "parallelism": f"utterances x{world}" if world > 1 else "single GPU",
					"small_models": bool(a.small), "vocoder_in_step": bool(a.with_vocoder)}
		if a.dtype in ("fp8", "fp8w"):      # BASELINE config 5: say exactly which contractions are fp8
			cfg_line["workload"] += ("; config 5 arithmetic: " + (
				"diffusion ResBlock / AttentionBlock projection GEMMs on v_mfma_f32_16x16x32_fp8_fp8 (e4m3 weights x power-of-two tensor scale, e4m3 activations); "
				"QK^T / PV, the k=3 input/output convs and the time-embedding linears on the bf16 MFMA; autoregressive side: fp8 WEIGHT bytes widened next to the bf16 MFMA "
				"(16-row GEMVs), bf16 KV cache" if a.dtype == "fp8" else
				"fp8-e4m3 block-GEMM WEIGHTS in both networks (power-of-two tensor scale), every contraction on the bf16 MFMA"))
			cfg_line["fp8_contractions"] = "diffusion block projection GEMMs" if a.dtype == "fp8" else "none (weights only)"
			cfg_line["ar_handle_dtype"] = "fp8w"
		if by_cand:
			cfg_line = {"workload": f"configs[3]: one long-form utterance = 2 lines x 256 text tokens, {n_cand * world} AR candidates ({n_cand} per GPU) x 500 mel "
									"tokens, latent pass + candidate choice on every shard, 200 DDIM steps with cond-free guidance at T=2176 (23.2 s audio per line), "
									"the lines' diffusions spread over the GPUs (latents + start noise broadcast from the winner's GPU); ids / scores all-gathered and the mels broadcast over RCCL",
						"text_tokens": n_text, "candidates": n_cand * world, "mel_tokens": n_mel, "ddim_steps": n_ddim, "lines": n_lines,
						"mel_frames": n_mel * 4 * 24000 // 22050, "parallelism": f"candidates x{world}", "small_models": bool(a.small)}
		line = {
			"metric": "audio-sec/wall-sec (RTF^-1), 16 AR candidates x 80 DDIM steps" if not by_cand else "audio-sec/wall-sec (RTF^-1), 32 AR candidates per GPU x 200 DDIM steps (configs[3])",
			"value": (1 if by_cand else world) * audio / dt,
			"unit": "audio-sec/wall-sec", "n_gpus": world, "n_ranks_seen": dist.get_world_size() if dist.is_initialized() else 1,
			"backend": dist.get_backend() if dist.is_initialized() else None, "steps": a.steps, "warmup": a.warmup,
			"ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
			"dtype": a.dtype, "data": "synthetic",
			"config": cfg_line,
			"roofline": roof, "cpu_baseline": cpu, "latent_k1_variant": k1, "pipelined_lines": piped, "stop_live": live,
		}
		print(json.dumps(line), flush=True)
	if dist.is_initialized():
		dist.destroy_process_group()


if __name__ == "__main__":
	main()
