"""CPU tests of the host-side logic of the product (no GPU, no compute calls into libttk)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tortoise_oracle as O
from tortoise_tts_amd import _lib, diffusion as D, sampling, weights as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
	_lib.build()
	lib = ctypes.CDLL(_lib.LIB_PATH)
	header = open(os.path.join(ROOT, "include", "ttk.h")).read()
	declared = set(re.findall(r"\b(ttk_[a-z0-9_]+)\s*\(", header))
	assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
	for name in declared:
		assert hasattr(lib, name), name
	assert _lib.load().ttk_version() == 1


def test_struct_layouts_match_header_sizes():
	assert ctypes.sizeof(_lib.ARConfigC) == 14 * 4
	assert ctypes.sizeof(_lib.DiffConfigC) == 7 * 4
	assert ctypes.sizeof(_lib.StepC) == 8 + 9 * 4 + 2 * 4 + 4      # int64 + 9 floats + 2 ints, padded to 8
	assert ctypes.sizeof(_lib.WeightView) == 8 + 8 + 8 + 32
	assert ctypes.sizeof(_lib.SampleArgs) == 168                    # ttk_sample_args (static_assert'ed in csrc/sample.hip)
	from tortoise_tts_amd.vocoder import VocConfigC
	assert ctypes.sizeof(VocConfigC) == (2 + 8 + 8 + 2 + 4 + 12 + 2) * 4          # ttk_voc_config: 38 ints


@pytest.mark.parametrize("steps", [4, 30, 80, 200])
def test_product_schedule_equals_golden(golden, steps):
	g = golden("schedule")
	s = D.get_diffuser(steps=steps)
	assert s.timestep_map == g[f"map_{steps}"].tolist()
	for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
				"posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2"):
		assert np.array_equal(getattr(s, name), g[f"{name}_{steps}"]), name


def test_step_coefs_follow_reference_float_conversions():
	s = D.get_diffuser(steps=80, cond_free=True)
	o = O.SpacedSchedule(steps=80, cond_free=True)
	for i in (0, 1, 40, 79):
		c = s.step_coefs(i, "ddim")
		assert c.t == o.timestep_map[i]
		assert c.sqrt_recip_ac == float(o._f(o.sqrt_recip_alphas_cumprod, i))
		assert c.sqrt_ac_prev == float(torch.sqrt(o._f(o.alphas_cumprod_prev, i)))
		assert c.sqrt_1m_ac_prev == float(torch.sqrt(1 - o._f(o.alphas_cumprod_prev, i)))
		assert abs(c.cfk - 2 * (1 - i / 80)) < 1e-7
	assert D.get_diffuser(steps=8, cond_free=False).step_coefs(3, "p").cfk < 0


@pytest.mark.parametrize("M,T", [(10, 43), (250, 1088), (48, 208), (7, 7), (5, 10), (500, 2176), (1, 4), (13, 5)])
def test_nearest_index_equals_f_interpolate(M, T):
	src = torch.arange(M, dtype=torch.float32).view(1, 1, M)
	ref = F.interpolate(src, size=T, mode="nearest").view(-1).long()
	assert torch.equal(D.nearest_index(M, T).long(), ref)


def test_relbias_table_equals_relative_position_bias():
	emb = torch.randn(32, 4, generator=torch.Generator().manual_seed(0))
	tab = D.relbias_table(emb, 64)                       # [H, 129]
	full = O.rel_pos_bias(emb, 300, 300, 8.0)            # [H, q, k]
	q = torch.arange(300)[:, None]
	k = torch.arange(300)[None, :]
	idx = (k - q).clamp(-64, 64) + 64
	assert torch.equal(tab[:, idx], full)


def test_logits_pipeline_equals_oracle_processing():
	g = torch.Generator().manual_seed(5)
	scores = torch.randn(3, 8194, generator=g) * 3
	ids = torch.randint(0, 8194, (3, 20), generator=g)
	kw = dict(temperature=0.7, top_k=16, top_p=0.8, repetition_penalty=2.0, suppress_tokens=[8193])
	pipe = sampling.LogitsPipeline(vocab=8194, device="cpu", **kw)
	assert torch.equal(pipe(ids, scores), O.process_logits(ids, scores, **kw))
	pipe = sampling.LogitsPipeline(vocab=8194, device="cpu", temperature=0.8)
	assert not pipe.needs_history and torch.equal(pipe(None, scores), O.process_logits(ids, scores, temperature=0.8))


def test_synthetic_weights_are_deterministic_and_complete():
	a = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 3)
	b = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 3)
	assert all(torch.equal(a[k], b[k]) for k in a)
	assert W.n_params(W.ar_shapes(W.AR_FULL)) == 395978754 and W.n_params(W.diffusion_shapes(W.DIFF_FULL)) > 150e6
	r = W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 3, bf16_exact=True)
	assert torch.equal(r["layers.0.attn.qkv.weight"], r["layers.0.attn.qkv.weight"].bfloat16().float())
	assert r["layers.0.attn.proj_out.weight"].abs().max() > 0     # the reference's zero init would hide the attention


def test_product_path_has_no_cpu_fallback():
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	sd = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 3)
	with pytest.raises(_lib.TTKError):
		UnifiedVoice(sd, W.AR_SMALL, device="cpu")
	src = "".join(open(os.path.join(ROOT, "tortoise_tts_amd", f)).read() for f in os.listdir(os.path.join(ROOT, "tortoise_tts_amd")) if f.endswith(".py"))
	assert "tortoise_oracle" not in src and "import oracle" not in src


def test_bench_launches_its_own_ranks_when_started_without_a_distributed_environment():
	"""`python bench.py --gpus 2` as the driver starts it (no torchrun, no WORLD_SIZE): the parent starts the ranks as child processes under
	torch.distributed.run, relays rank 0's ONE line and exits with their status (VERDICT r02 missing #1).  TTK_BENCH_PROBE=1 makes the ranks
	only rendezvous and count themselves, so the branch runs without a GPU; on the GPU box the same branch runs the benchmark."""
	import json
	import subprocess
	import sys
	sys.path.insert(0, ROOT)
	import bench
	cmd = bench.launcher_command(4, ["--gpus", "4", "--steps", "2"], 12345)
	assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
	assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
	env["TTK_BENCH_PROBE"] = "1"
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], env=env,
					   capture_output=True, text=True, timeout=600)
	assert r.returncode == 0, r.stderr[-2000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
	assert len(lines) == 1, r.stdout
	line = json.loads(lines[0])
	assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["world_size_env"] == 2 and line["steps"] == 2 and line["warmup"] == 1
	assert "starting 2 ranks" in r.stderr
	# a failing rank fails the launcher: without the probe the ranks need a GPU, which this container does not have
	if not torch.cuda.is_available():
		env.pop("TTK_BENCH_PROBE")
		r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--small", "--no-cpu-baseline"],
						   env=env, capture_output=True, text=True, timeout=600)
		assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def _bench_probe(extra_env, *argv, timeout=300):
	import subprocess
	import sys
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
	env.update(TTK_BENCH_PROBE="1", **extra_env)
	return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", *argv], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_multi_rank_run_fails_fast_with_a_reason_when_a_rank_stalls():
	"""the first N > 1 run must either print its line or end non-zero within its stage budget WITH diagnostics.  Two gloo ranks, rank 1
	made to hang in front of a stage marker (TTK_BENCH_STALL_*), stage budgets cut to 5 s: (a) the stalled rank's own watchdog names the marker, dumps the
	thread stacks into gpurun_out/rank1.err and exits 75, torchrun tears the job down, the launcher prints every rank's last lines; (b) with the ranks' own
	watchdogs off, the PARENT-side watchdog terminates the child process group after budget + grace and exits 75.  No result line either way."""
	import time
	t0 = time.time()
	r = _bench_probe(dict(TTK_BENCH_STALL_RANK="1", TTK_BENCH_STALL_AT="timed region", TTK_BENCH_STAGE_BUDGET="5"))
	assert r.returncode != 0 and time.time() - t0 < 90 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
	assert "marker 'timed region' not reached within 5s" in r.stderr and "last lines of rank 0" in r.stderr and "last lines of rank 1" in r.stderr
	assert "stalling in front of 'timed region'" in r.stderr and "in mark" in r.stderr                       # the stack dump shows where the rank stood
	err1 = open(os.path.join(ROOT, "gpurun_out", "rank1.err")).read()
	assert "[watchdog] rank 1/2" in err1 and "[reached] models built" in err1 and "[reached] timed region" not in err1
	t0 = time.time()
	r = _bench_probe(dict(TTK_BENCH_STALL_RANK="1", TTK_BENCH_STALL_AT="models built", TTK_BENCH_STAGE_BUDGET="5", TTK_BENCH_NO_RANK_WATCHDOG="1", TTK_BENCH_PARENT_GRACE="2"))
	assert r.returncode == 75 and time.time() - t0 < 90 and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
	assert "[parent watchdog] rank 1 has not logged 'models built'" in r.stderr and "terminating the 2 child ranks" in r.stderr


def test_bench_parent_teardown_reaches_a_rank_that_ignores_sigterm():
	"""the elastic agent starts its workers in sessions of their own, so killing the launcher's process group does not reach them, and a rank wedged in the driver
	ignores the SIGTERM the agent forwards.  The ranks record their pids in their rank files; the parent signals those itself, SIGKILLs what is still alive after its wait and
	checks that nothing survives.  Rank 1 stalls AND ignores SIGTERM here (wait cut to 3 s): the run ends 75, the log names the SIGKILL, and the recorded pid is gone."""
	import re
	import time
	t0 = time.time()
	r = _bench_probe(dict(TTK_BENCH_STALL_RANK="1", TTK_BENCH_STALL_AT="models built", TTK_BENCH_STALL_IGNORE_TERM="1", TTK_BENCH_STAGE_BUDGET="5",
						  TTK_BENCH_NO_RANK_WATCHDOG="1", TTK_BENCH_PARENT_GRACE="2", TTK_BENCH_TERM_WAIT="3"))
	assert r.returncode == 75 and time.time() - t0 < 120, r.stderr[-1500:]
	err1 = open(os.path.join(ROOT, "gpurun_out", "rank1.err")).read()
	m = re.search(r"\[pid\] rank 1 pid (\d+)", err1)
	assert m and "ignoring SIGTERM" in err1
	pid = int(m.group(1))
	assert f"pid {pid} ignored SIGTERM" in r.stderr and "SIGKILL" in r.stderr and "still alive after SIGKILL" not in r.stderr
	import bench
	time.sleep(0.5)
	assert not bench.alive(pid)


def test_bench_stage_budgets_grow_with_the_steps_asked_for():
	"""a healthy `--steps 100` run must not be ended by a constant budget"""
	import argparse
	import bench
	base = dict(bench.stage_budgets(argparse.Namespace(steps=0, warmup=0, shard="utterances")))
	big = dict(bench.stage_budgets(argparse.Namespace(steps=100, warmup=10, shard="utterances")))
	cand = dict(bench.stage_budgets(argparse.Namespace(steps=100, warmup=10, shard="candidates")))
	assert base == dict(bench.STAGE_BUDGET_S)
	assert big["timed done"] == base["timed done"] + 200 and big["timed region"] == base["timed region"] + 20 and big["rendezvous ok"] == base["rendezvous ok"]
	assert cand["timed done"] == base["timed done"] + 1000


def test_bench_preflight_falls_back_to_the_other_ipc_setting_in_fresh_children():
	"""the one-collective pre-flight runs in a fresh child per attempt; when the inherited HSA_ENABLE_IPC_MODE_LEGACY setting fails it, the other one is tried
	and the run proceeds under the one that passed (stated order: inherited / 0 first); when both fail no model is built and the exit status is 76"""
	import json
	r = _bench_probe(dict(HSA_ENABLE_IPC_MODE_LEGACY="0", TTK_BENCH_PREFLIGHT_FAIL_IPC="0"))
	assert r.returncode == 0, r.stderr[-2000:]
	line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
	assert line["ipc_mode_legacy"] == "1" and line["n_ranks_seen"] == 2
	assert "attempt 0 (HSA_ENABLE_IPC_MODE_LEGACY=0): exit 5" in r.stderr and "attempt 1 (HSA_ENABLE_IPC_MODE_LEGACY=1): exit 0" in r.stderr
	r = _bench_probe(dict(HSA_ENABLE_IPC_MODE_LEGACY="0"))
	assert r.returncode == 0 and json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])["ipc_mode_legacy"] == "0"


def test_bench_phase_roofline_counts_the_work_of_the_configuration_it_is_given():
	"""bench.phase_roofline on fake phase events: configs[1] figures equal BASELINE.md section 4's formulas; the fp8 mode's DDIM phase is graded
	against the 5 PFLOP/s fp8 peak; a two-line configs[3] shard doubles the work and sums the lines' times"""
	import sys
	sys.path.insert(0, ROOT)
	import bench

	class Ev:
		def __init__(self, t): self.t = t
		def elapsed_time(self, other): return other.t - self.t
	marks = [("start", Ev(0.0)), ("ar_decode", Ev(200.0)), ("latent_pass", Ev(210.0)), ("ddim", Ev(350.0))]
	ph = bench.phase_roofline([marks], "bf16")
	T = 1088
	F = 236 * 1024 ** 2 * T + 52 * 1024 * T * T + 1_843_200 * T
	assert ph["ddim"]["flop"] == 160.0 * F and abs(ph["ddim"]["floor_ms"] - 160.0 * F / 2.5e15 * 1e3) < 1e-9
	assert abs(ph["ddim"]["frac"] - 160.0 * F / 0.140 / 2.5e15) < 1e-12 and ph["whole_step_ms"] == 350.0
	assert 35.0 < ph["ar_decode"]["floor_ms"] < 38.0                       # 249 steps x (773 MB of weights + KV + logits) at 8 TB/s + the prefill
	ph8 = bench.phase_roofline([marks], "fp8")
	assert ph8["ddim"]["peak_TFLOPs"] == 5000.0 and abs(ph8["ddim"]["frac"] - ph["ddim"]["frac"] / 2) < 1e-12
	assert abs(ph8["latent_pass"]["frac"] - ph["latent_pass"]["frac"]) < 1e-12
	two = bench.phase_roofline([marks, marks], "bf16", 256, 32, 500, 200)
	one = bench.phase_roofline([marks], "bf16", 256, 32, 500, 200)
	assert two["lines"] == 2 and two["ddim"]["flop"] == 2 * one["ddim"]["flop"] and two["ddim"]["ms"] == 2 * one["ddim"]["ms"]
	assert abs(two["ddim"]["frac"] - one["ddim"]["frac"]) < 1e-12 and two["ar_decode"]["algorithmic_bytes"] == 2 * one["ar_decode"]["algorithmic_bytes"]


def test_bench_effective_floor_follows_its_stated_formula():
	"""`roofline.phases.*.effective_floor_ms` = sum over dependent launches of [boundary + max(hbm bytes / 6.4 TB/s, flop / peak; a GEMM: the MFMA
	phase of its tile at the in-kernel clock, bytes per CU / (L2 intake per clock x clock), staged bytes / the chip's L2 bandwidth)] -- hardware rates only since round 6; round 5's
	pricing of the k-loops (68 GB/s per CU, read off this implementation) stays one round as `launch_chain_model`.  The launch lists and a few hand-computed terms are pinned here."""
	import bench
	assert (bench.BOUNDARY_US, bench.HBM_STREAM, bench.CU_L2_INTAKE) == (1.45, 6.4e12, 68e9)
	assert (bench.CLOCK_HZ, bench.CU_INTAKE_BPC, bench.L2_CHIP) == (2.3e9, {4: 59.0, 8: 72.0}, 34.5e12)
	e = bench.effective_floor("bf16")
	c = bench.effective_floor("bf16", model="launch_chain")
	assert e["launches"] == c["launches"] == {"decode_token": 152, "ddim_step": 123}
	# one decode token at context c: 152 boundaries + (block weights + head + KV cache of 16 candidates + logits) / 6.4 TB/s
	P1 = 68
	want = 0.0
	for k in range(1, 250):
		hbm = 377_886_720 * 2 + (8_398_850 + 4_096) * 2 + 16 * 30 * 2 * (P1 + k) * 1024 * 2 + 16 * 8194 * 4
		want += 152 * 1.45 + hbm / 6.4e12 * 1e6
	assert want * 1e-3 + 0.2 < e["ar_decode_ms"] < want * 1e-3 + 1.5                     # + the prefill's dense pass over 68 rows
	# the 1x1 conv of a DDIM step: 2176 x 1024 x 1024 -> 272 tiles of 128 x 64 on 4 waves, 1.0625 rounds: intake 192 rows x 1024 x 2 B per CU at 59 B/clk x 2.3 GHz (the largest of the
	# three terms: MFMA phase 1.0625 x 2 x 128 x 64 x 1024 / (2.5e15 / 256) x 2.4 / 2.3 = 1.9 us, chip L2 272 x 393216 B / 34.5 TB/s = 3.1 us)
	g = bench._gemm_floor_us(2176, 1024, 1024, 1, 2, 2.5e15)
	intake = (272 / 256) * 192 * 1024 * 2 / (59.0 * 2.3e9) * 1e6
	l2 = 272 * 192 * 1024 * 2 / 34.5e12 * 1e6
	assert abs(g - max(intake, l2)) < 1e-9 and 3.0 < g < 3.2
	assert abs(bench._gemm_floor_us(2176, 1024, 1024, 3, 2, 2.5e15) - 3 * g) < 1e-9       # the k = 3 conv priced as three taps' bytes (the shared image of round 6 stages fewer: the floor is not lowered for it)
	q = bench._gemm_floor_us(2176, 3072, 1024, 1, 2, 2.5e15)                              # QKV: 408 tiles of 128 x 128 -> 216 of 256 x 128, one round, MFMA-phase-bound
	assert abs(q - 2.0 * 256 * 128 * 1024 / (2.5e15 / 256) * (2.4 / 2.3) * 1e6) < 1e-9 and 7.0 < q < 7.3
	# the launch-chain model is round 5's formula, unchanged
	gc = bench._gemm_floor_us(2176, 1024, 1024, 1, 2, 2.5e15, "launch_chain")
	assert abs(gc - (272 / 256) * 192 * 1024 * 2 / 68e9 * 1e6) < 1e-9 and 6.0 < gc < 6.3
	assert abs(bench._gemm_floor_us(2176, 3072, 1024, 1, 2, 2.5e15, "launch_chain") - 384 * 1024 * 2 / 68e9 * 1e6) < 1e-9
	assert 880 < c["ddim_step_us"] < 950 and 560 < e["ddim_step_us"] < 660 and abs(e["ddim_ms"] - (80 * e["ddim_step_us"]) * 1e-3) < 0.2
	f8 = bench.effective_floor("fp8")
	assert f8["ddim_ms"] < e["ddim_ms"] and f8["ar_decode_ms"] < e["ar_decode_ms"] and f8["latent_pass_ms"] == e["latent_pass_ms"]
	big = bench.effective_floor("bf16", 256, 32, 500, 200, 2, 1)
	assert big["ar_decode_ms"] > 4 * e["ar_decode_ms"] and big["ddim_ms"] > 4 * e["ddim_ms"]
	# and the phases carry it next to the spec-peak fraction
	class Ev:
		def __init__(self, t): self.t = t
		def elapsed_time(self, o): return o.t - self.t
	marks = [("start", Ev(0.0)), ("ar_decode", Ev(168.0)), ("latent_pass", Ev(176.0)), ("ddim", Ev(302.0))]
	ph = bench.phase_roofline([marks], "bf16")
	for k in ("ar_decode", "latent_pass", "ddim"):
		assert 0 < ph[k]["frac"] < ph[k]["frac_of_effective_floor"] < 1 and ph[k]["effective_floor_ms"] > ph[k]["floor_ms"]
		assert ph[k]["effective_floor_ms"] <= ph[k]["launch_chain_model_ms"] and "frac_of_launch_chain_model" in ph[k]
	assert abs(ph["whole_step_effective_floor_ms"] - (e["ar_decode_ms"] + e["latent_pass_ms"] + e["ddim_ms"])) < 1e-9
	assert ph["effective_floor"]["constants"]["boundary_us"] == 1.45 and "formula" in ph["effective_floor"]


def test_bench_phase_roofline_on_a_rank_that_diffused_nothing_or_one_of_two_lines():
	"""at N > 1 only the rank a line's diffusion is assigned to has a "ddim" interval for it; the other ranks' marks end with the
	latent pass.  phase_roofline must report that instead of raising KeyError (rank >= 1 of `bench.py --shard candidates` crashed there), and count
	the DDIM work of the lines THIS rank diffused; "_"-prefixed marks stay out of the whole-step sum."""
	import sys
	sys.path.insert(0, ROOT)
	import bench

	class Ev:
		def __init__(self, t): self.t = t
		def elapsed_time(self, other): return other.t - self.t
	no_ddim = [("start", Ev(0.0)), ("ar_decode", Ev(900.0)), ("latent_pass", Ev(950.0))]
	ph = bench.phase_roofline([no_ddim, no_ddim], "bf16", 256, 32, 500, 200)
	assert ph["ddim"]["ms"] is None and ph["ddim"]["frac"] is None and ph["ddim"]["flop"] == 0 and ph["ddim"]["lines_diffused_here"] == 0
	assert ph["ar_decode"]["ms"] == 1800.0 and ph["whole_step_ms"] == 1900.0
	# rank 0 of a 2-line text at N >= 2: line 0 is diffused here (after line 1 was sampled: the wait is bracketed by "_before_ddim"), line 1 elsewhere
	# one rank, both lines diffused as ONE batch: the interval sits on the last line's marks and says that it served two lines
	both = bench.phase_roofline([no_ddim, no_ddim + [("_before_ddim", Ev(1000.0)), ("ddim", Ev(2200.0), 2)]], "bf16", 256, 32, 500, 200)
	one_line = bench.phase_roofline([no_ddim + [("ddim", Ev(1550.0))]], "bf16", 256, 32, 500, 200)
	assert both["ddim"]["lines_diffused_here"] == 2 and both["ddim"]["flop"] == 2 * one_line["ddim"]["flop"] and both["ddim"]["ms"] == 1200.0
	line0 = no_ddim + [("_before_ddim", Ev(1900.0)), ("ddim", Ev(2600.0))]
	ph = bench.phase_roofline([line0, no_ddim], "bf16", 256, 32, 500, 200)
	one = bench.phase_roofline([no_ddim + [("ddim", Ev(1650.0))]], "bf16", 256, 32, 500, 200)
	assert ph["ddim"]["lines_diffused_here"] == 1 and ph["ddim"]["ms"] == 700.0 and ph["ddim"]["flop"] == one["ddim"]["flop"]
	assert abs(ph["ddim"]["frac"] - one["ddim"]["frac"]) < 1e-12 and ph["whole_step_ms"] == 1900.0 + 700.0


def test_every_batch_a_handle_accepts_requests_no_more_fragment_rows_than_create_allocates():
	"""VERDICT r03 next #6a.  Round 3's fault: the 4-tile GEMV instantiation (33..64 rows) requests FOUR sixteen-row tiles of the fragment-order operands
	while max_batch = 48 allocated three.  The allocation and the launchers now use one function (csrc/ttk_kernels.h: decode_row_tiles); this enumerates
	every (dtype, max_batch, rows <= max_batch) pair through the host-only query the library exports -- no GPU call."""
	lib = _lib.load()
	out = (ctypes.c_int32 * 4)()
	seen = set()
	for dtype, cap in ((_lib.DTYPES["f32"], 32), (_lib.DTYPES["bf16"], 64), (_lib.DTYPES["f16"], 64), (_lib.DTYPES["fp8w"], 64)):
		for max_batch in range(1, cap + 1):
			for rows in range(1, max_batch + 1):
				assert lib.ttk_ar_decode_geometry(dtype, max_batch, rows, out) == 0
				alloc, tiles, req, slices = out[0], out[1], out[2], out[3]
				assert tiles in (1, 2, 3, 4) and req == 16 * tiles and req >= rows        # the instantiation covers the batch ...
				assert req <= alloc and slices == max_batch                            # ... and everything it requests exists
				seen.add((max_batch > 32, tiles))
		assert lib.ttk_ar_decode_geometry(dtype, cap + 1, 1, out) != 0 and lib.ttk_ar_decode_geometry(dtype, 8, 9, out) != 0
	assert (True, 4) in seen and lib.ttk_ar_decode_geometry(_lib.DTYPES["bf16"], 48, 48, out) == 0 and out[0] == 48 and out[2] == 48      # round 3's faulting case (now a 3-tile form)
	assert lib.ttk_ar_decode_geometry(_lib.DTYPES["bf16"], 64, 40, out) == 0 and out[0] == 64 and out[1] == 3 and out[2] == 48
