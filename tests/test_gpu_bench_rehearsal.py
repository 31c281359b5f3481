"""`bench.py --gpus 2 --shard candidates` end to end with the roofline pass ON, two ranks on the one GPU of the test box (TTK_BENCH_REHEARSAL=1: gloo
instead of RCCL, both ranks on cuda:0; not a scaling measurement).  ADVICE r03 (high): every rank >= 1 of this configuration used to die with KeyError
'ddim' in the roofline block -- only the rank a line's diffusion runs on has that interval -- and torchrun then tore the job down without a result line.
Round 4 also spreads the two lines' diffusions over the two ranks (dist.assign_diffusers), so rank 0 reports ONE diffused line.  GPU only."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_candidate_sharded_bench_prints_one_line_with_phases():
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
	env["TTK_BENCH_REHEARSAL"] = "1"
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shard", "candidates", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
					   env=env, capture_output=True, text=True, timeout=900)
	assert r.returncode == 0, r.stderr[-3000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
	assert len(lines) == 1, r.stdout[-2000:]
	line = json.loads(lines[0])
	assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["backend"] == "gloo" and line["config"]["lines"] == 2
	ph = line["roofline"]["phases"]
	assert ph["lines"] == 2 and ph["ddim"]["lines_diffused_here"] == 1 and ph["ddim"]["ms"] > 0 and 0 < ph["ddim"]["frac"] < 1
	assert ph["ar_decode"]["ms"] > 0 and ph["latent_pass"]["ms"] > 0 and line["value"] > 0


def test_two_rank_utterance_sharded_bench_prints_one_line():
	"""the driver's N > 1 form (configs[2]: one utterance per rank, the candidate ids all-gathered), two ranks on one GPU over gloo: one result line from rank 0,
	whole-job value = 2 utterances' audio over the slower rank's time"""
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
	env["TTK_BENCH_REHEARSAL"] = "1"
	r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
					   env=env, capture_output=True, text=True, timeout=900)
	assert r.returncode == 0, r.stderr[-3000:]
	lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
	assert len(lines) == 1, r.stdout[-2000:]
	line = json.loads(lines[0])
	assert line["n_gpus"] == 2 and line["n_ranks_seen"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
	audio = 2 * 1088 * 256 / 24000
	assert abs(line["value"] - audio / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
	assert line["roofline"]["phases"]["ddim"]["ms"] > 0 and line["cpu_baseline"] is None


def test_preflight_child_on_rccl_with_one_rank():
	"""the one-collective pre-flight child of an N > 1 rank (`bench.py --preflight`), here on the backend the real run uses -- RCCL -- with the one rank a 1-GPU box allows: rendezvous on its
	own port, all_reduce of the rank ids on cuda:0, sum checked, exit 0; RCCL's own WARN lines go to NCCL_DEBUG_FILE, not to stdout (round 5)"""
	import tempfile
	env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TTK_BENCH_REHEARSAL", "TTK_BENCH_PROBE")}
	with tempfile.TemporaryDirectory() as td:
		env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN", NCCL_DEBUG_FILE=os.path.join(td, "rccl.log"))
		r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--preflight"], env=env, capture_output=True, text=True, timeout=300)
	assert r.returncode == 0, r.stderr[-2000:]
	assert "all_reduce of the rank ids = 0.0 (want 0.0)" in r.stderr and "NCCL WARN" not in r.stdout
