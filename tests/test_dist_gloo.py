"""N>1 path on CPU: 2 gloo ranks exercising the candidate partition and the id all-gather of tortoise_tts_amd/dist.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tortoise_tts_amd import dist as D


def test_candidate_shards_partition_exactly():
	for n in (1, 7, 16, 256):
		for world in (1, 2, 3, 8):
			spans = [D.candidate_shard(n, r, world) for r in range(world)]
			assert spans[0][0] == 0 and spans[-1][1] == n
			assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
			sizes = [hi - lo for lo, hi in spans]
			assert max(sizes) - min(sizes) <= 1
	assert D.utterance_shard(8, 1, 4) == [1, 5]
	with pytest.raises(ValueError):
		D.candidate_shard(4, 2, 2)


def _worker(rank, world, port, q):
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		n = 5
		lo, hi = D.candidate_shard(n, rank, world)
		L = 4 + rank                                        # ragged lengths: ranks stop at different steps
		local = torch.arange(lo, hi)[:, None] * 100 + torch.arange(L)[None, :]
		ids = D.gather_candidate_ids(local, n, pad_token=8193)
		# every rank scores its own shard; only the scores travel.  Candidate 3 (on rank 1) wins; a tie with candidate 4 goes to the first.
		all_scores = torch.tensor([0.1, 0.7, -0.2, 0.9, 0.9])
		owner, idx, gathered = D.pick_best_candidate(all_scores[lo:hi], n)
		assert torch.equal(gathered, all_scores) and (owner, idx) == (1, 3 - D.candidate_shard(n, 1, world)[0])
		q.put((rank, ids))
	finally:
		dist.destroy_process_group()


class _FakeStages(D.ShardStages):
	"""dist.ShardStages on plain CPU tensors: candidate c has a deterministic length, ids, latents and score, so the sharded control flow
	can be checked against the unsharded result computed directly."""
	pad_token = 8193
	SCORES = torch.tensor([0.3, -1.0, 2.5, 0.1, 2.5, 0.7, -0.4])       # a tie between candidates 2 and 4: the first wins

	SCORES_LATE = torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 9.0, 0.0])    # the winner lives on the last rank

	def __init__(self, with_scorer):
		self.with_scorer, self.calls, self.aligned = with_scorer, [], None

	@staticmethod
	def row(c, L):
		n = 3 + (c * 5) % 4                                             # ragged: candidates stop at different steps
		r = torch.full((L,), 8193, dtype=torch.long)
		r[:min(n, L)] = c * 100 + torch.arange(min(n, L))
		return r

	def sample(self, lo, hi, n_candidates):
		L = max(3 + (c * 5) % 4 for c in range(lo, hi))                 # the shard's loop ends with ITS last row
		self.lo = lo
		self.calls.append(("sample", lo, hi))
		return torch.stack([self.row(c, L) for c in range(lo, hi)])

	def align_rng(self, steps):
		self.aligned = steps

	def latents(self, ids):
		return ids.clone(), ids[:, :, None].float() * torch.tensor([1.0, 0.5])

	def score(self, codes):
		if not self.with_scorer:
			return None
		table = self.SCORES_LATE if self.with_scorer == "late" else self.SCORES
		return table[self.lo:self.lo + codes.shape[0]]

	def diffuse(self, codes, latents):
		self.calls.append(("diffuse", int(codes[0, 0]) // 100))
		return (latents.sum(dim=(1, 2)).view(1, 1, 1) + torch.arange(12.0).view(1, 3, 4))


def _worker_sharded(rank, world, port, q):
	os.environ["MASTER_ADDR"] = "127.0.0.1"
	os.environ["MASTER_PORT"] = str(port)
	dist.init_process_group("gloo", rank=rank, world_size=world)
	try:
		out = {}
		for with_scorer in (True, False, "late"):
			st = _FakeStages(with_scorer)
			mel, ids, scores, best = D.sharded_candidates(st, 7)
			out[with_scorer] = dict(mel=mel, ids=ids, scores=scores, best=best, calls=st.calls, aligned=st.aligned)
		q.put((rank, out))
	finally:
		dist.destroy_process_group()


def _run_two_ranks(target=None):
	target = target or _worker
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	ctx = mp.get_context("spawn")
	q = ctx.Queue()
	procs = [ctx.Process(target=target, args=(r, 2, port, q)) for r in range(2)]
	for p in procs:
		p.start()
	try:
		got = dict(q.get(timeout=120) for _ in range(2))
	finally:
		for p in procs:
			p.join(timeout=60)
			if p.is_alive():
				p.kill()
	assert all(p.exitcode == 0 for p in procs)
	return got


def test_gather_candidate_ids_two_ranks():
	got = None
	for attempt in range(3):          # the rendezvous port is picked by bind(0) and released: retry if something else grabbed it
		try:
			got = _run_two_ranks()
			break
		except (EOFError, AssertionError, OSError):
			if attempt == 2:
				raise
	assert torch.equal(got[0], got[1])
	ids = got[0]
	assert ids.shape == (5, 5)
	assert ids[0].tolist() == [0, 1, 2, 3, 8193]             # rank 0 rows: length 4, padded
	assert ids[4].tolist() == [400, 401, 402, 403, 404]       # rank 1 rows: length 5


def test_sharded_candidates_control_flow_two_ranks():
	"""dist.sharded_candidates under gloo, 2 ranks, 7 candidates (shards of 4 and 3): ids gathered in candidate order and padded, the RNG
	aligned to the longest shard, every rank scoring its own candidates, the first maximum winning, ONLY its owner diffusing, and every
	rank ending with the same mel -- all equal to the unsharded computation."""
	got = None
	for attempt in range(3):
		try:
			got = _run_two_ranks(_worker_sharded)
			break
		except (EOFError, AssertionError, OSError):
			if attempt == 2:
				raise
	Lmax = max(3 + (c * 5) % 4 for c in range(7))
	want_ids = torch.stack([_FakeStages.row(c, Lmax) for c in range(7)])
	for with_scorer, best in ((True, 2), (False, 0), ("late", 5)):
		a, b = got[0][with_scorer], got[1][with_scorer]
		assert torch.equal(a["ids"], want_ids) and torch.equal(b["ids"], want_ids)
		assert a["best"] == b["best"] == best and a["aligned"] == b["aligned"] == Lmax
		if with_scorer:
			table = _FakeStages.SCORES_LATE if with_scorer == "late" else _FakeStages.SCORES
			assert torch.equal(a["scores"], table) and torch.equal(b["scores"], table)
		else:
			assert a["scores"] is None and b["scores"] is None
		want_mel = (want_ids[best].float().sum() * 1.5).view(1, 1, 1) + torch.arange(12.0).view(1, 3, 4)
		assert torch.equal(a["mel"], want_mel) and torch.equal(b["mel"], want_mel)
		# candidates 0..3 live on rank 0, 4..6 on rank 1: only the winner's owner diffuses, the other rank receives the mel
		if best < 4:
			assert a["calls"] == [("sample", 0, 4), ("diffuse", best)] and b["calls"] == [("sample", 4, 7)]
		else:
			assert a["calls"] == [("sample", 0, 4)] and b["calls"] == [("sample", 4, 7), ("diffuse", best)]
