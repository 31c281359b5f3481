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


def _run_two_ranks():
	with socket.socket() as s:
		s.bind(("127.0.0.1", 0))
		port = s.getsockname()[1]
	ctx = mp.get_context("spawn")
	q = ctx.Queue()
	procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
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
