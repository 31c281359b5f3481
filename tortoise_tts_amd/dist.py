"""Multi-GPU sharding of the hot path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  The path partitions by (utterance, candidate) with replicated weights and NO data-path collective.  Two forms
of the one exchange: `gather_candidate_ids` (an all-gather of the sampled ids, for a scorer that lives on one rank -- the reference
path's CLVP) and `pick_best_candidate` (every rank scores its own shard with its CLVP replica -- scores are per candidate, so no ids
need to travel -- and only the scores are all-gathered; the owner of the winner runs the diffusion).
The reference itself has no multi-GPU inference ("Inference multi-GPU: none"): this is new design.
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def candidate_shard(n_candidates: int, rank: int, world: int) -> Tuple[int, int]:
	"""[lo, hi) of the candidates rank `rank` samples: contiguous, sizes differing by at most one."""
	if world < 1 or not (0 <= rank < world):
		raise ValueError(f"bad rank {rank} / world {world}")
	base, extra = divmod(n_candidates, world)
	lo = rank * base + min(rank, extra)
	return lo, lo + base + (1 if rank < extra else 0)


def utterance_shard(n_utterances: int, rank: int, world: int) -> List[int]:
	"""Indices of the utterances rank `rank` owns (round-robin: equal work when lengths are i.i.d.)."""
	return list(range(rank, n_utterances, world))


def gather_candidate_ids(local_ids: torch.Tensor, n_candidates: int, pad_token: int) -> torch.Tensor:
	"""All-gather of per-rank id blocks [c_r, L_r] -> [n_candidates, max L] on every rank, rows in candidate order, padded with
	`pad_token` (the stop token, as `generate` pads finished rows).  Ranks may hold different candidate counts and lengths."""
	world = dist.get_world_size()
	rank = dist.get_rank()
	dev = local_ids.device
	shape = torch.tensor([local_ids.shape[0], local_ids.shape[1]], dtype=torch.long, device=dev)
	shapes = [torch.zeros_like(shape) for _ in range(world)]
	dist.all_gather(shapes, shape)
	cmax = int(max(s[0] for s in shapes))
	lmax = int(max(s[1] for s in shapes))
	buf = torch.full((cmax, lmax), pad_token, dtype=torch.long, device=dev)
	buf[: local_ids.shape[0], : local_ids.shape[1]] = local_ids
	out = [torch.empty_like(buf) for _ in range(world)]
	dist.all_gather(out, buf)
	rows = [out[r][: int(shapes[r][0])] for r in range(world)]
	ids = torch.cat(rows, dim=0)
	assert ids.shape[0] == n_candidates, (ids.shape, n_candidates, rank)
	return ids


def pick_best_candidate(local_scores: torch.Tensor, n_candidates: int) -> Tuple[int, int, torch.Tensor]:
	"""All-gather of per-rank score blocks [c_r] -> (owner rank, index inside the owner's shard, all scores [n_candidates] in candidate
	order).  The winner is the first maximum in candidate order (`torch.argmax` on the gathered vector), identical on every rank."""
	world = dist.get_world_size()
	dev = local_scores.device
	n = torch.tensor([local_scores.shape[0]], dtype=torch.long, device=dev)
	counts = [torch.zeros_like(n) for _ in range(world)]
	dist.all_gather(counts, n)
	cmax = int(max(c[0] for c in counts))
	buf = torch.full((cmax,), float("-inf"), dtype=torch.float32, device=dev)
	buf[: local_scores.shape[0]] = local_scores.to(torch.float32)
	out = [torch.empty_like(buf) for _ in range(world)]
	dist.all_gather(out, buf)
	scores = torch.cat([out[r][: int(counts[r][0])] for r in range(world)])
	assert scores.shape[0] == n_candidates, (scores.shape, n_candidates)
	best = int(torch.argmax(scores))
	lo = 0
	for r in range(world):
		c = int(counts[r][0])
		if best < lo + c:
			return r, best - lo, scores
		lo += c
	raise AssertionError("unreachable")
