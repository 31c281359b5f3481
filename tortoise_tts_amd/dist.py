"""Multi-GPU sharding of the hot path: one process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  The path partitions by (utterance, candidate) with replicated weights and NO data-path collective.  Two forms
of the one exchange: `gather_candidate_ids` (an all-gather of the sampled ids, for a scorer that lives on one rank -- the reference
path's CLVP) and `pick_best_candidate` (every rank scores its own shard with its CLVP replica -- scores are per candidate, so no ids
need to travel -- and only the scores are all-gathered; the owner of the winner runs the diffusion).
`sharded_candidates` is the product entry for ONE utterance whose candidates are spread over the ranks (BASELINE configs[3]: 256 candidates
over 8 GPUs); `TTSHotPath.inference_sharded` binds it to the libttk-backed stages.
The reference itself has no multi-GPU inference ("Inference multi-GPU: none"): this is new design.

RNG contract of a candidate shard (defined here because the reference has none): every rank reseeds to 0 (`generate`,
stream_generator.py:296), draws the multinomial noise of ALL C candidates for every token -- the identical Philox stream on every rank --
and consumes the rows of its own candidates; after the id gather every rank advances its generator to where the unsharded loop would
have left it (the longest shard's step count).  The union of the shards is therefore bit for bit the single-GPU result for C candidates,
the winner's diffusion start noise included, for any number of ranks.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

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


def gather_candidate_ids(local_ids: torch.Tensor, n_candidates: int, pad_token: int, group=None) -> torch.Tensor:
	"""All-gather of per-rank id blocks [c_r, L_r] -> [n_candidates, max L] on every rank of `group` (None = the default group), rows in
	candidate order (= group-rank order), padded with `pad_token` (the stop token, as `generate` pads finished rows).  Ranks may hold
	different candidate counts and lengths."""
	world = dist.get_world_size(group)
	rank = dist.get_rank(group)
	dev = local_ids.device
	shape = torch.tensor([local_ids.shape[0], local_ids.shape[1]], dtype=torch.long, device=dev)
	shapes = [torch.zeros_like(shape) for _ in range(world)]
	dist.all_gather(shapes, shape, group=group)
	cmax = int(max(s[0] for s in shapes))
	lmax = int(max(s[1] for s in shapes))
	buf = torch.full((cmax, lmax), pad_token, dtype=torch.long, device=dev)
	buf[: local_ids.shape[0], : local_ids.shape[1]] = local_ids
	out = [torch.empty_like(buf) for _ in range(world)]
	dist.all_gather(out, buf, group=group)
	rows = [out[r][: int(shapes[r][0])] for r in range(world)]
	ids = torch.cat(rows, dim=0)
	assert ids.shape[0] == n_candidates, (ids.shape, n_candidates, rank)
	return ids


def pick_best_candidate(local_scores: torch.Tensor, n_candidates: int, group=None) -> Tuple[int, int, torch.Tensor]:
	"""All-gather of per-rank score blocks [c_r] over `group` -> (owner = rank INSIDE the group, index inside the owner's shard, all scores
	[n_candidates] in candidate order).  The winner is the first maximum in candidate order (`torch.argmax` on the gathered vector),
	identical on every rank of the group."""
	world = dist.get_world_size(group)
	dev = local_scores.device
	n = torch.tensor([local_scores.shape[0]], dtype=torch.long, device=dev)
	counts = [torch.zeros_like(n) for _ in range(world)]
	dist.all_gather(counts, n, group=group)
	cmax = int(max(c[0] for c in counts))
	buf = torch.full((cmax,), float("-inf"), dtype=torch.float32, device=dev)
	buf[: local_scores.shape[0]] = local_scores.to(torch.float32)
	out = [torch.empty_like(buf) for _ in range(world)]
	dist.all_gather(out, buf, group=group)
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


class ShardStages:
	"""The per-rank pieces `sharded_candidates` strings together.  `tortoise_tts_amd.inference.HotPathStages` implements them on libttk;
	the CPU tests implement them on plain tensors to run the control flow under gloo."""

	pad_token: int = 0
	# does `run_diffusion` draw from the rank's generator WHILE it runs (an ancestral sampler's per-step noise)?  Then only the winner's owner has the generator
	# state the unsharded run would have, and the line must not be handed to another rank (only the start noise travels with a prepared item)
	diffusion_draws_while_running: bool = False

	def sample(self, lo: int, hi: int, n_candidates: int) -> torch.Tensor:
		"""ids [hi - lo, L_r] of candidates lo..hi-1 (rows lo..hi-1 of the unsharded sampling, ending with this shard's last row)"""
		raise NotImplementedError

	def align_rng(self, steps: int) -> None:
		"""leave the generator where the unsharded loop of `steps` tokens would"""
		raise NotImplementedError

	def latents(self, ids: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
		"""(codes after the stop-token fix-up, latents [c, L, d]) of this rank's candidates"""
		raise NotImplementedError

	def score(self, codes: torch.Tensor) -> Optional[torch.Tensor]:
		"""CLVP scores [c] of this rank's candidates, or None when no scorer is attached (candidate 0 is diffused then)"""
		return None

	def diffuse(self, codes: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:
		"""mel [1, 100, T] of ONE candidate (codes [1, L], latents [1, L, d]); only the winner's owner calls this"""
		raise NotImplementedError

	# several lines of one text (sharded_candidates_lines): the owner's random draws stay where `diffuse` would make them, the loops run together
	def prepare_diffusion(self, codes: torch.Tensor, latents: torch.Tensor):
		"""everything of `diffuse` that draws from the generator (start noise ...), made right after this line's sampling; returns an opaque item"""
		return (codes, latents)

	def run_diffusion(self, prepared) -> List[torch.Tensor]:
		"""mels of the prepared items, as one batch where the implementation can; default: one `diffuse` per item"""
		return [self.diffuse(c, l) for c, l in prepared]

	# a prepared item travels when its diffusion is assigned to another rank than the winner's owner (assign_diffusers)
	def pack_prepared(self, prepared) -> List[torch.Tensor]:
		"""the tensors of a prepared item, in an order `unpack_prepared` understands"""
		return list(prepared)

	def unpack_prepared(self, tensors: List[torch.Tensor]):
		return tuple(tensors)


def assign_diffusers(owners: List[int], world: int) -> List[int]:
	"""Which rank diffuses which line.  A text's lines are independent DDIM loops (utterances / lines shard naturally), so they are
	spread over the ranks instead of piling up on the winners' owners (without a scorer every line's winner is candidate 0 = rank 0, and the other
	ranks idled through more than half of a configs[3] step).  Greedy in line order, a pure function of (owners, world) so every rank computes the same
	answer without talking: a line stays on its owner while the owner is among the least loaded ranks (nothing travels), else it goes to the least
	loaded rank, the nearest one after the owner first.  Lines that land on one rank still run as one ragged batch there."""
	load = [0] * world
	out = []
	for o in owners:
		m = min(load)
		r = o if load[o] == m else min((q for q in range(world) if load[q] == m), key=lambda q: (q - o) % world)
		load[r] += 1
		out.append(r)
	return out


_DTYPES = (torch.float32, torch.int64, torch.int32, torch.float64, torch.bfloat16, torch.float16)


def _move_item(tensors: Optional[List[torch.Tensor]], src: int, dev, group=None) -> List[torch.Tensor]:
	"""a prepared item from group rank `src` to every rank of the group (the receiver that needs it keeps it): one header broadcast (count, dtype,
	shape) + one broadcast per tensor, on the communicator the mel broadcast uses anyway.  <= 2 MB per line (latents [1, L, d] + start noise [1, 100, T])."""
	rank = dist.get_rank(group)
	gsrc = dist.get_global_rank(group, src) if group is not None else src
	words = [0] * (1 + 8 * 6)
	if rank == src:
		assert len(tensors) <= 8 and all(t.dim() <= 4 for t in tensors)
		words[0] = len(tensors)
		for i, t in enumerate(tensors):
			words[1 + 6 * i], words[2 + 6 * i] = _DTYPES.index(t.dtype), t.dim()
			words[3 + 6 * i: 3 + 6 * i + t.dim()] = list(t.shape)
	head = torch.tensor(words, dtype=torch.long).to(dev)      # (built on the host: one copy, not one per word)
	dist.broadcast(head, src=gsrc, group=group)
	h = head.tolist()
	out = []
	for i in range(h[0]):
		shape = h[3 + 6 * i: 3 + 6 * i + h[2 + 6 * i]]
		t = tensors[i].to(dev).contiguous() if rank == src else torch.empty(shape, dtype=_DTYPES[h[1 + 6 * i]], device=dev)
		dist.broadcast(t, src=gsrc, group=group)
		out.append(t)
	return out


def sharded_candidates(stages: ShardStages, n_candidates: int, group=None):
	"""One utterance, candidates sharded over the ranks of `group` (weights replicated, no data-path collective):
	   every rank samples its `candidate_shard` -> all-gather of the ids (the only payload that is not a scalar per candidate; 8 B x L
	   per candidate) -> every rank runs the latent pass and its CLVP replica on its own candidates -> all-gather of the scores ->
	   the owner of the best candidate diffuses it -> the mel is broadcast.
	Returns (mel [1, 100, T], ids [C, L], scores [C] or None, best) on every rank, equal to the single-rank result for C candidates."""
	world = dist.get_world_size(group)
	rank = dist.get_rank(group)
	lo, hi = candidate_shard(n_candidates, rank, world)
	if hi <= lo:
		raise ValueError(f"{n_candidates} candidates over {world} ranks leaves rank {rank} without work")
	local = stages.sample(lo, hi, n_candidates)
	ids = gather_candidate_ids(local, n_candidates, stages.pad_token, group)
	stages.align_rng(ids.shape[1])
	codes, lat = stages.latents(ids[lo:hi].contiguous())
	sc = stages.score(codes)
	if sc is None:
		owner, idx, scores, best = 0, 0, None, 0
	else:
		owner, idx, scores = pick_best_candidate(sc, n_candidates, group)
		best = candidate_shard(n_candidates, owner, world)[0] + idx
	dev = ids.device
	shape = torch.zeros(3, dtype=torch.long, device=dev)
	mel = None
	if rank == owner:
		mel = stages.diffuse(codes[idx:idx + 1], lat[idx:idx + 1]).to(torch.float32).contiguous()
		shape = torch.tensor(mel.shape, dtype=torch.long, device=dev)
	src = dist.get_global_rank(group, owner) if group is not None else owner
	dist.broadcast(shape, src=src, group=group)
	if rank != owner:
		mel = torch.empty([int(v) for v in shape], dtype=torch.float32, device=dev)
	dist.broadcast(mel, src=src, group=group)
	return mel, ids, scores, best


def sharded_candidates_lines(stages_list, n_candidates: int, group=None, spread: bool = True):
	"""`sharded_candidates` for the lines of one text: per line the same exchange (shard sampling, id all-gather, RNG alignment, latent pass and
	scores on every shard, first maximum wins), with the winner's owner making that line's random draws at once (`prepare_diffusion`: the next
	line's `generate` reseeds, so the start noise is drawn where the single-GPU run draws it and TRAVELS with the item) -- then the lines'
	diffusions are spread over the ranks (`assign_diffusers`; spread=False keeps every line on its winner's owner; a stage with `diffusion_draws_while_running` -- the ancestral sampler --
	is diffused on its winner's owner AT ONCE, behind its own start noise, and takes no part in the spreading): a line assigned to another rank
	than its owner has its prepared item (latents + start noise, <= 2 MB) broadcast inside the group, every rank runs ONE diffusion over the lines
	assigned to it (`run_diffusion`: a ragged batch on libttk), and the mels are broadcast line by line from where they were made.
	Returns [(mel, ids, scores, best)] per line, each equal to that line's own `sharded_candidates` result."""
	world = dist.get_world_size(group)
	rank = dist.get_rank(group)
	picked = []
	for st in stages_list:
		lo, hi = candidate_shard(n_candidates, rank, world)
		if hi <= lo:
			raise ValueError(f"{n_candidates} candidates over {world} ranks leaves rank {rank} without work")
		local = st.sample(lo, hi, n_candidates)
		ids = gather_candidate_ids(local, n_candidates, st.pad_token, group)
		st.align_rng(ids.shape[1])
		codes, lat = st.latents(ids[lo:hi].contiguous())
		sc = st.score(codes)
		if sc is None:
			owner, idx, scores, best = 0, 0, None, 0
		else:
			owner, idx, scores = pick_best_candidate(sc, n_candidates, group)
			best = candidate_shard(n_candidates, owner, world)[0] + idx
		prep = st.prepare_diffusion(codes[idx:idx + 1], lat[idx:idx + 1]) if rank == owner else None
		home = bool(getattr(st, "diffusion_draws_while_running", False))
		mel_now = None
		if home and rank == owner:
			# A sampler that draws per-step noise inside the loop (sampler="p") consumes the generator WHILE it runs: its loop must run here, right behind this line's
			# start noise and in front of the next line's reseed / sampling / start noise -- the generator state its own `sharded_candidates` call (and the single-GPU
			# per-line path) runs it in -- not in the batch behind the last line's prepare (every line but the last drew different noise there).
			mel_now = st.run_diffusion([prep])[0].to(torch.float32).contiguous()
			prep = None
		picked.append(dict(owner=owner, ids=ids, scores=scores, best=best, prep=prep, mel_now=mel_now, home=home))
	owners = [p["owner"] for p in picked]
	# such lines stay on their winner's owner (already diffused above); the others are spread with their prepared item (latents + start noise)
	movable = [k for k, p in enumerate(picked) if not p["home"]]
	spread_to = assign_diffusers([owners[k] for k in movable], world) if spread and movable else [owners[k] for k in movable]
	diffusers = list(owners)
	for k, d in zip(movable, spread_to):
		diffusers[k] = d
	for k, (p, st) in enumerate(zip(picked, stages_list)):
		p["diffuser"] = diffusers[k]
		if diffusers[k] != p["owner"]:      # every rank takes part in the broadcast; only the assigned rank keeps the item
			moved = _move_item(st.pack_prepared(p["prep"]) if rank == p["owner"] else None, p["owner"], p["ids"].device, group)
			p["prep"] = st.unpack_prepared(moved) if rank == diffusers[k] else None
	mine = [k for k, p in enumerate(picked) if p["diffuser"] == rank and not p["home"]]
	mels = {k: p["mel_now"] for k, p in enumerate(picked) if p["mel_now"] is not None}
	if mine:
		# (the LAST assigned line's stages run the batch: their phase marks then end with the shared diffusion, right behind that line's own stages)
		for k, m in zip(mine, stages_list[mine[-1]].run_diffusion([picked[k]["prep"] for k in mine])):
			mels[k] = m.to(torch.float32).contiguous()
	out = []
	for k, p in enumerate(picked):
		dev = p["ids"].device
		shape = torch.tensor(mels[k].shape, dtype=torch.long, device=dev) if k in mels else torch.zeros(3, dtype=torch.long, device=dev)
		src = dist.get_global_rank(group, p["diffuser"]) if group is not None else p["diffuser"]
		dist.broadcast(shape, src=src, group=group)
		mel = mels[k] if k in mels else torch.empty([int(v) for v in shape], dtype=torch.float32, device=dev)
		dist.broadcast(mel, src=src, group=group)
		out.append((mel, p["ids"], p["scores"], p["best"]))
	return out
