"""Logit processing of the reference's sample branch, as device-side torch ops.

The reference builds these from HuggingFace classes (`_get_logits_warper`,
/root/reference/tortoise_tts/models/stream_generator.py:56-101, order Temperature -> TopK -> TopP; processors from
HF `_get_logits_processor`: repetition penalty, suppress_tokens) and samples with `torch.multinomial` on fp32
probabilities (HF:generation/utils.py:2894-2923).  Sampling stays a torch op on purpose: the drawn ids depend on the
torch generator stream (Philox on the GPU), which is part of the reference's observable behaviour.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch


def repetition_penalty_(input_ids: torch.Tensor, scores: torch.Tensor, penalty: float) -> torch.Tensor:
	score = torch.gather(scores, 1, input_ids)
	score = torch.where(score < 0, score * penalty, score / penalty)
	return scores.scatter(1, input_ids, score)


def typical_filter(scores: torch.Tensor, mass: float, min_tokens_to_keep: int = 1) -> torch.Tensor:
	"""TypicalLogitsWarper, /root/reference/tortoise_tts/models/unified_voice.py:47-75."""
	normalized = torch.nn.functional.log_softmax(scores, dim=-1)
	p = torch.exp(normalized)
	ent = -(normalized * p).nansum(-1, keepdim=True)
	shifted = torch.abs((-normalized) - ent)
	sorted_scores, sorted_indices = torch.sort(shifted, descending=False)
	sorted_logits = scores.gather(-1, sorted_indices)
	cumulative = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
	last_ind = (cumulative < mass).sum(dim=1)
	last_ind[last_ind < 0] = 0
	remove_sorted = sorted_scores > sorted_scores.gather(1, last_ind.view(-1, 1))
	if min_tokens_to_keep > 1:
		remove_sorted[..., :min_tokens_to_keep] = 0
	remove = remove_sorted.scatter(1, sorted_indices, remove_sorted)
	return scores.masked_fill(remove, -float("inf"))


class LogitsPipeline:
	"""repetition penalty -> suppress_tokens -> [typical] -> temperature -> top-k -> top-p (each only when active)."""

	def __init__(self, *, temperature: Optional[float] = 1.0, top_k: Optional[int] = 0, top_p: Optional[float] = 1.0,
				 repetition_penalty: Optional[float] = 1.0, suppress_tokens: Optional[Sequence[int]] = None,
				 typical_mass: Optional[float] = None, vocab: int = 0, device="cuda"):
		self.temperature = None if temperature in (None, 1.0) else float(temperature)
		self.top_k = None if not top_k else int(top_k)
		self.top_p = None if (top_p is None or top_p >= 1.0) else float(top_p)
		self.repetition_penalty = None if repetition_penalty in (None, 1.0) else float(repetition_penalty)
		self.typical_mass = typical_mass
		self.suppress_mask = None
		if suppress_tokens:
			m = torch.zeros(vocab, dtype=torch.bool, device=device)
			m[torch.as_tensor(list(suppress_tokens), device=device)] = True
			self.suppress_mask = m

	@property
	def needs_history(self) -> bool:
		return self.repetition_penalty is not None

	def __call__(self, input_ids: Optional[torch.Tensor], scores: torch.Tensor) -> torch.Tensor:
		s = scores
		if self.repetition_penalty is not None:
			s = repetition_penalty_(input_ids, s, self.repetition_penalty)
		if self.suppress_mask is not None:
			s = torch.where(self.suppress_mask, -float("inf"), s)
		if self.typical_mass is not None:
			s = typical_filter(s, self.typical_mass)
		if self.temperature is not None:
			s = s / self.temperature
		if self.top_k is not None:
			k = min(self.top_k, s.size(-1))
			s = s.masked_fill(s < torch.topk(s, k)[0][..., -1, None], -float("inf"))
		if self.top_p is not None:
			sorted_logits, sorted_indices = torch.sort(s, descending=False)
			cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
			rm = cum <= (1 - self.top_p)
			rm[..., -1:] = 0
			s = s.masked_fill(rm.scatter(1, sorted_indices, rm), -float("inf"))
		return s


def multinomial1(probs: torch.Tensor) -> torch.Tensor:
	"""One draw per row, bit-identical to `torch.multinomial(probs, num_samples=1).squeeze(1)` including the generator stream:
	for n_sample == 1 ATen's multinomial IS `q = empty_like(p).exponential_(1); argmax(p / q)` (aten/src/ATen/native/
	Distributions.cpp, "s = argmax(p / q) where q ~ Exp(1)") preceded by three validity reductions + `_assert_async` on the
	input.  Calling the same three ops directly drops those ~8 tiny launches per token; `tests/test_gpu_parity.py` checks the
	equality on the device so a future change of ATen's algorithm cannot pass unnoticed."""
	q = torch.empty_like(probs).exponential_(1)
	return torch.argmax(probs / q, dim=-1)


def setup_seed(seed: int):
	"""stream_generator.py:38-46 (called with seed=0 on every generate, :223,:296)."""
	import random

	import numpy as np
	if seed == -1:
		return
	torch.manual_seed(seed)
	if torch.cuda.is_available():
		torch.cuda.manual_seed_all(seed)
	np.random.seed(seed)
	random.seed(seed)
