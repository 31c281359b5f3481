"""CPU oracle: a functional fp32 restatement of the reference's inference hot path.

THIS FILE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import it; the product path (`tortoise_tts_amd/`) never does and
fails loudly when its HIP library is missing.

What it restates (own code, written from reading the reference; every function cites the lines it
follows, paths relative to /root/reference, `HF:` = transformers 5.15.0):

  a1-a4,a7  UnifiedVoice / GPT2InferenceModel / GPT-2 block     tortoise_tts/models/unified_voice.py
  a5        generate() sample loop + logits warpers             tortoise_tts/models/stream_generator.py, HF
  a8        stop/calm token post-processing                     tortoise_tts/inference.py:353-389
  a9-a13    DiffusionTTS (timestep_independent, forward)        tortoise_tts/models/diffusion.py, arch_utils.py
  a14-a16   Gaussian/Spaced diffusion schedule, DDIM + p sampler  tortoise_tts/models/diffusion.py
  a17       denormalize_tacotron_mel                             tortoise_tts/models/arch_utils.py:532-537

Pinning: the reference has no tests or golden vectors of its own, so the oracle
is pinned against the reference itself run in the build container: `oracle/make_golden.py` imports
the reference modules by path (`oracle/ref_shim.py`), loads the same seeded synthetic weights into
them and stores inputs + outputs under `tests/golden/`; `tests/test_oracle_golden.py` checks this file
against those vectors.  The one part whose reference code cannot run here is the `generate()` loop (a5): the
reference's fork of HuggingFace's loop raises AttributeError on transformers 5.15.  It is
pinned against what it forks instead: the warpers against the installed HF classes the reference instantiates,
and the loop (processor order, softmax + multinomial and the generator stream it consumes, pad-after-EOS,
stopping, max_length, GenerationConfig defaults) against the installed `GenerationMixin.generate` driving a
model-free stub (`oracle/stub_lm.py`, fixture `tests/golden/hf_sample_loop.npz`, `tests/test_oracle_sampling.py`).
`setup_seed` and the fake id row handed to generate() are pinned against the reference run here
(`tests/golden/wrapper.npz`).  What stays restated from source only, and is said so in DESIGN.md, are two lines of
`inference_speech`: `max_length = trunc_index + max_generate_length` (unified_voice.py:660) and the final slice
`gen[:, trunc_index:]` (:668).

Arithmetic is plain torch fp32 on the CPU (a floating-point path: the torch reference the task
keeps for floating-point kernels); schedule tables are numpy float64 exactly as the reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor
W = Dict[str, Tensor]

# --------------------------------------------------------------------------------------------
# GPT-2 stack (third-party arithmetic: HF:models/gpt2/modeling_gpt2.py)
# --------------------------------------------------------------------------------------------


def gelu_new(x: Tensor) -> Tensor:
	"""HF:activations.py:59-66 NewGELUActivation."""
	return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * torch.pow(x, 3.0))))


def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
	return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def conv1d_hf(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
	"""HF:pytorch_utils.py:95-120 Conv1D: weight is [in, out], y = addmm(b, x, W)."""
	return torch.addmm(b, x.reshape(-1, x.shape[-1]), w).reshape(*x.shape[:-1], w.shape[1])


def gpt2_block(w: W, i: int, x: Tensor, heads: int, past: Optional[Tuple[Tensor, Tensor]]):
	"""HF:models/gpt2/modeling_gpt2.py:246-310 GPT2Block (pre-LN), :144-226 GPT2Attention
	(scale 1/sqrt(head_dim), causal), :229-243 GPT2MLP.  `past` = (k, v) each [B, H, ctx, hd];
	returns (x, (k, v)) with the new rows appended (DynamicCache.update)."""
	p = f"gpt.h.{i}."
	B, q_len, d = x.shape
	hd = d // heads
	h = layer_norm(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"])
	qkv = conv1d_hf(h, w[p + "attn.c_attn.weight"], w[p + "attn.c_attn.bias"])
	q, k, v = qkv.split(d, dim=2)
	q = q.view(B, q_len, heads, hd).transpose(1, 2)
	k = k.view(B, q_len, heads, hd).transpose(1, 2)
	v = v.view(B, q_len, heads, hd).transpose(1, 2)
	if past is not None:
		k = torch.cat([past[0], k], dim=2)
		v = torch.cat([past[1], v], dim=2)
	ctx = k.shape[2]
	att = torch.matmul(q, k.transpose(-1, -2)) / math.sqrt(hd)
	# causal: query row r (absolute position ctx - q_len + r) sees keys <= its position
	qpos = torch.arange(ctx - q_len, ctx).view(q_len, 1)
	kpos = torch.arange(ctx).view(1, ctx)
	att = att.masked_fill(kpos > qpos, float("-inf"))
	att = torch.softmax(att, dim=-1)
	a = torch.matmul(att, v).transpose(1, 2).reshape(B, q_len, d)
	x = x + conv1d_hf(a, w[p + "attn.c_proj.weight"], w[p + "attn.c_proj.bias"])
	h = layer_norm(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"])
	h = gelu_new(conv1d_hf(h, w[p + "mlp.c_fc.weight"], w[p + "mlp.c_fc.bias"]))
	x = x + conv1d_hf(h, w[p + "mlp.c_proj.weight"], w[p + "mlp.c_proj.bias"])
	return x, (k, v)


def gpt2_stack(w: W, layers: int, heads: int, emb: Tensor, past: Optional[List] = None):
	"""HF:models/gpt2/modeling_gpt2.py:514-634 GPT2Model.forward with inputs_embeds; the position table
	is nulled by the reference (unified_voice.py:77,425-426) so hidden = inputs_embeds.  Ends in ln_f."""
	x = emb
	new_past = []
	for i in range(layers):
		x, kv = gpt2_block(w, i, x, heads, None if past is None else past[i])
		new_past.append(kv)
	return layer_norm(x, w["gpt.ln_f.weight"], w["gpt.ln_f.bias"]), new_past


# --------------------------------------------------------------------------------------------
# UnifiedVoice (a1-a4, a7)
# --------------------------------------------------------------------------------------------


class AROracle:
	"""Functional UnifiedVoice over a reference-layout state dict."""

	def __init__(self, w: W, cfg):
		self.w, self.cfg = w, cfg

	# unified_voice.py:639-644 / :614-619
	def prefix_embeddings(self, cond_latent: Tensor, text: Tensor) -> Tensor:
		c = self.cfg
		t = F.pad(text, (0, 1), value=c.stop_text_token)
		t = F.pad(t, (1, 0), value=c.start_text_token)
		pos = self.w["text_pos_embedding.emb.weight"][: t.shape[1]]
		emb = self.w["text_embedding.weight"][t] + pos
		return torch.cat([cond_latent.unsqueeze(1), emb], dim=1)          # [b, P = Tt + 3, d]

	def lm_head(self, hidden: Tensor) -> Tensor:
		"""unified_voice.py:106,239  lm_head = Sequential(final_norm, mel_head)."""
		h = layer_norm(hidden, self.w["final_norm.weight"], self.w["final_norm.bias"])
		return F.linear(h, self.w["mel_head.weight"], self.w["mel_head.bias"])

	def prefill(self, prefix: Tensor, B: int, prompt: Optional[Tensor] = None):
		"""unified_voice.py:203-211: cat[cached prefix (repeat_interleave to B), mel_emb(start)+mel_pos(0)].
		prompt [B, n] (prompted continuation, :651-656): the ids behind the fake prefix are [start_mel | prompt], embedded at mel positions 0 .. n (:205-206)."""
		c = self.cfg
		if prefix.shape[0] != B:
			prefix = prefix.repeat_interleave(B // prefix.shape[0], 0)
		start = self.w["mel_embedding.weight"][c.start_mel_token] + self.w["mel_pos_embedding.emb.weight"][0]
		emb = torch.cat([prefix, start.view(1, 1, -1).expand(B, 1, -1)], dim=1)
		if prompt is not None and prompt.shape[1]:
			n = prompt.shape[1]
			emb = torch.cat([emb, self.w["mel_embedding.weight"][prompt] + self.w["mel_pos_embedding.emb.weight"][1:n + 1]], dim=1)
		hidden, past = gpt2_stack(self.w, c.layers, c.heads, emb)
		return self.lm_head(hidden), past, hidden

	def decode(self, tok: Tensor, k: int, past):
		"""unified_voice.py:212-214, KV-cached one-token step.  `k` (1-based) is the index of the generated
		token being fed back; the reference indexes the mel position table with
		attention_mask.shape[1] - mel_len = k + 1 (the quirk recorded in SURVEY.md section 0)."""
		emb = self.w["mel_embedding.weight"][tok] + self.w["mel_pos_embedding.emb.weight"][k + 1]
		hidden, past = gpt2_stack(self.w, self.cfg.layers, self.cfg.heads, emb.unsqueeze(1), past)
		return self.lm_head(hidden)[:, -1], past, hidden[:, -1]

	def teacher_forced_logits(self, cond_latent: Tensor, text: Tensor, toks: Tensor, steps: Sequence[int]) -> Tensor:
		"""Logits after `j` fed-back tokens for every j in `steps`, of a teacher-forced decode that feeds toks[:, 0], toks[:, 1], ...:
		ONE dense causal pass over [prefix | start_mel at mel position 0 | token k at mel position k + 1] -- the sequence the
		KV-cached loop of `prefill` + `decode` builds (unified_voice.py:203-214, position quirk included), so equal to its steps
		up to f32 summation order (tests/test_oracle_golden.py pins that).  Long contexts at full size cost one pass instead of
		hundreds of cached steps.  Returns [b, len(steps), V]."""
		w, c = self.w, self.cfg
		b, n = toks.shape
		prefix = self.prefix_embeddings(cond_latent, text).expand(b, -1, -1)
		P = prefix.shape[1]
		start = (w["mel_embedding.weight"][c.start_mel_token] + w["mel_pos_embedding.emb.weight"][0]).view(1, 1, -1).expand(b, 1, -1)
		last = max(steps)
		fed = w["mel_embedding.weight"][toks[:, :last]] + w["mel_pos_embedding.emb.weight"][2:last + 2]
		hidden, _ = gpt2_stack(w, c.layers, c.heads, torch.cat([prefix, start, fed], dim=1))
		return self.lm_head(hidden[:, [P + j for j in steps]])

	def forward_latents(self, cond_latent: Tensor, text: Tensor, codes: Tensor) -> Tensor:
		"""unified_voice.py:544-599 with return_latent=True, clip_inputs=False, text_first=True, and
		get_logits :508-522.  set_mel_padding (:494-506) is a no-op when wav_lengths =
		codes.shape[-1]*mel_length_compression, which is what inference.py:368 passes."""
		c, w = self.cfg, self.w
		t = F.pad(text, (0, 1), value=c.stop_text_token)
		m = F.pad(codes, (0, 1), value=c.stop_mel_token)
		t = F.pad(t, (1, 0), value=c.start_text_token)                  # build_aligned_inputs_and_targets :489-492
		text_emb = w["text_embedding.weight"][t] + w["text_pos_embedding.emb.weight"][: t.shape[1]]
		m = F.pad(m, (1, 0), value=c.start_mel_token)
		mel_emb = w["mel_embedding.weight"][m] + w["mel_pos_embedding.emb.weight"][: m.shape[1]]
		emb = torch.cat([cond_latent.unsqueeze(1), text_emb, mel_emb], dim=1)
		hidden, _ = gpt2_stack(w, c.layers, c.heads, emb)
		enc = layer_norm(hidden[:, 1:], w["final_norm.weight"], w["final_norm.bias"])
		return enc[:, -m.shape[1]:][:, :-2]


# --------------------------------------------------------------------------------------------
# generate() sample loop (a5) -- restated; no runnable reference on transformers 5.15
# --------------------------------------------------------------------------------------------


def warp_repetition_penalty(input_ids: Tensor, scores: Tensor, penalty: float) -> Tensor:
	"""HF:generation/logits_process.py RepetitionPenaltyLogitsProcessor.__call__ (2-D branch)."""
	score = torch.gather(scores, 1, input_ids)
	score = torch.where(score < 0, score * penalty, score / penalty)
	return scores.scatter(1, input_ids, score)


def warp_suppress(scores: Tensor, suppress: Sequence[int]) -> Tensor:
	"""HF SuppressTokensLogitsProcessor.__call__."""
	mask = torch.isin(torch.arange(scores.shape[-1], device=scores.device), torch.tensor(list(suppress), device=scores.device))
	return torch.where(mask, -float("inf"), scores)


def warp_temperature(scores: Tensor, temperature: float) -> Tensor:
	return scores / temperature


def warp_top_k(scores: Tensor, top_k: int, min_tokens_to_keep: int = 1) -> Tensor:
	k = min(max(top_k, min_tokens_to_keep), scores.size(-1))
	remove = scores < torch.topk(scores, k)[0][..., -1, None]
	return scores.masked_fill(remove, -float("inf"))


def warp_top_p(scores: Tensor, top_p: float, min_tokens_to_keep: int = 1) -> Tensor:
	sorted_logits, sorted_indices = torch.sort(scores, descending=False)
	cum = sorted_logits.softmax(dim=-1).cumsum(dim=-1)
	rm = cum <= (1 - top_p)
	rm[..., -min_tokens_to_keep:] = 0
	remove = rm.scatter(1, sorted_indices, rm)
	return scores.masked_fill(remove, -float("inf"))


def warp_typical(scores: Tensor, mass: float, min_tokens_to_keep: int = 1) -> Tensor:
	"""unified_voice.py:47-75 TypicalLogitsWarper.__call__ (the reference's own class; `inference_speech(typical_sampling=True)` hands it to generate() as
	a custom logits_processor, :657): tokens ordered by |(-log p) - H|, the closest ones kept until their mass reaches `mass`.  Pinned by
	tests/golden/stress_ar.npz, whose `*_typical` cases ran that class inside the reference's sample_stream."""
	logp = F.log_softmax(scores, dim=-1)
	p = torch.exp(logp)
	ent = -(logp * p).nansum(-1, keepdim=True)
	dist = torch.abs((-logp) - ent)
	sorted_dist, order = torch.sort(dist, descending=False)
	cum = scores.gather(-1, order).softmax(dim=-1).cumsum(dim=-1)
	last = (cum < mass).sum(dim=1)
	last[last < 0] = 0
	remove_sorted = sorted_dist > sorted_dist.gather(1, last.view(-1, 1))
	if min_tokens_to_keep > 1:
		remove_sorted[..., :min_tokens_to_keep] = 0
	remove = remove_sorted.scatter(1, order, remove_sorted)
	return scores.masked_fill(remove, -float("inf"))


def process_logits(input_ids: Tensor, logits: Tensor, *, temperature=1.0, top_k=0, top_p=1.0,
					repetition_penalty=1.0, suppress_tokens=None, typical_mass=None) -> Tensor:
	"""Processor order of the reference's sample branch: HF `_get_logits_processor` (repetition penalty,
	... suppress_tokens, then the caller's `logits_processor` -- the typical warper, unified_voice.py:657 -- which HF's
	`_merge_criteria_processor_list` appends behind the defaults) then `_get_logits_warper` (stream_generator.py:80-85: Temperature -> TopK -> TopP)."""
	s = logits
	if repetition_penalty is not None and repetition_penalty != 1.0:
		s = warp_repetition_penalty(input_ids, s, repetition_penalty)
	if suppress_tokens:
		s = warp_suppress(s, suppress_tokens)
	if typical_mass is not None:
		s = warp_typical(s, typical_mass)
	if temperature is not None and temperature != 1.0:
		s = warp_temperature(s, temperature)
	if top_k is not None and top_k != 0:
		s = warp_top_k(s, top_k)
	if top_p is not None and top_p < 1.0:
		s = warp_top_p(s, top_p)
	return s


def inference_speech(ar: AROracle, cond_latent: Tensor, text: Tensor, *, num_return_sequences=1,
					max_generate_length=None, temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.0,
					suppress_tokens=None, sample_device="cpu", seed=0, return_logits=False, forced_tokens=None, input_tokens=None,
					typical_sampling=False, typical_mass=0.9):
	"""unified_voice.py:632-668 + stream_generator.py:213-639 (sample branch) + HF `_sample`
	HF:generation/utils.py:2894-2937.

	* `input_tokens` [R, n] (prompted continuation, :651-656; restated from source, `TTS.inference` never passes it): the reference tiles the fake prefix and the
	  prompts to `num_return_sequences` rows (:653-655, needs num_return_sequences % R == 0) and THEN hands generate() `num_return_sequences` again, which
	  expands every row that many times (HF `_expand_inputs_for_generation`: repeat_interleave) -- num_return_sequences ** 2 sequences come back, row
	  i * nrs + j = the j-th sample of prompt row i % R.  The prompt tokens are part of the returned ids and count towards max_generate_length (:660, :668).

	* RNG: `setup_seed(seed)` with seed=0 is unconditional (stream_generator.py:223,296).
	* Keyword defaults are HF `GenerationConfig`'s, which is what an omitted `**hf_generate_kwargs` entry means in the reference
	  (stream_generator.py:262-276): temperature 1, **top_k 50**, top_p 1, repetition_penalty 1.  `TTS.inference` always passes
	  top_k explicitly (0 by default, inference.py:157,334-346).  Pinned against the installed HF loop: oracle/stub_lm.py.
	* fake prefix ids are 1, last = start_mel (unified_voice.py:647-649); the repetition penalty sees them.
	* max_length = trunc_index + max_generate_length (:660); loop stops when every row hit EOS or at max_length.
	* finished rows emit pad (= stop_mel_token).
	* `sample_device`: the device torch.multinomial runs on.  CPU (mt19937) and GPU (Philox) streams differ,
	  so bit-exact ids are defined per device type: the GPU parity test runs this oracle with
	  sample_device="cuda" (fp32 logits computed here on the CPU, moved for sampling only).
	* `forced_tokens` [B, n]: teacher forcing (ids fed back instead of the sampled ones), for logits parity.
	"""
	c = ar.cfg
	B = num_return_sequences * text.shape[0]
	prompt = None
	if input_tokens is not None:
		assert text.shape[0] == 1 and num_return_sequences % input_tokens.shape[0] == 0, "The number of return sequences must be divisible by the number of input sequences"
		prompt = input_tokens.repeat(num_return_sequences // input_tokens.shape[0], 1).repeat_interleave(num_return_sequences, 0)
		B = prompt.shape[0]
	prefix = ar.prefix_embeddings(cond_latent, text)
	P = prefix.shape[1]
	trunc_index = P + 1
	max_len = trunc_index + (c.max_mel_tokens - 1 if max_generate_length is None else max_generate_length)
	torch.manual_seed(seed)
	if sample_device != "cpu":
		torch.cuda.manual_seed_all(seed)
	input_ids = torch.ones((B, trunc_index), dtype=torch.long)
	input_ids[:, -1] = c.start_mel_token
	if prompt is not None:
		input_ids = torch.cat([input_ids, prompt], dim=1)
	unfinished = torch.ones(B, dtype=torch.long)
	logits, past, _ = ar.prefill(prefix, B) if prompt is None else ar.prefill(prefix, B, prompt)
	logits = logits[:, -1]
	all_logits = []
	k = 0 if prompt is None else prompt.shape[1]      # mel tokens behind start_mel so far: the fed-back token k sits at mel position k + 1
	n_drawn = 0
	while True:
		logits = logits.float()
		if return_logits:
			all_logits.append(logits.clone())
		scores = process_logits(input_ids, logits, temperature=temperature, top_k=top_k, top_p=top_p,
								repetition_penalty=repetition_penalty, suppress_tokens=suppress_tokens, typical_mass=typical_mass if typical_sampling else None)
		probs = F.softmax(scores.to(sample_device), dim=-1)
		nxt = torch.multinomial(probs, num_samples=1).squeeze(1).cpu()
		if forced_tokens is not None and n_drawn < forced_tokens.shape[1]:
			nxt = forced_tokens[:, n_drawn]
		n_drawn += 1
		nxt = nxt * unfinished + c.stop_mel_token * (1 - unfinished)
		input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
		k += 1
		done = (nxt == c.stop_mel_token) | (input_ids.shape[1] >= max_len)
		unfinished = unfinished & ~done
		if unfinished.max() == 0:
			break
		logits, past, _ = ar.decode(nxt, k, past)
	gen = input_ids[:, trunc_index:]
	if return_logits:
		return gen, torch.stack(all_logits, dim=1)
	return gen


def sample_stream(ar: AROracle, cond_latent: Tensor, text: Tensor, *, num_return_sequences=1, max_generate_length=None,
				  temperature=1.0, top_k=50, top_p=1.0, repetition_penalty=1.0, suppress_tokens=None, sample_device="cpu", seed=0, prompt=None,
				  typical_mass=None, return_logits=False):
	"""a6: `NewGenerationMixin.sample_stream` (stream_generator.py:911-1190) as `get_generator` drives it (unified_voice.py:670-679),
	a generator of (next_tokens [B], latent [B, d]).  Pinned by tests/golden/sample_stream.npz, which the REFERENCE's own loop produced
	(oracle/make_golden.py: sample_stream_case).  What that pin fixes:
	  * the latent yielded with token k is final_norm(hidden[-1][:, -1]) of the forward whose logits token k was SAMPLED FROM
	    (:1172 reads `outputs` of the same iteration): the prefill's last row for the first token, the step that consumed token k-1
	    afterwards -- not the step that consumes token k;
	  * every sampled token is yielded, the last one included (the yield sits in front of the stopping test :1186);
	  * finished rows yield the pad token (:1165-1171) and the loop ends right after the token with which the last row finishes,
	    or when input_ids reaches max_length.
	Logits are not upcast before the processors here (the model is f32 already)."""
	c = ar.cfg
	B = num_return_sequences * text.shape[0]
	prefix = ar.prefix_embeddings(cond_latent, text)
	trunc_index = prefix.shape[1] + 1
	max_len = trunc_index + (c.max_mel_tokens - 1 if max_generate_length is None else max_generate_length)
	torch.manual_seed(seed)
	if sample_device != "cpu":
		torch.cuda.manual_seed_all(seed)
	input_ids = torch.ones((B, trunc_index), dtype=torch.long)
	input_ids[:, -1] = c.start_mel_token
	if prompt is not None:      # `inputs` with mel tokens behind the fake prefix (one row per sequence): a prompted continuation, pinned by the fixture's "prompted" case
		input_ids = torch.cat([input_ids, prompt], dim=1)
	unfinished = torch.ones(B, dtype=torch.long)
	logits, past, hidden = ar.prefill(prefix, B, prompt)
	logits, hidden = logits[:, -1], hidden[:, -1]
	k = 0 if prompt is None else prompt.shape[1]
	while True:
		scores = process_logits(input_ids, logits, temperature=temperature, top_k=top_k, top_p=top_p,
								repetition_penalty=repetition_penalty, suppress_tokens=suppress_tokens, typical_mass=typical_mass)
		probs = F.softmax(scores.to(sample_device), dim=-1)
		nxt = torch.multinomial(probs, num_samples=1).squeeze(1).cpu()
		nxt = nxt * unfinished + c.stop_mel_token * (1 - unfinished)
		lat = layer_norm(hidden, ar.w["final_norm.weight"], ar.w["final_norm.bias"])
		yield (nxt, lat, logits) if return_logits else (nxt, lat)
		input_ids = torch.cat([input_ids, nxt[:, None]], dim=-1)
		k += 1
		unfinished = unfinished * (nxt != c.stop_mel_token).long()
		if unfinished.max() == 0 or input_ids.shape[1] >= max_len:
			return
		logits, past, hidden = ar.decode(nxt, k, past)


def fix_stop_tokens(codes: Tensor, stop_mel_token: int) -> Tensor:
	"""inference.py:353-366.  The reference calls `.min()` on a possibly empty index set before the
	emptiness check (:355 vs :357) and would raise; rows without a stop token are left untouched here,
	which is what the check intends.  Note `stm - 3 < len` is always true when a stop exists."""
	codes = codes.clone()
	for i in range(codes.shape[0]):
		idx = (codes[i] == stop_mel_token).nonzero()
		if len(idx) == 0:
			continue
		stm = int(idx.min())
		codes[i][idx] = 83
		codes[i][stm:] = 83
		if stm - 3 < codes[i].shape[0]:
			codes[i][-3] = 45
			codes[i][-2] = 45
			codes[i][-1] = 248
	return codes


def trim_calm_tokens(codes: Tensor, latents: Tensor, calm_token: int = 83) -> Tensor:
	"""inference.py:381-389: cut latents after more than 8 consecutive calm tokens in row 0."""
	calm = 0
	for k in range(codes.shape[-1]):
		calm = calm + 1 if int(codes[0, k]) == calm_token else 0
		if calm > 8:
			return latents[:, :k]
	return latents


# --------------------------------------------------------------------------------------------
# DiffusionTTS network (a9-a13)
# --------------------------------------------------------------------------------------------


def group_norm32(x: Tensor, w: Tensor, b: Tensor, groups: int = 32) -> Tensor:
	"""arch_utils.py:24-44: GroupNorm computed in float; 32 groups for channels > 64."""
	return F.group_norm(x.float(), groups, w, b, 1e-5).type(x.dtype)


def rel_pos_bucket(rel: Tensor, num_buckets: int = 32, max_distance: int = 64) -> Tensor:
	"""xtransformers.py:157-177 with causal=False (arch_utils.py:174)."""
	n = -rel
	nb = num_buckets // 2
	ret = (n < 0).long() * nb
	n = torch.abs(n)
	max_exact = nb // 2
	is_small = n < max_exact
	large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
	large = torch.min(large, torch.full_like(large, nb - 1))
	return ret + torch.where(is_small, n, large)


def rel_pos_bias(table: Tensor, i: int, j: int, scale: float) -> Tensor:
	"""xtransformers.py:179-188: bias[h, q, k] = table[bucket(k - q), h] * scale."""
	q = torch.arange(i)
	k = torch.arange(j)
	bucket = rel_pos_bucket(k[None, :] - q[:, None])
	return table[bucket].permute(2, 0, 1) * scale


# Operand rounding of the build's fp8 mode (include/ttk.h TTK_FP8), for tests that pin "fp8 mode == the same arithmetic on operands rounded to
# fp8-e4m3": when set, the activation entering each ResBlock convolution and each AttentionBlock's proj_out passes through it (the qkv projection keeps 16-bit operands).  None = the reference's arithmetic.
BLOCK_OPERAND_ROUNDING = None


def fp8_e4m3_round(x: Tensor) -> Tensor:
	"""OCP e4m3 (torch.float8_e4m3fn), round to nearest even, saturating at +-448 like v_cvt_pk_fp8_f32."""
	return x.clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(x.dtype)


def _q(x: Tensor) -> Tensor:
	return x if BLOCK_OPERAND_ROUNDING is None else BLOCK_OPERAND_ROUNDING(x)


def attention_block(w: W, p: str, x: Tensor, heads: int) -> Tensor:
	"""arch_utils.py:136-190 AttentionBlock._forward + :59-94 QKVAttentionLegacy (head-major [H,3,ch]
	channel split, q*s and k*s with s = ch^-1/4, softmax in float)."""
	b, c, T = x.shape
	qkv = F.conv1d(group_norm32(x, w[p + "norm.weight"], w[p + "norm.bias"]), w[p + "qkv.weight"], w[p + "qkv.bias"])      # (never rounded: the build's fp8 modes keep this projection 16-bit)
	ch = c // heads
	q, k, v = qkv.reshape(b * heads, ch * 3, T).split(ch, dim=1)
	s = 1 / math.sqrt(math.sqrt(ch))
	weight = torch.einsum("bct,bcs->bts", q * s, k * s)
	bias = rel_pos_bias(w[p + "relative_pos_embeddings.relative_attention_bias.weight"], T, T, ch ** 0.5)
	weight = (weight.reshape(b, heads, T, T) + bias).reshape(b * heads, T, T)
	weight = torch.softmax(weight.float(), dim=-1)
	a = torch.einsum("bts,bcs->bct", weight, v).reshape(b, -1, T)
	return x + F.conv1d(_q(a), w[p + "proj_out.weight"], w[p + "proj_out.bias"])


def res_block(w: W, p: str, x: Tensor, emb: Tensor) -> Tensor:
	"""diffusion.py:1316-1376 ResBlock(use_scale_shift_norm=True, efficient_config=True, kernel 3)."""
	h = F.conv1d(_q(F.silu(group_norm32(x, w[p + "in_layers.0.weight"], w[p + "in_layers.0.bias"]))),
				w[p + "in_layers.2.weight"], w[p + "in_layers.2.bias"])
	e = F.linear(F.silu(emb), w[p + "emb_layers.1.weight"], w[p + "emb_layers.1.bias"])[..., None]
	scale, shift = torch.chunk(e, 2, dim=1)
	h = group_norm32(h, w[p + "out_layers.0.weight"], w[p + "out_layers.0.bias"]) * (1 + scale) + shift
	h = F.conv1d(_q(F.silu(h)), w[p + "out_layers.3.weight"], w[p + "out_layers.3.bias"], padding=1)
	return x + h


def timestep_embedding(t: Tensor, dim: int, max_period: int = 10000) -> Tensor:
	"""diffusion.py:1277-1295."""
	half = dim // 2
	freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
	args = t[:, None].float() * freqs[None]
	return torch.cat([torch.cos(args), torch.sin(args)], dim=-1)


class DiffusionOracle:
	"""Functional DiffusionTTS over a reference-layout state dict."""

	def __init__(self, w: W, cfg):
		self.w, self.cfg = w, cfg

	def timestep_independent(self, latents: Tensor, cond: Tensor, T: int) -> Tensor:
		"""diffusion.py:1487-1510 (latent branch; eval mode so no unconditioned masking)."""
		w, heads = self.w, self.cfg.num_heads
		x = latents.permute(0, 2, 1)
		scale, shift = torch.chunk(cond, 2, dim=1)
		h = F.conv1d(x, w["latent_conditioner.0.weight"], w["latent_conditioner.0.bias"], padding=1)
		for i in range(1, 5):
			h = attention_block(w, f"latent_conditioner.{i}.", h, heads)
		h = group_norm32(h, w["code_norm.weight"], w["code_norm.bias"]) * (1 + scale.unsqueeze(-1)) + shift.unsqueeze(-1)
		return F.interpolate(h, size=T, mode="nearest")

	def time_embed(self, t: Tensor) -> Tensor:
		"""diffusion.py:1417-1421,1549."""
		w = self.w
		e = timestep_embedding(t, self.cfg.model_channels)
		e = F.linear(e, w["time_embed.0.weight"], w["time_embed.0.bias"])
		return F.linear(F.silu(e), w["time_embed.2.weight"], w["time_embed.2.bias"])

	def forward(self, x: Tensor, t: Tensor, E: Optional[Tensor], conditioning_free: bool = False) -> Tensor:
		"""diffusion.py:1517-1574 with precomputed_aligned_embeddings (the DDP `extraneous_addition*0` is a no-op)."""
		w, c = self.w, self.cfg
		heads = c.num_heads
		if conditioning_free:
			code = w["unconditioned_embedding"].repeat(x.shape[0], 1, x.shape[-1])
		else:
			code = E
		temb = self.time_embed(t)
		for i in range(3):
			p = f"conditioning_timestep_integrator.{i}."
			code = attention_block(w, p + "attn.", res_block(w, p + "resblk.", code, temb), heads)
		h = F.conv1d(x, w["inp_block.weight"], w["inp_block.bias"], padding=1)
		h = F.conv1d(torch.cat([h, code], dim=1), w["integrating_conv.weight"], w["integrating_conv.bias"])
		for i in range(c.num_layers):
			p = f"layers.{i}."
			h = attention_block(w, p + "attn.", res_block(w, p + "resblk.", h, temb), heads)
		for i in range(c.num_layers, c.num_layers + 3):
			h = res_block(w, f"layers.{i}.", h, temb)
		h = F.silu(group_norm32(h.float(), w["out.0.weight"], w["out.0.bias"]))
		return F.conv1d(h, w["out.2.weight"], w["out.2.bias"], padding=1)


# --------------------------------------------------------------------------------------------
# Diffusion schedule + samplers (a14-a16), numpy float64 tables exactly as the reference
# --------------------------------------------------------------------------------------------


def linear_betas(n: int) -> np.ndarray:
	"""diffusion.py:107-124 get_named_beta_schedule('linear', n)."""
	scale = 1000 / n
	return np.linspace(scale * 0.0001, scale * 0.02, n, dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts: Sequence[int]) -> List[int]:
	"""diffusion.py:1169-1222 (integer-list form).  `cur_idx` accumulates in Python float and `round()` is
	banker's rounding -- both kept."""
	size_per = num_timesteps // len(section_counts)
	extra = num_timesteps % len(section_counts)
	start_idx = 0
	all_steps: List[int] = []
	for i, count in enumerate(section_counts):
		size = size_per + (1 if i < extra else 0)
		if size < count:
			raise ValueError(f"cannot divide section of {size} steps into {count}")
		frac_stride = 1 if count <= 1 else (size - 1) / (count - 1)
		cur = 0.0
		for _ in range(count):
			all_steps.append(start_idx + round(cur))
			cur += frac_stride
		start_idx += size
	return sorted(set(all_steps))


class SpacedSchedule:
	"""diffusion.py:1110-1133 SpacedDiffusion.__init__ over :205-262 GaussianDiffusion.__init__, as built by
	get_diffuser(steps, cond_free, cond_free_k=2, trained_diffusion_steps=4000) :1576-1590."""

	def __init__(self, steps: int = 80, cond_free: bool = True, cond_free_k: float = 2, trained_steps: int = 4000):
		base = np.cumprod(1.0 - linear_betas(trained_steps), axis=0)
		use = set(space_timesteps(trained_steps, [steps]))
		last = 1.0
		betas, self.timestep_map = [], []
		for i, ac in enumerate(base):
			if i in use:
				betas.append(1 - ac / last)
				last = ac
				self.timestep_map.append(i)
		betas = np.array(betas, dtype=np.float64)
		self.betas = betas
		self.num_timesteps = len(betas)
		self.conditioning_free, self.conditioning_free_k = cond_free, cond_free_k
		alphas = 1.0 - betas
		self.alphas_cumprod = np.cumprod(alphas, axis=0)
		self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
		self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
		self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
		self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
		self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
		self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
		self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

	@staticmethod
	def _f(arr, i):
		"""diffusion.py:1254-1267 _extract_into_tensor: float64 table entry -> float32 scalar."""
		return torch.tensor(arr[i]).float()

	def p_mean_variance(self, model: DiffusionOracle, x: Tensor, i: int, E: Tensor):
		"""diffusion.py:325-431 (epsilon model, learned_range variance, clip_denoised) via the
		_WrappedModel index map :1225-1237."""
		b, C = x.shape[:2]
		t = torch.tensor([self.timestep_map[i]] * b)
		out = model.forward(x, t, E)
		eps, var_values = torch.split(out, C, dim=1)
		if self.conditioning_free:
			out_u = model.forward(x, t, E, conditioning_free=True)
			eps_u, _ = torch.split(out_u, C, dim=1)
		min_log = self._f(self.posterior_log_variance_clipped, i)
		max_log = self._f(np.log(self.betas), i)
		frac = (var_values + 1) / 2
		log_var = frac * max_log + (1 - frac) * min_log
		if self.conditioning_free:
			cfk = self.conditioning_free_k * (1 - i / self.num_timesteps)       # :391-393 (ramp)
			eps = (1 + cfk) * eps - cfk * eps_u
		x0 = (self._f(self.sqrt_recip_alphas_cumprod, i) * x - self._f(self.sqrt_recipm1_alphas_cumprod, i) * eps).clamp(-1, 1)
		mean = self._f(self.posterior_mean_coef1, i) * x0 + self._f(self.posterior_mean_coef2, i) * x
		return mean, log_var, x0

	def ddim_step(self, model, x, i, E):
		"""diffusion.py:646-694 with eta = 0 (sigma = 0; the unused randn_like is still drawn :685)."""
		_, _, x0 = self.p_mean_variance(model, x, i, E)
		eps = (self._f(self.sqrt_recip_alphas_cumprod, i) * x - x0) / self._f(self.sqrt_recipm1_alphas_cumprod, i)
		ab_prev = self._f(self.alphas_cumprod_prev, i)
		torch.randn_like(x)
		return x0 * torch.sqrt(ab_prev) + torch.sqrt(1 - ab_prev) * eps

	def p_step(self, model, x, i, E):
		"""diffusion.py:510-554 ancestral step."""
		mean, log_var, _ = self.p_mean_variance(model, x, i, E)
		noise = torch.randn_like(x)
		return mean + (0.0 if i == 0 else 1.0) * torch.exp(0.5 * log_var) * noise

	def sample_loop(self, model: DiffusionOracle, noise: Tensor, E: Tensor, sampler: str = "ddim") -> Tensor:
		"""diffusion.py:500-508, :734-810, :556-644."""
		x = noise
		for i in reversed(range(self.num_timesteps)):
			x = self.ddim_step(model, x, i, E) if sampler == "ddim" else self.p_step(model, x, i, E)
		return x


TACOTRON_MEL_MAX = 2.3143386840820312
TACOTRON_MEL_MIN = -11.512925148010254


def denormalize_tacotron_mel(m: Tensor) -> Tensor:
	"""arch_utils.py:532-537."""
	return ((m + 1) / 2) * (TACOTRON_MEL_MAX - TACOTRON_MEL_MIN) + TACOTRON_MEL_MIN


def mel_frames_for(M: int) -> int:
	"""inference.py:400."""
	return M * 4 * 24000 // 22050
