"""Drop-in for the reference's autoregressive module object: same method names, arguments and return values as
`UnifiedVoice` (/root/reference/tortoise_tts/models/unified_voice.py:334-679) for the calls `TTS.inference` makes
(tortoise_tts/inference.py:266-285, 334-346, 371-379), backed by libttk (HIP, gfx950).  No torch fallback exists:
every forward goes through the C ABI or raises.

What runs where
  * GPT-2 stack, embeddings, final norms, mel head, KV cache ........ libttk  (csrc/ar.hip and kernels)
  * logits processors / warpers, softmax, multinomial(1) given its noise, stop/pad bookkeeping, the next step's input
    embedding ......................................................... libttk  (csrc/sample.hip, one launch per token)
  * the Exp(1) noise of torch.multinomial ............................ libttk, inside the mel-head launch: torch's own Philox
    stream restated (csrc/ttk_rng.h), used only after a bitwise comparison with `exponential_` on this device and shape
    (else torch `exponential_` on the same stream) -- the generator stream is part of the reference's observable behaviour
    (seed 0 on every call) and is left where torch's draws would have left it; typical sampling (rare, off by default)
    runs inside the kernel too since round 3 (sampling.py keeps the torch-op forms for `hf_exact_top_p`).
  * the per-token loop .............................................. here; one HIP-graph replay per token.
"""
from __future__ import annotations

import dataclasses
import os
import warnings
from typing import Dict, Iterator, Optional, Tuple

import torch

from . import _lib
from .sampling import LogitsPipeline, multinomial1, setup_seed
from .weights import ARConfig, ar_shapes


class UnifiedVoice:
	def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: ARConfig = ARConfig(), dtype: str = "bf16",
				 device: str = "cuda:0", max_batch: int = 16, max_ctx: Optional[int] = None, use_graph: bool = True,
				 hf_exact_top_p: bool = False):
		"""hf_exact_top_p: run HF's TopPLogitsWarper as torch ops in front of the sampling kernel when top_p < 1 (its f32 cumsum rounding, bit for
		bit) instead of the kernel's exact fixed-point cut -- the two differ only on rows whose cumulative mass ties with 1 - top_p within f32
		rounding (csrc/sample.hip; tests/test_gpu_parity.py::test_top_p_boundary_stress); costs the per-token torch launches the kernel removed."""
		self.hf_exact_top_p = hf_exact_top_p
		self.cfg = cfg
		self.device = torch.device(device)
		if self.device.type != "cuda":
			raise _lib.TTKError("tortoise_tts_amd runs on an MI355X only (device must be cuda:N)")
		self.lib = _lib.load()
		self.dtype = _lib.DTYPES[dtype]
		if self.dtype == _lib.TTK_FP8:      # the decode GEMVs have 16 rows: fp8 activations buy nothing there; 'fp8' means fp8 weights
			self.dtype = _lib.TTK_FP8W
		self.max_batch = max_batch
		self.max_ctx = max_ctx or (cfg.max_text_seq_len + 2 + cfg.max_mel_seq_len)
		self.use_graph = use_graph
		# attributes TTS.inference reads (inference.py:354,368)
		self.stop_mel_token, self.start_mel_token = cfg.stop_mel_token, cfg.start_mel_token
		self.mel_length_compression = cfg.mel_length_compression
		self.max_mel_tokens, self.max_text_tokens = cfg.max_mel_tokens, cfg.max_text_tokens
		self.model_dim, self.layers, self.heads = cfg.model_dim, cfg.layers, cfg.heads
		self.number_mel_codes = cfg.number_mel_codes

		names = list(ar_shapes(cfg).keys())
		missing = [n for n in names if n not in state_dict]
		if missing:
			raise _lib.TTKError(f"state_dict lacks {len(missing)} hot-path tensors, e.g. {missing[:3]}")
		views, keep = _lib.weight_views(state_dict, names)
		c = _lib.ARConfigC(cfg.layers, cfg.model_dim, cfg.heads, cfg.max_mel_seq_len, cfg.max_text_seq_len,
						   cfg.number_text_tokens + 1, cfg.number_mel_codes, cfg.start_text_token, cfg.stop_text_token,
						   cfg.start_mel_token, cfg.stop_mel_token, self.dtype, max_batch, self.max_ctx)
		self._h = _lib.C.c_void_p()
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_ar_create(_lib.C.byref(self._h), _lib.C.byref(c), views, len(names)), "ttk_ar_create")
		del keep
		self._states: Dict[tuple, "_GenState"] = {}
		self._prefix = None
		self._streaming = False          # a streamed generation is open on this handle (its KV cache, noise and latent ring belong to it)

	def __del__(self):
		h = getattr(self, "_h", None)
		if h:
			self.lib.ttk_ar_destroy(h)
			self._h = None

	def parameters(self):
		yield torch.empty(0, device=self.device)

	def to(self, *a, **k):
		return self

	def eval(self):
		return self

	# ------------------------------------------------------------------ C-ABI calls
	def _check_ids(self, ids: torch.Tensor, n: int, what: str):
		"""token ids index embedding rows on the device: an id outside the table must be the IndexError nn.Embedding raises in the
		reference, not an out-of-bounds read"""
		if ids.numel() and (int(ids.min()) < 0 or int(ids.max()) >= n):
			raise IndexError(f"{what} ids must lie in [0, {n}); got [{int(ids.min())}, {int(ids.max())}]")

	def _prefill(self, cond: torch.Tensor, text: torch.Tensor, B: int, prompt: Optional[torch.Tensor] = None) -> torch.Tensor:
		"""prompt [1 or B, n] int64: the mel tokens of a prompted continuation, cached behind start_mel (include/ttk.h: ttk_ar_prefill_prompted)"""
		cond = cond.to(self.device, torch.float32).contiguous()
		text = text.to(self.device, torch.int64).contiguous().view(-1)
		_lib.require_cuda(cond, text)
		self._check_ids(text, self.cfg.number_text_tokens + 1, "text token")
		logits = torch.empty((B, self.cfg.number_mel_codes), device=self.device, dtype=torch.float32)
		if prompt is not None and prompt.shape[1]:
			prompt = prompt.to(self.device, torch.int64).contiguous()
			self._check_ids(prompt, self.cfg.number_mel_codes, "prompt mel code")
			_lib.check(self.lib.ttk_ar_prefill_prompted(self._h, cond.data_ptr(), cond.shape[0], text.data_ptr(), text.numel(), prompt.data_ptr(), prompt.shape[0],
														prompt.shape[1], B, logits.data_ptr(), _lib.stream_ptr()), "ttk_ar_prefill_prompted")
			return logits
		_lib.check(self.lib.ttk_ar_prefill(self._h, cond.data_ptr(), cond.shape[0], text.data_ptr(), text.numel(), B,
										   logits.data_ptr(), _lib.stream_ptr()), "ttk_ar_prefill")
		return logits

	def _prefill_lines(self, cond: torch.Tensor, texts, rows_per_line: int) -> torch.Tensor:
		"""several text lines as one decode batch (include/ttk.h: ttk_ar_prefill_lines): line g -> candidates [g * rows, (g + 1) * rows)"""
		G = len(texts)
		cond = cond.to(self.device, torch.float32)
		cond = (cond.expand(G, -1) if cond.shape[0] == 1 else cond).contiguous()
		if cond.shape[0] != G:
			raise ValueError("one conditioning latent, or one per line")
		flat = torch.cat([t.to(self.device, torch.int64).reshape(-1) for t in texts]).contiguous()
		_lib.require_cuda(cond, flat)
		self._check_ids(flat, self.cfg.number_text_tokens + 1, "text token")
		lens = (_lib.C.c_int * G)(*[int(t.numel()) for t in texts])
		logits = torch.empty((G * rows_per_line, self.cfg.number_mel_codes), device=self.device, dtype=torch.float32)
		_lib.check(self.lib.ttk_ar_prefill_lines(self._h, cond.data_ptr(), flat.data_ptr(), lens, G, rows_per_line, logits.data_ptr(),
												 _lib.stream_ptr()), "ttk_ar_prefill_lines")
		return logits

	def _decode(self, tok: torch.Tensor, logits: torch.Tensor, hidden: Optional[torch.Tensor] = None):
		"""one KV-cached step fed with `tok` (callers pass ids this module sampled, or teacher-forced ones they validated)"""
		_lib.check(self.lib.ttk_ar_decode(self._h, tok.data_ptr(), logits.data_ptr(), _lib.ptr(hidden), _lib.stream_ptr()),
				   "ttk_ar_decode")

	def _decode_next(self, logits: torch.Tensor, hidden: Optional[torch.Tensor] = None):
		"""the same step started from the input row the fused sampling launch (ttk_ar_sample_next) left in the handle"""
		_lib.check(self.lib.ttk_ar_decode_next(self._h, logits.data_ptr(), _lib.ptr(hidden), _lib.stream_ptr()), "ttk_ar_decode_next")

	def _check_health(self):
		"""after a generation: did the decode step's folded LayerNorm meet rows it cannot represent well (include/ttk.h: ttk_ar_health)?  The
		reference has no such failure mode -- it normalises in f32 before the matmul -- so the deviation is reported where it can occur."""
		flags = _lib.C.c_int(0)
		_lib.check(self.lib.ttk_ar_health(self._h, _lib.C.byref(flags), _lib.stream_ptr()), "ttk_ar_health")
		self.last_health = flags.value
		if flags.value & 1:
			warnings.warn("tortoise_tts_amd: a row of the GPT-2 residual stream had |mean| > 8 std during decoding; the folded-LayerNorm launches lose "
						  "precision on such rows (set TTK_AR_LNFOLD=0 before building the model for the form that normalises in f32 first)", RuntimeWarning)
		if flags.value & 2:
			warnings.warn("tortoise_tts_amd: non-finite LayerNorm statistics during decoding (an fp16 operand above 65504?); "
						  "use dtype='bf16' or TTK_AR_LNFOLD=0", RuntimeWarning)
		return flags.value

	# ------------------------------------------------------------------ reference surface
	def forward(self, speech_conditioning_latent, text_inputs, text_lengths, mel_codes, wav_lengths, types=None,
				text_first=True, raw_mels=None, return_attentions=False, return_latent=False, clip_inputs=True):
		"""unified_voice.py:544-599 in the one mode inference uses: return_latent=True, clip_inputs=False
		(inference.py:371-379).  Returns f32 [B, M, model_dim]."""
		if not return_latent or clip_inputs or types is not None or raw_mels is not None or not text_first or return_attentions:
			raise NotImplementedError("only forward(..., return_latent=True, clip_inputs=False) is on the inference hot path")
		B, M = mel_codes.shape
		cond = speech_conditioning_latent.to(self.device, torch.float32)
		if cond.shape[0] != B:
			cond = cond.expand(B, -1)
		cond = cond.contiguous()
		text = text_inputs.to(self.device, torch.int64)
		if text.shape[0] != B:
			text = text.expand(B, -1)
		text = text.contiguous()
		codes = mel_codes.to(self.device, torch.int64).contiguous()
		self._check_ids(text, self.cfg.number_text_tokens + 1, "text token")
		self._check_ids(codes, self.cfg.number_mel_codes, "mel code")
		# set_mel_padding (:494-506) rewrites codes past wav_lengths // compression + 1 with the stop token
		mel_lengths = torch.div(torch.as_tensor(wav_lengths).view(-1).to("cpu"), self.mel_length_compression, rounding_mode="trunc")
		if mel_lengths.numel() not in (1, B):
			raise ValueError("wav_lengths must have 1 or B entries")
		if int(mel_lengths.min()) + 1 < M:
			codes = codes.clone()
			for b in range(B):
				end = int(mel_lengths[b if mel_lengths.numel() == B else 0]) + 1
				if end < M:
					codes[b, end:] = self.stop_mel_token
		out = torch.empty((B, M, self.cfg.model_dim), device=self.device, dtype=torch.float32)
		_lib.check(self.lib.ttk_ar_latents(self._h, cond.data_ptr(), text.data_ptr(), text.shape[1], codes.data_ptr(), M, B,
										   out.data_ptr(), _lib.stream_ptr()), "ttk_ar_latents")
		return out

	__call__ = forward

	def inference_speech(self, speech_conditioning_latent, text_inputs, input_tokens=None, num_return_sequences=1,
						 max_generate_length=None, typical_sampling=False, typical_mass=.9, kv_cache=True, candidate_shard=None,
						 **hf_generate_kwargs):
		"""unified_voice.py:632-668 + the sample branch of `generate` (stream_generator.py:213-639, HF `_sample`).
		Returns int64 [B, L <= max_generate_length], rows padded with stop_mel_token after their EOS.

		candidate_shard=(lo, hi) (no reference counterpart; tortoise_tts_amd/dist.py): sample only candidates lo..hi-1 of the
		`num_return_sequences`, as one rank of a candidate-sharded run.  RNG contract: the rank draws the multinomial noise of ALL
		candidates per token (the same Philox stream on every rank: `generate` reseeds to 0) and consumes its own rows, so the ids it
		returns are bit for bit the rows lo..hi-1 of the unsharded call -- up to the length, which here ends with the shard's own last
		row (the gather pads with the stop token, as the unsharded loop does for finished rows)."""
		if text_inputs.shape[0] != 1:
			raise NotImplementedError("one text line per call, as inference.py:244-246 does")
		prompt = None
		if input_tokens is not None:
			# Prompted continuation (unified_voice.py:651-656; `TTS.inference` never passes it).  The reference tiles the fake prefix and the prompts to
			# num_return_sequences rows (:653-655) and then hands generate() num_return_sequences AGAIN, which expands every row that many times
			# (HF _expand_inputs_for_generation: repeat_interleave): num_return_sequences ** 2 sequences come back, row i * nrs + j = sample j of prompt row
			# i % R, the prompt tokens in front (:668 cuts at trunc_index only) and counted in max_generate_length (:660).  Restated from source; the loop on
			# such rows is pinned by the reference's own sample_stream (tests/golden/sample_stream.npz, "prompted").
			if candidate_shard is not None:
				raise NotImplementedError("a prompted continuation is not sharded over ranks")
			if input_tokens.dim() != 2 or input_tokens.shape[0] < 1 or num_return_sequences % input_tokens.shape[0] != 0:
				raise ValueError("The number of return sequences must be divisible by the number of input sequences")
			prompt = input_tokens.to(self.device, torch.int64).repeat(num_return_sequences // input_tokens.shape[0], 1).repeat_interleave(num_return_sequences, 0)
			if bool((prompt == self.stop_mel_token).any()):
				raise ValueError("input_tokens must not contain the stop token")
		# omitted keywords mean HF GenerationConfig defaults in the reference (stream_generator.py:262-276): do_sample False (greedy
		# search, not on the hot path: TTS.inference always samples, inference.py:336), top_k 50, temperature / top_p / penalty 1
		if hf_generate_kwargs.get("num_beams", 1) not in (None, 1) or not hf_generate_kwargs.get("do_sample", False):
			raise NotImplementedError("only the sampling branch (do_sample=True, num_beams=1) is implemented; pass do_sample=True")
		gen, _ = self._generate(speech_conditioning_latent, text_inputs, num_return_sequences if prompt is None else prompt.shape[0], max_generate_length,
								typical_mass if typical_sampling else None, hf_generate_kwargs, stream=False, shard=candidate_shard, prompt=prompt)
		return gen

	def inference_speech_lines(self, speech_conditioning_latent, texts, num_return_sequences=1, max_generate_length=None, **hf_generate_kwargs):
		"""`inference_speech` for several text lines as ONE decode batch (no reference counterpart: TTS.inference walks its lines one by one,
		inference.py:244-246, and every line streams the GPT-2 weights again for its 16 candidates; a token step over 2 / 4 lines costs 1.36x /
		2.05x the step over one).  texts: list of [1, Tt_g] id tensors.  Returns a list of int64 [num_return_sequences, L_g]: element g is bit for
		bit what `inference_speech(latent, texts[g], ...)` returns -- each row keeps the cache length of its own line, every line draws the same
		[num_return_sequences, V] multinomial noise (the reference reseeds to 0 per line), and line g is cut where its own last row finished.
		`self.last_generate_lines[g]` says where the generator stands after line g's sampling in the reference (steps, rng_start, rng_step):
		a caller that draws per line afterwards (TTSHotPath.inference_lines) re-positions it with `position_rng_after_line(g)`."""
		if hf_generate_kwargs.get("num_beams", 1) not in (None, 1) or not hf_generate_kwargs.get("do_sample", False):
			raise NotImplementedError("only the sampling branch (do_sample=True, num_beams=1) is implemented; pass do_sample=True")
		texts = list(texts)
		if any(t.dim() != 2 or t.shape[0] != 1 for t in texts):
			raise NotImplementedError("texts: a list of [1, Tt] id tensors, one per line")
		if hf_generate_kwargs.get("input_tokens") is not None:
			raise NotImplementedError("input_tokens (prompted continuation) is not on the inference hot path")
		hf_generate_kwargs.pop("input_tokens", None)
		hf_generate_kwargs.pop("kv_cache", None)
		typical = bool(hf_generate_kwargs.get("typical_sampling", False))
		if len(texts) == 1 or (typical and self.hf_exact_top_p):      # one line, or HF's warpers as torch ops in front of the kernel (built per call): the per-line calls
			out, self.last_generate_lines = [], []
			for t in texts:
				out.append(self.inference_speech(speech_conditioning_latent, t, num_return_sequences=num_return_sequences,
												 max_generate_length=max_generate_length, **hf_generate_kwargs))
				self.last_generate_lines.append(dict(self.last_generate, seed=hf_generate_kwargs.get("seed", 0)))
			return out
		ts, tm = hf_generate_kwargs.pop("typical_sampling", False), hf_generate_kwargs.pop("typical_mass", .9)
		typical_mass = tm if ts else None
		return self._generate_lines(speech_conditioning_latent, texts, num_return_sequences, max_generate_length, hf_generate_kwargs, typical_mass)

	def position_rng_after_line(self, g: int):
		"""leave the torch generators as the reference's `generate` on line g alone leaves them (reseeded, advanced by that line's draws)"""
		info = self.last_generate_lines[g]
		setup_seed(info["seed"])
		torch.cuda.default_generators[self.device.index or 0].set_offset(info["rng_start"] + info["steps"] * info["rng_step"])

	def _generate_lines(self, cond, texts, C, max_generate_length, kw, typical_mass=None):
		c = self.cfg
		self._require_idle()            # a line batch rewrites the KV cache, the noise arming and (through the ring) an open stream's latent buffer
		G = len(texts)
		B = G * C
		if B > self.max_batch:
			raise _lib.TTKError(f"{G} lines x {C} candidates exceed max_batch={self.max_batch}")
		Tmax = max(int(t.shape[1]) for t in texts)
		max_new = (c.max_mel_tokens - 1) if max_generate_length is None else int(max_generate_length)
		if Tmax + 4 + max_new > self.max_ctx or max_new + 2 > c.max_mel_seq_len:
			raise _lib.TTKError(f"prefix {Tmax + 4} + {max_new} new tokens exceed max_ctx={self.max_ctx} or the mel position table ({c.max_mel_seq_len})")
		suppress = tuple(kw.get("suppress_tokens") or ())
		pipe_key = (kw.get("temperature", 1.0), kw.get("top_k", 50), kw.get("top_p", 1.0), kw.get("repetition_penalty", 1.0), suppress, typical_mass)
		can_stop = c.stop_mel_token not in suppress
		seed = kw.get("seed", 0)
		with torch.cuda.device(self.device):
			st = self._gen_state(B, max_new, pipe_key, C, 0, lines=G)
			setup_seed(seed)
			gen = torch.cuda.default_generators[self.device.index or 0]
			off_start = gen.get_offset()
			st.reset(c)
			if st.own_rng:
				st.arm_noise(gen, 0)
			try:
				n = self._token_loop(st, gen, off_start, lambda: self._prefill_lines(cond, texts, C), max_new, can_stop)
			finally:
				if st.own_rng:
					_lib.check(self.lib.ttk_ar_set_noise(self._h, None, None, None), "ttk_ar_set_noise")
			step = st.noise_step if st.own_rng else (gen.get_offset() - off_start) // max(n, 1)
			self._check_health()
			ids = st.ids[:, :n]
			first_stop = torch.where((ids == c.stop_mel_token).any(dim=1), (ids == c.stop_mel_token).float().argmax(dim=1), torch.full((B,), n, device=ids.device)).view(G, C)
			done = bool(can_stop) & (first_stop < n).all(dim=1)
			n_line = torch.where(done, first_stop.max(dim=1).values + 1, torch.full((G,), n, device=ids.device)).tolist()
			out, self.last_generate_lines = [], []
			for g in range(G):
				out.append(ids[g * C:(g + 1) * C, :int(n_line[g])].clone())
				self.last_generate_lines.append(dict(steps=int(n_line[g]), rng_start=off_start, rng_step=step, seed=seed))
			self.position_rng_after_line(G - 1)
			self.last_generate = dict(self.last_generate_lines[-1])
			return out

	def _require_idle(self):
		"""one generation at a time per handle: the KV cache, the noise arming and the latent ring of an open streamed generation are the handle's (the
		reference's module is re-entrant because HF keeps all of that per call); starting another one would corrupt both silently, so it is refused"""
		if self._streaming:
			raise _lib.TTKError("a streamed generation is still open on this model: exhaust or close() its generator before starting another generation")

	def compute_embeddings(self, cond_latents, text_inputs, kv_cache=True):
		"""unified_voice.py:614-630: remembers the prefix, returns the fake id row [b, P+1]."""
		self._prefix = (cond_latents, text_inputs)
		P1 = text_inputs.shape[1] + 4
		ids = torch.ones((cond_latents.shape[0], P1), dtype=torch.long, device=self.device)
		ids[:, -1] = self.start_mel_token
		return ids

	def get_generator(self, inputs, max_length=500, **hf_generate_kwargs) -> Iterator[Tuple[torch.Tensor, torch.Tensor]]:
		"""unified_voice.py:670-679 / stream_generator.py:911-1190: yields (codes [B], latent [B, model_dim]) per token,
		latent = final_norm(last hidden) (:1172)."""
		if self._prefix is None:
			raise _lib.TTKError("call compute_embeddings first")
		cond, text = self._prefix
		n_new = max_length - inputs.shape[1]
		B = hf_generate_kwargs.pop("num_return_sequences", 1) or 1
		return self._generate(cond, text, B, n_new, None, hf_generate_kwargs, stream=True)

	# ------------------------------------------------------------------ the token loop
	def _generate(self, cond, text, num_return_sequences, max_generate_length, typical_mass, kw, stream, shard=None, prompt=None):
		c = self.cfg
		self._require_idle()
		n_in = 0 if prompt is None else int(prompt.shape[1])
		if stream and n_in:
			raise NotImplementedError("the streaming generator takes no prompt tokens")
		C = num_return_sequences * text.shape[0]          # candidates the noise is drawn for
		lo, hi = (0, C) if shard is None else (int(shard[0]), int(shard[1]))
		if not (0 <= lo < hi <= C):
			raise ValueError(f"candidate_shard {shard} is not a non-empty range inside [0, {C})")
		if stream and shard is not None:
			raise NotImplementedError("the streaming generator is not sharded")
		B = hi - lo
		if B > self.max_batch:
			raise _lib.TTKError(f"{B} candidates exceed max_batch={self.max_batch}")
		Tt = text.shape[1]
		trunc_index = Tt + 4
		max_new = (c.max_mel_tokens - 1) if max_generate_length is None else int(max_generate_length)
		if trunc_index + max_new > self.max_ctx or max_new + 2 > c.max_mel_seq_len:
			raise _lib.TTKError(f"prefix {trunc_index} + {max_new} new tokens exceed max_ctx={self.max_ctx} "
								f"or the mel position table ({c.max_mel_seq_len})")
		if n_in >= max_new:      # HF's stopping criterion is only looked at after a token has been appended: the reference would still sample one
			raise ValueError(f"{n_in} prompt tokens leave no room below max_generate_length={max_new}")
		suppress = tuple(kw.get("suppress_tokens") or ())
		pipe_key = (kw.get("temperature", 1.0), kw.get("top_k", 50), kw.get("top_p", 1.0), kw.get("repetition_penalty", 1.0),
					suppress, typical_mass)
		if stream:
			return self._loop_stream(cond, text, B, max_new, pipe_key, c.stop_mel_token not in suppress)
		can_stop = c.stop_mel_token not in suppress
		with torch.cuda.device(self.device):
			st = self._gen_state(B, max_new, pipe_key, C, lo)
			setup_seed(kw.get("seed", 0))
			gen = torch.cuda.default_generators[self.device.index or 0]
			off_start = gen.get_offset()
			st.reset(c, prompt)
			if st.own_rng:
				st.arm_noise(gen, lo, n_in)
			try:
				n = self._token_loop(st, gen, off_start, lambda: self._prefill(cond, text, B, prompt), max_new, can_stop, n_in)
			finally:
				if st.own_rng:
					_lib.check(self.lib.ttk_ar_set_noise(self._h, None, None, None), "ttk_ar_set_noise")
			if st.own_rng:
				gen.set_offset(off_start + (n - n_in) * st.noise_step)        # what the n - n_in torch draws would have consumed
			# what the sampling consumed from the generator: dist.py aligns a shard's stream with the unsharded run's from this
			self.last_generate = dict(steps=n - n_in, rng_start=off_start, rng_step=(gen.get_offset() - off_start) // max(n - n_in, 1))
			self._check_health()
			return st.ids[:, :n].clone(), None

	def _token_loop(self, st, gen, off_start, prefill, max_new, can_stop, n0=0):
		"""n0: id columns that are filled already (the prompt tokens of a prompted continuation); returns the filled columns at the end"""
		c = self.cfg
		st.logits.copy_(prefill())
		n = n0
		if not (self.use_graph and st.graphable):
			while True:
				st.sample(n)
				n += 1
				if n >= max_new or (can_stop and int(st.unfinished.max()) == 0):
					break
				self._decode_next(st.logits)
		else:
			# tokens 1 and 2 eagerly (the second pass also warms every kernel before a capture), then one HIP-graph
			# replay per token: {ttk_ar_decode_next; [typical warper]; exponential_; ttk_ar_sample_next}.  Every position-dependent
			# quantity (cache length, mel position, output column, length of the token history the repetition penalty reads) lives in
			# device memory, so ONE graph serves all tokens and every text length.
			# HF's stopping test (`unfinished_sequences.max() == 0` after every token, a host round trip that idles the GPU)
			# becomes a flag the sampling kernel raises in pinned memory; the host looks at it LAG replays late, so the GPU
			# always has work queued.  The <= LAG tokens generated past the true end are all padding; they are cut off below
			# and the generator offset they consumed is handed back, so ids AND the RNG stream equal the reference's.
			LAG = 2
			st.sample(n0)
			if st.rng_step is None:
				st.rng_step = gen.get_offset() - off_start
			n = n0 + 1
			events = []
			stopped = can_stop and int(st.unfinished.max()) == 0
			while n < max_new and not stopped:
				if st.graph is None:
					self._decode_next(st.logits)
					st.sample(n)
					n += 1
					stopped = can_stop and int(st.unfinished.max()) == 0
					if n < max_new and not stopped:
						torch.cuda.synchronize(self.device)
						g = torch.cuda.CUDAGraph()
						# thread-local capture mode: another host thread may be enqueuing (and allocating for) the previous
						# line's diffusion meanwhile (TTSHotPath.inference_lines); in the default global mode its hipMalloc /
						# hipFree would invalidate this capture.
						# capture_begin / capture_end run OUTSIDE inference mode whatever the caller's mode: torch creates the
						# generator's graph-side seed / offset tensors at the first capture and updates them in place at every
						# later capture and replay -- created under inference_mode they would make any later capture from a
						# caller without it fail ("inplace update to inference tensor outside InferenceMode").
						ctx = torch.cuda.graph(g, capture_error_mode="thread_local")
						with torch.inference_mode(False):
							ctx.__enter__()
						try:
							self._decode_next(st.logits)
							st.sample(0)
						except BaseException:
							with torch.inference_mode(False):
								ctx.__exit__(*__import__("sys").exc_info())
							raise
						with torch.inference_mode(False):
							ctx.__exit__(None, None, None)
						st.graph = g
						# no random numbers are drawn inside the captured step when the mel head draws the noise: launch the instantiated
						# graph directly then -- CUDAGraph.replay() first refills the generator's seed / offset tensors, two launches per token
						st.graph_exec = g.raw_cuda_graph_exec() if st.own_rng and os.environ.get("TTK_AR_RAW_REPLAY", "1") != "0" else None
					continue
				if st.graph_exec:
					_lib.check(self.lib.ttk_graph_launch(st.graph_exec, _lib.stream_ptr()), "ttk_graph_launch")
				else:
					st.graph.replay()
				n += 1
				if can_stop:
					ev = torch.cuda.Event()
					ev.record()
					events.append(ev)
					if len(events) > LAG:
						events.pop(0).synchronize()          # the replay LAG tokens back is complete: its flag is visible
						stopped = int(st.done[0]) != 0
			if can_stop:
				# exact end: HF stops right after the token with which the last row finishes
				torch.cuda.synchronize(self.device)
				ids = st.ids[:, :n]
				is_stop = ids == c.stop_mel_token
				if bool(is_stop.any(dim=1).all()):
					n_true = max(int(is_stop.float().argmax(dim=1).max()) + 1, n0 + 1)
					if n_true < n:
						if not st.own_rng:
							gen.set_offset(gen.get_offset() - (n - n_true) * st.rng_step)
						n = n_true
		return n

	def _gen_state(self, B, max_new, pipe_key, C=None, lo=0, lines=1):
		"""generation states (device buffers + the captured token step) keyed by what is baked into them; a few are kept so that
		alternating shapes (e.g. `TTS.inference` lines with different max lengths) do not re-capture every call"""
		C = B if C is None else C
		key = (B, max_new, pipe_key, C, lo, lines)
		states = self._states
		if key in states:
			states[key] = states.pop(key)             # most recently used last
		else:
			while len(states) >= 4:
				states.pop(next(iter(states)))
			states[key] = _GenState(self, B, max_new, pipe_key, C, lo, lines)
		return states[key]

	def _loop_stream(self, cond, text, B, max_new, pipe_key, can_stop):
		"""stream_generator.py:1106-1190, pinned by the reference's own loop (tests/golden/sample_stream.npz): token k is yielded with
		final_norm(hidden) of the forward it was sampled from -- the prefill's last row for the first token -- and every token is
		yielded, the last one included; the loop ends after the token with which the last row finishes or after max_new tokens.

		Same machinery as the non-streaming loop: the fused sampling launch (processors, warpers, multinomial, padding, next input row), the noise
		drawn in the mel-head launch, ONE captured token step replayed per token.  Between two yields nothing runs on the host but a graph launch
		and an event wait: the yielded tokens are a column of the state's id buffer, the latents a slot of a per-call [max_new, B, d] buffer that
		the step's LayerNorm launch fills directly (ttk_ar_set_hidden_ring: the slot index is the device-side token counter, so the captured launch
		serves every step).  The host runs LAG steps ahead of the consumer; HF's `unfinished_sequences.max() == 0` test (a host round trip per
		token) is the pinned word the sampling kernel leaves -- the token count at which the last row finished -- read after the event of the
		token about to be yielded.  The yielded tensors are slots of two per-call buffers (tokens [max_new, B], latents [max_new, B, d]): like the
		reference's fresh tensors they stay valid whatever runs on the model afterwards."""
		c = self.cfg
		LAG = 2
		self._require_idle()            # (a generator body runs at its first next(): two generators may have been created, only one may run)
		with torch.cuda.device(self.device):
			st = self._gen_state(B, max_new, pipe_key, B, 0)
			setup_seed(0)
			gen = torch.cuda.default_generators[self.device.index or 0]
			off_start = gen.get_offset()
			st.reset(c)
			if st.own_rng:
				st.arm_noise(gen, 0)
			hid = torch.empty((max_new, B, c.model_dim), device=self.device, dtype=torch.float32)
			# the yielded tokens get this call's lifetime too, as the reference's fresh tensors have (stream_generator.py:1172): `st.ids` belongs to the
			# cached generation state, which the next generation of the same shape -- streamed or not -- refills
			toks = torch.empty((max_new, B), device=self.device, dtype=torch.long)
			fast = self.use_graph and st.graphable and st.own_rng        # captured step; else the same launches issued eagerly
			n_done = 0
			self._streaming = True
			try:
				st.logits.copy_(self._prefill(cond, text, B))
				_lib.check(self.lib.ttk_ar_last_hidden(self._h, hid[0].data_ptr(), _lib.stream_ptr()), "ttk_ar_last_hidden")
				_lib.check(self.lib.ttk_ar_set_hidden_ring(self._h, hid.data_ptr(), st.col.data_ptr(), B * c.model_dim, _lib.stream_ptr()), "ttk_ar_set_hidden_ring")
				events, produced = [], 0
				while True:
					while produced < min(max_new, n_done + 1 + LAG):       # tokens n_done .. n_done + LAG sampled or in flight
						if produced == 0:
							st.sample(0)
							if st.rng_step is None:
								st.rng_step = gen.get_offset() - off_start       # what one torch-drawn token consumes (0 with the head-drawn noise)
						elif fast and st.stream_graph is not None:
							_lib.check(self.lib.ttk_graph_launch(st.stream_graph_exec, _lib.stream_ptr()), "ttk_graph_launch")
						elif fast and produced >= 2:
							# capture {decode_next into the ring; sample}: step 1 ran eagerly (warm kernels), the stream is idle of other work
							torch.cuda.synchronize(self.device)
							g = torch.cuda.CUDAGraph()
							ctx = torch.cuda.graph(g, capture_error_mode="thread_local")
							with torch.inference_mode(False):
								ctx.__enter__()
							try:
								self._decode_next(st.logits)
								st.sample(0)
							except BaseException:
								with torch.inference_mode(False):
									ctx.__exit__(*__import__("sys").exc_info())
								raise
							with torch.inference_mode(False):
								ctx.__exit__(None, None, None)
							st.stream_graph, st.stream_graph_exec = g, g.raw_cuda_graph_exec()
							continue
						else:
							self._decode_next(st.logits)
							st.sample(produced)
						ev = torch.cuda.Event()
						ev.record()
						events.append(ev)
						produced += 1
					events[n_done].synchronize()                               # token n_done and its flag are there
					toks[n_done].copy_(st.ids[:, n_done])
					yield toks[n_done], hid[n_done]
					n_done += 1
					end = int(st.done[0]) if can_stop else 0                   # tokens sampled when the last row finished (0: still running)
					if n_done >= max_new or (end and end <= n_done):
						return
			finally:
				self._streaming = False
				_lib.check(self.lib.ttk_ar_set_hidden_ring(self._h, None, None, 0, _lib.stream_ptr()), "ttk_ar_set_hidden_ring")
				if st.own_rng:
					_lib.check(self.lib.ttk_ar_set_noise(self._h, None, None, None), "ttk_ar_set_noise")
					# the generator where the reference's loop leaves it: one draw per token it produced (steps sampled ahead of an early end drew nothing
					# from torch; without the head-drawn noise they did, and are handed back)
					gen.set_offset(off_start + n_done * st.noise_step)
				elif st.rng_step:
					gen.set_offset(off_start + n_done * st.rng_step)
				self._check_health()


class _GenState:
	"""Persistent device buffers of one generation shape, so a captured token step can be replayed across calls (and across text
	lengths: nothing in it depends on the prefix length)."""

	def __init__(self, model: UnifiedVoice, B, max_new, pipe_key, C=None, lo=0, lines=1):
		c, dev = model.cfg, model.device
		C = B if C is None else C
		self.lines, self.noise_rows = lines, C                          # lines > 1: B = lines * C rows, every line draws the same [C, V] noise
		self.model = model
		self.B, self.max_new = B, max_new
		self.pipe = LogitsPipeline(temperature=pipe_key[0], top_k=pipe_key[1], top_p=pipe_key[2], repetition_penalty=pipe_key[3],
								   suppress_tokens=pipe_key[4], typical_mass=pipe_key[5], vocab=c.number_mel_codes, device=dev)
		self.stop = c.stop_mel_token
		self.logits = torch.empty((B, c.number_mel_codes), device=dev, dtype=torch.float32)
		self.ids = torch.empty((B, max_new), dtype=torch.long, device=dev)
		self.tok = torch.empty(B, dtype=torch.long, device=dev)
		self.unfinished = torch.ones(B, dtype=torch.long, device=dev)
		self.col = torch.zeros(B, dtype=torch.long, device=dev)         # per-row output column (all rows move together)
		# Exp(1) noise of multinomial for ALL C candidates of the call (C == B unless this is one shard of a candidate-sharded run:
		# every rank draws the same [C, V] block and reads its rows lo..lo+B-1, see inference_speech)
		self.q = torch.empty((B if lines > 1 else C, c.number_mel_codes), device=dev, dtype=torch.float32)
		self.qc = torch.empty((C, c.number_mel_codes), device=dev, dtype=torch.float32) if lines > 1 else None      # torch-drawn fallback of a line batch
		self.live = torch.zeros(1, dtype=torch.int32, device=dev)        # unfinished rows, decremented on the device
		self.done = torch.zeros(1, dtype=torch.int32).pin_memory()       # raised by the row that finishes last; polled by the host
		self.rng_step = None                                             # generator offset consumed by one sample() call
		# torch's `q.exponential_(1)` restated inside the mel-head launch (include/ttk.h: ttk_ar_set_noise) -- one launch per token less.
		# Relied on only after a bitwise comparison with torch's own draw on this device, for this very shape (below).
		self.rng = torch.zeros(6, dtype=torch.long, device=dev)          # RngArgs {seed, offset0, threads, step, row0, candidates per line or 0}
		self.own_rng = os.environ.get("TTK_AR_OWN_RNG", "1") != "0" and self._noise_matches_torch(model, dev)
		# input_ids as the repetition penalty sees them: the fake prefix ids are all 1 with start_mel last (unified_voice.py:647-649),
		# i.e. the SET {1, start_mel} whatever the text length (the penalty acts once per distinct id), then the sampled tokens
		self.history = None
		if self.pipe.needs_history:
			self.history = torch.ones((B, 2 + max_new), dtype=torch.long, device=dev)
		# every processor and warper runs inside the fused kernel (V <= 9216: the row lives in registers), the typical-sampling warper included
		# (round 3).  With hf_exact_top_p the cumulative-mass warpers (top-p, typical) run as torch ops in front of it instead, in HF's order, which
		# means the processors then run as torch ops too and the kernel sees finished scores for that part
		p = self.pipe
		self.in_kernel = c.number_mel_codes <= 9216 and not (model.hf_exact_top_p and (p.top_p is not None or p.typical_mass is not None))
		self.graphable = self.in_kernel or not p.needs_history      # torch-op penalty: its history slice grows with the host's step count
		a = _lib.SampleArgs()
		a.ld, a.B, a.V = self.logits.stride(0), B, c.number_mel_codes
		a.q, a.ldq = self.q[lo].data_ptr(), self.q.stride(0)
		a.stop_token = self.stop
		a.unfinished, a.tok, a.ids = self.unfinished.data_ptr(), self.tok.data_ptr(), self.ids.data_ptr()
		a.ids_ld, a.ids_cols, a.col = self.ids.stride(0), self.ids.shape[1], self.col.data_ptr()
		if self.history is not None:
			a.history, a.hist_ld, a.hist_off = self.history.data_ptr(), self.history.stride(0), 2
		a.live_rows, a.all_done = self.live.data_ptr(), self.done.data_ptr()
		if self.in_kernel:
			a.scores = self.logits.data_ptr()
			a.suppress = _lib.ptr(p.suppress_mask)
			a.temperature = p.temperature or 1.0
			a.top_k, a.top_p, a.repetition_penalty = p.top_k or 0, p.top_p or 1.0, p.repetition_penalty or 1.0
			a.typical_mass = float(p.typical_mass or 0.0)
		else:
			a.temperature, a.top_k, a.top_p, a.repetition_penalty = 1.0, 0, 1.0, 1.0
		self.args = a
		self.graph = None
		self.graph_exec = None
		self.stream_graph = None          # the streaming generator's captured step (it also fills the hidden ring)
		self.stream_graph_exec = None

	def _noise_geometry(self, dev):
		"""(threads, offset step per draw) of ATen's launch for `self.q.exponential_()` (ATen/native/cuda/DistributionTemplates.h:
		distribution_nullary_kernel -- 256-thread blocks, a grid capped at the resident blocks of the device, four values per Philox call)"""
		props = torch.cuda.get_device_properties(dev)
		numel = self.noise_rows * self.q.shape[1]
		grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
		threads = 256 * grid
		return threads, ((numel - 1) // (threads * 4) + 1) * 4

	def _noise_matches_torch(self, model, dev):
		gen = torch.cuda.default_generators[dev.index or 0]
		keep = gen.get_state()
		try:
			threads, step = self._noise_geometry(dev)
			seed, off = gen.initial_seed(), gen.get_offset()
			seed = seed - (1 << 64) if seed >= (1 << 63) else seed
			shape = (self.noise_rows, self.q.shape[1])
			want = [torch.empty(shape, device=dev).exponential_(1) for _ in range(2)]
			if gen.get_offset() - off != 2 * step:
				return False
			got = torch.empty(shape, device=dev)
			for draw in range(2):
				_lib.check(model.lib.ttk_exponential_like_torch(got.data_ptr(), got.numel(), seed, off, threads, step, draw, _lib.stream_ptr()),
						   "ttk_exponential_like_torch")
				if not torch.equal(got.view(torch.int32), want[draw].view(torch.int32)):
					return False
			self.noise_threads, self.noise_step = threads, step
			return True
		finally:
			gen.set_state(keep)

	def arm_noise(self, gen, lo, n0=0):
		"""point the mel-head launches of this call at q, starting from the generator's current state.  n0: id columns filled before the first draw (prompt
		tokens): the launches number their draws by the column counter, so the first offset is moved back by n0 draws"""
		seed = gen.initial_seed()
		seed = seed - (1 << 64) if seed >= (1 << 63) else seed
		self.rng.copy_(torch.tensor([seed, gen.get_offset() - n0 * self.noise_step, self.noise_threads, self.noise_step, lo, self.noise_rows if self.lines > 1 else 0], dtype=torch.long))
		m = self.model
		_lib.check(m.lib.ttk_ar_set_noise(m._h, self.rng.data_ptr(), self.col.data_ptr(), self.q[lo].data_ptr()), "ttk_ar_set_noise")

	def reset(self, c, prompt=None):
		self.ids.fill_(self.stop)
		self.unfinished.fill_(1)
		self.col.zero_()
		self.live.fill_(self.B)
		self.done.zero_()
		if self.history is not None:
			self.history.fill_(1)
			self.history[:, 1] = c.start_mel_token
		if prompt is not None and prompt.shape[1]:      # a prompted continuation: the prompt tokens are the first id columns (and part of what the repetition penalty sees)
			n0 = prompt.shape[1]
			self.ids[:, :n0] = prompt
			self.col.fill_(n0)
			if self.history is not None:
				self.history[:, 2:2 + n0] = prompt

	def sample(self, n):
		"""one token from self.logits: process / warp, sample, pad finished rows, record, and write the next step's input row
		(HF:generation/utils.py:2894-2937; unified_voice.py:212-214)"""
		a = self.args
		if not self.in_kernel:
			hist = None if self.history is None else self.history[:, :2 + n]
			scores = self.pipe(hist, self.logits)
			self.scores = scores if scores.is_contiguous() else scores.contiguous()      # kept alive until the launch has run
			a.scores, a.ld = self.scores.data_ptr(), self.scores.stride(0)
		# multinomial(softmax(scores), 1) == argmax(softmax(scores) / q), q ~ Exp(1) from the torch generator (see sampling.multinomial1)
		# (own_rng: the mel-head launch that produced self.logits has written q already)
		if not self.own_rng:
			if self.lines > 1:
				self.qc.exponential_(1)
				self.q.view(self.lines, *self.qc.shape).copy_(self.qc)
			else:
				self.q.exponential_(1)
		_lib.check(self.model.lib.ttk_ar_sample_next(self.model._h, _lib.C.byref(a), _lib.stream_ptr()), "ttk_ar_sample_next")
