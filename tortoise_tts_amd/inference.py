"""The hot-path section of `TTS.inference` (/root/reference/tortoise_tts/inference.py:331-413) over the libttk-backed
modules: AR sampling -> stop-token fix-up -> latent pass -> calm-token trim -> [candidate pick] -> timestep-independent
conditioning -> DDIM loop -> mel denormalisation.  What comes before (tokenizer, conditioning latents: tokenizer.py, mel.py,
conditioning.py) and after (CLVP, vocoder: clvp.py, vocoder.py) is assembled around this class by `tortoise_tts_amd.tts.TTS`;
here those stages appear as inputs (token ids, latents) and optional attachments (`clvp=`, `vocoder=`).
"""
from __future__ import annotations

from typing import Optional

import torch

from .autoregressive import UnifiedVoice
from .diffusion import DiffusionTTS, denormalize_tacotron_mel, get_diffuser

CALM_TOKEN = 83
SAMPLE_RATE = 24_000
HOP = 256   # BigVGAN 24 kHz / 100-band hop (assumed: its config JSON is download-only)


def fix_stop_tokens(codes: torch.Tensor, stop_mel_token: int) -> torch.Tensor:
	"""inference.py:353-366.  The reference takes `.min()` of the stop positions before checking that any exist (:355 vs
	:357) and raises on a row without a stop token; such rows are left untouched here (what the check intends)."""
	codes = codes.clone()
	is_stop = codes == stop_mel_token
	has = is_stop.any(dim=1)
	if not bool(has.any()):
		return codes
	L = codes.shape[1]
	first = torch.where(has, is_stop.float().argmax(dim=1), torch.full_like(has, L, dtype=torch.long))
	pos = torch.arange(L, device=codes.device)[None, :]
	codes = torch.where((pos >= first[:, None]) & has[:, None], torch.full_like(codes, 83), codes)
	tail = torch.tensor([45, 45, 248], device=codes.device, dtype=codes.dtype)
	codes[has, -3:] = tail
	return codes


def trim_calm_tokens(codes: torch.Tensor, latents: torch.Tensor) -> torch.Tensor:
	"""inference.py:381-389 on row 0: cut the latents at the position where the 9th consecutive calm token sits."""
	row = codes[0].tolist()            # one host copy instead of a sync per position
	calm = 0
	for k, c in enumerate(row):
		calm = calm + 1 if c == CALM_TOKEN else 0
		if calm > 8:
			return latents[:, :k]
	return latents


class HotPathStages:
	"""`dist.ShardStages` on the libttk-backed modules: what one rank of a candidate-sharded utterance runs (dist.sharded_candidates)."""

	def __init__(self, tts: "TTSHotPath", text_tokens, autoregressive_latents, diffusion_latents, *, max_ar_steps=500,
				 max_diffusion_steps=80, ar_temp=0.8, diffusion_temp=1.0, top_p=1.0, top_k=0, repetition_penalty=1.0, length_penalty=1.0,
				 diffusion_sampler="ddim", cond_free=True, suppress_tokens=None, phase_marks=None):
		self.tts, self.ar, self.diff = tts, tts.autoregressive, tts.diffusion
		self.phase_marks = phase_marks      # measurement only: receives (name, event) at the phase boundaries, as TTSHotPath.inference does
		self.text = text_tokens.to(self.ar.device)
		self.al, self.dl = autoregressive_latents, diffusion_latents
		self.kw = dict(do_sample=True, top_k=top_k, top_p=top_p, temperature=ar_temp, num_beams=1, length_penalty=length_penalty,
					   repetition_penalty=repetition_penalty, max_generate_length=max_ar_steps)
		if suppress_tokens:
			self.kw["suppress_tokens"] = suppress_tokens
		self.diffuser = get_diffuser(steps=max_diffusion_steps, cond_free=cond_free)
		self.diffusion_temp, self.sampler = diffusion_temp, diffusion_sampler
		# dist.ShardStages: the ancestral sampler draws its per-step noise inside the loop, from the generator of the rank that runs it -- such a line stays on its winner's owner
		self.diffusion_draws_while_running = diffusion_sampler != "ddim"
		self.pad_token = self.ar.stop_mel_token

	def _mark(self, name, *extra):
		if self.phase_marks is not None:
			ev = torch.cuda.Event(enable_timing=True)
			ev.record()
			self.phase_marks.append((name, ev) + extra)

	def sample(self, lo, hi, n_candidates):
		self._mark("start")
		ids = self.ar.inference_speech(self.al, self.text, num_return_sequences=n_candidates, candidate_shard=(lo, hi), **self.kw)
		self._mark("ar_decode")
		return ids

	def align_rng(self, steps):
		g = self.ar.last_generate
		torch.cuda.default_generators[self.ar.device.index or 0].set_offset(g["rng_start"] + steps * g["rng_step"])

	def latents(self, ids):
		codes = fix_stop_tokens(ids, self.ar.stop_mel_token)
		B, M = codes.shape
		al = self.al.expand(B, -1) if self.al.shape[0] != B else self.al
		lat = self.ar.forward(al, self.text.expand(B, -1), torch.tensor([self.text.shape[1]], dtype=torch.int32).expand(B), codes,
							  torch.tensor([M * self.ar.mel_length_compression]).expand(B), return_latent=True, clip_inputs=False)
		self._mark("latent_pass")           # (includes the id all-gather that precedes it)
		return codes, lat

	def score(self, codes):
		return None if self.tts.clvp is None else self.tts.clvp(self.text, codes, return_loss=False)

	def prepare_diffusion(self, codes, latents):
		"""the part of `diffuse` that touches the torch generator, in the reference's order (inference.py:381-404, diffusion.py:685): calm-token
		trim, frame count, the start noise and DDIM's ignored per-step draws.  Returns what `run_diffusion` needs."""
		latents = trim_calm_tokens(codes, latents)
		T = latents.shape[1] * 4 * 24000 // 22050
		noise = torch.randn((1, 100, T), device=self.ar.device) * self.diffusion_temp
		if self.sampler == "ddim":
			for _ in range(self.diffuser.num_timesteps):
				torch.randn_like(noise)
		return latents, noise, T

	def pack_prepared(self, prepared):
		latents, noise, _ = prepared
		return [latents.to(torch.float32), noise.to(torch.float32)]

	def unpack_prepared(self, tensors):
		latents, noise = tensors
		return latents, noise, int(noise.shape[-1])

	def run_diffusion(self, prepared):
		"""mels [1, 100, T_i] of the prepared lines: one ragged DDIM batch (SpacedDiffusion.sample_loop_lines) when there are several and the
		sampler is ddim with conditioning-free guidance, else one loop per line; each mel bit for bit its own loop's either way"""
		self._mark("_before_ddim")          # what lies between this line's latent pass and the shared diffusion (later lines' sampling, item moves) is not "ddim"
		Es = [self.diff.timestep_independent(latents, self.dl, T, False) for latents, _, T in prepared]
		if len(prepared) > 1 and self.sampler == "ddim" and self.diffuser.conditioning_free:
			mels = self.diffuser.sample_loop_lines(self.diff, [n for _, n, _ in prepared], Es)
		else:
			mels = [self.diffuser.sample_loop(self.diff, (1, 100, T), sampler=self.sampler, noise=n, model_kwargs={"precomputed_aligned_embeddings": E},
											  progress=False, consume_rng=self.sampler != "ddim") for (_, n, T), E in zip(prepared, Es)]
		self._mark("ddim", len(prepared))      # (name, event, lines diffused in this batch): bench.phase_roofline prices the interval with that many lines' work
		return mels

	def diffuse(self, codes, latents):
		return self.run_diffusion([self.prepare_diffusion(codes, latents)])[0]


class TTSHotPath:
	def __init__(self, autoregressive: UnifiedVoice, diffusion: DiffusionTTS, vocoder=None, clvp=None):
		self.autoregressive, self.diffusion, self.vocoder, self.clvp = autoregressive, diffusion, vocoder, clvp

	@torch.inference_mode()
	def inference_sharded(self, text_tokens, autoregressive_latents, diffusion_latents, *, candidates, group=None, return_all=False, **kw):
		"""`inference` for ONE utterance whose `candidates` are sharded over the ranks of `group` (torch.distributed: RCCL over xGMI on
		the GPU box): BASELINE configs[3].  Every rank calls this with the same arguments and gets the same result, which equals the
		single-GPU `inference(..., candidates=candidates)` of a TTSHotPath with the same attachments (tortoise_tts_amd/dist.py states
		the RNG contract that makes it so).  Candidate choice as in `inference`: the best CLVP score when a scorer is attached, else
		candidate 0."""
		from . import dist as D
		st = HotPathStages(self, text_tokens, autoregressive_latents, diffusion_latents, **kw)
		mel, ids, scores, best = D.sharded_candidates(st, candidates, group)
		T = mel.shape[-1]
		mels = denormalize_tacotron_mel(mel)[:, :, :T]
		seconds = T * HOP / SAMPLE_RATE
		if return_all:
			return mels, seconds, dict(codes=fix_stop_tokens(ids, self.autoregressive.stop_mel_token), mel=mel, scores=scores, best=best)
		return mels, seconds

	@torch.inference_mode()
	def inference_sharded_lines(self, lines, autoregressive_latents, diffusion_latents, *, candidates, group=None, return_all=False, phase_marks=None, **kw):
		"""`inference_sharded` for the lines of ONE long-form text (BASELINE configs[3]: "2 lines"): every line's candidates are sharded over the
		ranks as there, line by line (sample, id gather, latent pass, scores, winner), and the winners are then DIFFUSED TOGETHER -- the lines a rank
		owns as one ragged DDIM batch (dist.sharded_candidates_lines) -- before the mels are broadcast.  Each line's result equals its own
		`inference_sharded` call.  phase_marks (measurement only): a list that receives one list of (name, event) per line; the shared
		diffusion's end is the "ddim" mark of every line it served.  The lines' diffusions are spread over the ranks (dist.assign_diffusers): a line
		whose winner lives on a busier rank has its latents + start noise sent to an idle one.  Returns a list of (mels, seconds[, aux])."""
		from . import dist as D
		stages = []
		for text in lines:
			lm = None
			if phase_marks is not None:
				lm = []
				phase_marks.append(lm)
			stages.append(HotPathStages(self, text, autoregressive_latents, diffusion_latents, phase_marks=lm, **kw))
		out = []
		for st, (mel, ids, scores, best) in zip(stages, D.sharded_candidates_lines(stages, candidates, group)):
			T = mel.shape[-1]
			mels = denormalize_tacotron_mel(mel)[:, :, :T]
			seconds = T * HOP / SAMPLE_RATE
			out.append((mels, seconds, dict(codes=fix_stop_tokens(ids, self.autoregressive.stop_mel_token), mel=mel, scores=scores, best=best)) if return_all else (mels, seconds))
		return out

	@torch.inference_mode()
	def inference_to_wav(self, text_tokens, autoregressive_latents, diffusion_latents, **kw):
		"""`inference` followed by the vocoder pass of inference.py:415-425 (`vocoder.inference(mels)`): returns (wav [1, 1, T * hop], 24000)."""
		if self.vocoder is None:
			raise ValueError("TTSHotPath was built without a vocoder (tortoise_tts_amd.BigVGAN)")
		mels, _ = self.inference(text_tokens, autoregressive_latents, diffusion_latents, **kw)
		return self.vocoder.inference(mels), SAMPLE_RATE

	@torch.inference_mode()
	def inference(self, text_tokens: torch.Tensor, autoregressive_latents: torch.Tensor, diffusion_latents: torch.Tensor, *,
				  max_ar_steps=500, max_diffusion_steps=80, ar_temp=0.8, diffusion_temp=1.0, top_p=1.0, top_k=0,
				  repetition_penalty=1.0, length_penalty=1.0, diffusion_sampler="ddim", cond_free=True, candidates=1,
				  suppress_tokens=None, return_all=False, phase_marks=None, latents_for="all"):
		"""text_tokens [1, Tt] int64; latents from the reference's conditioning encoders ([1,1024], [1,2048]).
		latents_for: "all" = the latent pass over every candidate, as the reference runs it before scoring (inference.py:370-379; the benchmarked
		workload); "winner" = the k = 1 variant: the candidate is chosen first and only its row goes through the dense
		pass (the reference's own to-do at :370) -- rows of the pass are independent, so the result is the same bits.
		Returns the denormalised mel [1, 100, T] (input of the vocoder) and the audio seconds it represents.
		phase_marks (measurement only): a list that receives (name, torch.cuda.Event) at the phase boundaries -- start, after the AR
		sampling, after the latent pass, after the diffusion -- for bench.py's per-phase roofline."""
		ar, diff = self.autoregressive, self.diffusion
		dev = ar.device

		def mark(name):
			if phase_marks is not None:
				ev = torch.cuda.Event(enable_timing=True)
				ev.record()
				phase_marks.append((name, ev))
		mark("start")
		text_tokens = text_tokens.to(dev)
		diffuser = get_diffuser(steps=max_diffusion_steps, cond_free=cond_free)
		extra = {"suppress_tokens": suppress_tokens} if suppress_tokens else {}
		codes = ar.inference_speech(autoregressive_latents, text_tokens, do_sample=True, top_k=top_k, top_p=top_p,
									temperature=ar_temp, num_return_sequences=candidates, num_beams=1,
									length_penalty=length_penalty, repetition_penalty=repetition_penalty,
									max_generate_length=max_ar_steps, **extra)
		mark("ar_decode")
		codes = fix_stop_tokens(codes, ar.stop_mel_token)
		B, M = codes.shape
		wav_lengths = torch.tensor([M * ar.mel_length_compression])
		text_lengths = torch.tensor([text_tokens.shape[1]], dtype=torch.int32)
		if latents_for not in ("all", "winner"):
			raise ValueError("latents_for: 'all' or 'winner'")
		pre_best, pre_scores = 0, None
		if latents_for == "winner":
			if self.clvp is not None and B > 1:
				pre_scores = self.clvp(text_tokens, codes, return_loss=False)
				pre_best = int(torch.argmax(pre_scores))
			al_row = autoregressive_latents if autoregressive_latents.shape[0] == 1 else autoregressive_latents[pre_best:pre_best + 1]
			latents = ar.forward(al_row, text_tokens, text_lengths, codes[pre_best:pre_best + 1], wav_lengths, return_latent=True, clip_inputs=False)
		else:
			latents = ar.forward(autoregressive_latents.expand(B, -1) if autoregressive_latents.shape[0] != B else autoregressive_latents,
								 text_tokens.expand(B, -1), text_lengths.expand(B), codes, wav_lengths.expand(B),
								 return_latent=True, clip_inputs=False)
		# Candidate choice.  The reference scores the candidates with CLVP and reorders `codes` (inference.py:392-396) but diffuses the
		# latents computed BEFORE that, in generation order (its own to-do at :370), trimmed where row 0 goes calm (:381-389).  Without
		# a CLVP model this path keeps that observable behaviour for the first candidate: row 0, trimmed by row 0.  With one
		# (`TTSHotPath(..., clvp=)`) it does what the to-do asks for: the best-scoring candidate's latents, trimmed by its own codes.
		mark("latent_pass")
		best, scores = pre_best, pre_scores
		if latents_for == "all" and self.clvp is not None and B > 1:
			scores = self.clvp(text_tokens, codes, return_loss=False)
			best = int(torch.argmax(scores))
		latents = trim_calm_tokens(codes[best:best + 1], latents[best:best + 1] if latents_for == "all" else latents)
		T = latents.shape[1] * 4 * 24000 // 22050
		E = diff.timestep_independent(latents, diffusion_latents, T, False)
		noise = torch.randn((1, 100, T), device=dev) * diffusion_temp
		mel = diffuser.sample_loop(diff, (1, 100, T), sampler=diffusion_sampler, noise=noise,
								   model_kwargs={"precomputed_aligned_embeddings": E}, progress=False)
		mark("ddim")
		mels = denormalize_tacotron_mel(mel)[:, :, :T]
		seconds = T * HOP / SAMPLE_RATE
		if return_all:
			return mels, seconds, dict(codes=codes, latents=latents, E=E, noise=noise, mel=mel, scores=scores, best=best)
		return mels, seconds

	@torch.inference_mode()
	def inference_lines(self, lines, autoregressive_latents, diffusion_latents, *, max_ar_steps=500, max_diffusion_steps=80, ar_temp=0.8,
						diffusion_temp=1.0, top_p=1.0, top_k=0, repetition_penalty=1.0, length_penalty=1.0, cond_free=True,
						candidates=1, suppress_tokens=None, ar_batch_lines=None, ddim_batch_lines=None):
		"""The reference's `for line in lines` loop (inference.py:237-422) software-pipelined and batched:
		(1) the autoregressive sampling of up to `ar_batch_lines` consecutive lines runs as ONE decode batch (UnifiedVoice.inference_speech_lines:
		the GPT-2 weights are streamed once per token for all of them; default: as many as fit max_batch, at most 4; 1 = one line per batch);
		(2) the lines sampled together are also DIFFUSED together, as one ragged batch (`ddim_batch_lines`, default: the sampled batch; 1 = line by
		line): each line's mel stays bit for bit its own loop's (SpacedDiffusion.sample_loop_lines) while every GEMM of a step runs over the rows of
		all of them; (3) the diffusion of a batch runs while the sampling of later lines does.  The two phases of different lines are independent and both are latency-bound chains of
		small kernels, so they interleave on the CUs -- provided BOTH keep being fed: the sampling loop needs the host once per token
		(graph replay), and enqueuing a diffusion is ~10k launches from one C call, so the diffusion is issued from a worker thread on its
		own HIP stream (ctypes releases the GIL for the call).  Measured at the benchmark's shape: 422.7 ms per line sequentially, 3 % less
		with both phases issued by one thread, 374.9 ms (-11 %) with the worker; two PROCESSES sharing the GPU reach 1.4x, so launch-path
		contention inside one process still costs.  (Stream priorities make it far worse -- 660 ms with either chain prioritised -- and
		GPU_MAX_HW_QUEUES=8 removes the gain; the default queue mapping is kept.)  Results are identical to calling `inference` per line: every `generate` reseeds the generator to 0
		(stream_generator.py:296), and the draws that follow a line's AR phase in the reference (the diffusion start noise,
		inference.py:404, and DDIM's per-step dummy draws, diffusion.py:685) are made by the main thread in that same order before the next
		line's AR phase starts; the worker draws nothing.
		Batched lines keep that property: each row decodes with its own line's cache length, every line draws the same noise, and before a line's
		own draws the generators are put where its `generate` alone would have left them (UnifiedVoice.position_rng_after_line).
		`lines`: list of [1, Tt] int64 token tensors.  Returns a list of (mels [1, 100, T], seconds, codes)."""
		from concurrent.futures import ThreadPoolExecutor
		ar, diff = self.autoregressive, self.diffusion
		dev = ar.device
		diffuser = get_diffuser(steps=max_diffusion_steps, cond_free=cond_free)
		extra = {"suppress_tokens": suppress_tokens} if suppress_tokens else {}
		main = torch.cuda.current_stream(dev)
		s_ar, s_df = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
		s_ar.wait_stream(main)
		s_df.wait_stream(main)

		def diffuse(group, ready):
			"""the DDIM loops of one sampled batch of lines as ONE ragged batch (SpacedDiffusion.sample_loop_lines: every GEMM of a step runs over
			the rows of all lines, each line's mel bit for bit what its own loop gives); a single line takes the single-line entry"""
			with torch.inference_mode(), torch.cuda.device(dev), torch.cuda.stream(s_df):
				s_df.wait_event(ready)
				Es = [diff.timestep_independent(latents, diffusion_latents, T, False) for latents, _, T in group]
				if len(group) == 1:
					mel = [diffuser.sample_loop(diff, (1, 100, group[0][2]), sampler="ddim", noise=group[0][1],
												model_kwargs={"precomputed_aligned_embeddings": Es[0]}, progress=False, consume_rng=False)]
				else:
					mel = diffuser.sample_loop_lines(diff, [noise for _, noise, _ in group], Es)
				mels = [denormalize_tacotron_mel(m)[:, :, :T] for m, (_, _, T) in zip(mel, group)]
				for (latents, noise, _), E in zip(group, Es):
					for t_ in (latents, noise, E):
						t_.record_stream(s_df)
				return mels

		out, pending = [], []
		lines = [t.to(dev) for t in lines]
		G = ar_batch_lines or max(1, min(4, ar.max_batch // max(candidates, 1)))
		G = max(1, min(G, ar.max_batch // max(candidates, 1)))
		DG = G if ddim_batch_lines is None else max(1, int(ddim_batch_lines))      # lines diffused as one batch (1 = line by line)
		if not cond_free:
			DG = 1                                                                     # the ragged batch is laid out as [cond | uncond]
		with ThreadPoolExecutor(max_workers=1) as pool:          # one worker: diffusions stay in line order on s_df
			group, meta = [], []
			for i, text_tokens in enumerate(lines):
				with torch.cuda.stream(s_ar):
					if i % G == 0:      # sample this line and the next G - 1 as one batch
						batch_codes = ar.inference_speech_lines(autoregressive_latents, lines[i:i + G], do_sample=True, top_k=top_k, top_p=top_p,
																temperature=ar_temp, num_return_sequences=candidates, num_beams=1, length_penalty=length_penalty,
																repetition_penalty=repetition_penalty, max_generate_length=max_ar_steps, **extra)
					ar.position_rng_after_line(i % G)       # the generators as this line's own `generate` leaves them
					codes = fix_stop_tokens(batch_codes[i % G], ar.stop_mel_token)
					B, M = codes.shape
					best = 0                                                # candidate choice as in `inference`
					if self.clvp is not None and B > 1:
						best = int(torch.argmax(self.clvp(text_tokens, codes, return_loss=False)))
					# the dense latent pass on the chosen row only (rows are independent: the bits `inference` gets from the all-candidates pass)
					al_row = autoregressive_latents if autoregressive_latents.shape[0] == 1 else autoregressive_latents[best:best + 1]
					latents = ar.forward(al_row, text_tokens, torch.tensor([text_tokens.shape[1]], dtype=torch.int32), codes[best:best + 1],
										 torch.tensor([M * ar.mel_length_compression]), return_latent=True, clip_inputs=False)
					latents = trim_calm_tokens(codes[best:best + 1], latents)     # host copy of the codes: the AR phase of this line is complete
					T = latents.shape[1] * 4 * 24000 // 22050
					noise = torch.randn((1, 100, T), device=dev) * diffusion_temp
					for _ in range(max_diffusion_steps):                    # DDIM's ignored per-step draws, in reference order
						torch.randn_like(noise)
					group.append((latents, noise, T))
					meta.append((T, codes))
					last_of_batch = (i % G == G - 1) or i == len(lines) - 1
					if len(group) >= DG or last_of_batch:
						ready = torch.cuda.Event()
						ready.record(s_ar)
						pending.append((pool.submit(diffuse, group, ready), meta))
						group, meta = [], []
			for fut, meta in pending:
				for mels, (T, codes) in zip(fut.result(), meta):
					out.append((mels, T * HOP / SAMPLE_RATE, codes))
		main.wait_stream(s_ar)
		main.wait_stream(s_df)
		return out
