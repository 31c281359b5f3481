"""`TTS` of the reference (inference.py:40-425), inference side, assembled from the libttk-backed parts: text -> tokens, reference clip ->
conditioning latents, then the hot path (AR sampling, latents, [CLVP], diffusion) and the vocoder, per line.  No config system, model
download, engine wrappers or file IO (out of scope) -- the parts are passed in, audio goes in and out as tensors.
"""
from __future__ import annotations

import random
import time
from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import mel as M
from .inference import SAMPLE_RATE, TTSHotPath


def set_seed(seed=None) -> int:
	"""utils/utils.py:124-132."""
	if not seed:
		seed = int(time.time())
	random.seed(seed)
	np.random.seed(seed)
	torch.manual_seed(seed)
	return seed


class TTS:
	def __init__(self, autoregressive, diffusion, tokenizer, *, vocoder=None, clvp=None, conditioning_encoder=None, contextual_embedder=None,
				 tms: Optional[M.TorchMelSpectrogram] = None, stft: Optional[M.TacotronSTFT] = None):
		self.hot = TTSHotPath(autoregressive, diffusion, vocoder=vocoder, clvp=clvp)
		self.tokenizer = tokenizer
		self.conditioning_encoder, self.contextual_embedder, self.tms, self.stft = conditioning_encoder, contextual_embedder, tms, stft
		self.device = autoregressive.device

	def encode_text(self, text: Union[str, torch.Tensor], language: str = "en") -> torch.Tensor:
		"""inference.py:104-111 over `tokenize` (data.py:279-282: a list of pieces is joined first)."""
		if isinstance(text, torch.Tensor):
			return text
		if isinstance(text, list):
			text = "".join(text)
		return torch.tensor(self.tokenizer.encode(text), dtype=torch.int64)

	def encode_audio(self, wav: Union[dict, torch.Tensor, Sequence[torch.Tensor]], sr: int = 22050) -> dict:
		"""inference.py:113-124 over emb/mel.py:84-137 (`encode` / `encode_from_files`): a mono clip [1, n] (or a list of them, concatenated
		in time like `encode_from_files`) -> {"conds", "latent", "metadata"}; a dict produced earlier is passed through."""
		if isinstance(wav, dict):
			return wav
		if any(p is None for p in (self.tms, self.stft, self.conditioning_encoder, self.contextual_embedder)):
			raise ValueError("TTS was built without the conditioning parts (tms, stft, conditioning_encoder, contextual_embedder)")
		if not isinstance(wav, torch.Tensor):
			wav = torch.cat([w[:1] if w.dim() == 2 else w[None] for w in wav], dim=-1)
		if wav.dim() == 1:
			wav = wav[None]
		return M.encode(wav[:1], sr, tms=self.tms, stft=self.stft, conditioning_encoder=self.conditioning_encoder, contextual_embedder=self.contextual_embedder)

	@torch.inference_mode()
	def inference(self, text: str, references, max_ar_steps=500, max_diffusion_steps=80, ar_temp=0.8, diffusion_temp=1.0, top_p=1.0, top_k=0,
				  repetition_penalty=1.0, length_penalty=1.0, beam_width=1, diffusion_sampler="ddim", cond_free=True, vocoder_type="bigvgan",
				  seed=None, candidates=1, references_sr: int = 22050) -> Tuple[torch.Tensor, int]:
		"""inference.py:142-425 (the BigVGAN branch): every line of `text` spoken in the voice of `references` (clip tensor(s), or the dict
		`encode_audio` returns) -> (wav [1, 1, samples] -- the lines concatenated in time -- , 24000)."""
		if vocoder_type != "bigvgan":
			raise NotImplementedError("only the BigVGAN vocoder path is built (the HiFiGAN streaming branch, inference.py:263-320, is not)")
		if beam_width != 1:
			raise NotImplementedError("beam search is not on the inference path (num_beams=1, inference.py:343)")
		if self.hot.vocoder is None:
			raise ValueError("TTS was built without a vocoder")
		ar_latent, diff_latent = self.encode_audio(references, references_sr)["latent"]
		set_seed(seed)
		lines = []
		for line in text.split("\n"):
			tokens = self.encode_text(line).to(self.device)[None]
			if tokens.shape[1] == 0:
				raise ValueError("empty line (the reference fails inside the embedding here)")
			lines.append(tokens)
		kw = dict(max_ar_steps=max_ar_steps, max_diffusion_steps=max_diffusion_steps, ar_temp=ar_temp, diffusion_temp=diffusion_temp, top_p=top_p, top_k=top_k,
				  repetition_penalty=repetition_penalty, length_penalty=length_penalty, cond_free=cond_free, candidates=candidates)
		if len(lines) > 1 and diffusion_sampler == "ddim":
			# several lines: their sampling as one decode batch, the diffusion of a line under the sampling of later ones (TTSHotPath.inference_lines:
			# the same waveforms as the line-by-line loop of inference.py:237-422, which is what the else branch runs)
			wavs = [self.hot.vocoder.inference(mels) for mels, _, _ in self.hot.inference_lines(lines, ar_latent, diff_latent, **kw)]
		else:
			wavs = [self.hot.inference_to_wav(tokens, ar_latent, diff_latent, diffusion_sampler=diffusion_sampler, **kw)[0] for tokens in lines]
		return torch.concat(wavs, dim=-1), SAMPLE_RATE
