"""Mel front-ends of the conditioning path on libttk: the two spectrogram modules `emb/mel.py:50-82` runs on a
reference clip before the conditioning encoders --

  TorchMelSpectrogram   models/arch_utils.py:361-395   AR side: torchaudio MelSpectrogram(n_fft 1024, hop 256, power 2, 80 HTK-spaced
                        slaney-normalised bands, 0-8 kHz, 22.05 kHz), log(clamp 1e-5), divided by the per-band `mel_norms` when given
  TacotronSTFT          models/arch_utils.py:662-700   diffusion side: clip to [-1, 1], hann-windowed DFT magnitudes (STFT :560-623),
                        librosa's Slaney mel basis (100 bands, 0-12 kHz, 24 kHz), log(clamp 1e-5)

Both are "reflect-padded frames x windowed DFT matrix -> |.|^p -> mel matrix -> log" and run as two f32 GEMMs behind `ttk_mel_*`; this
module builds the two matrices on the host in float64.  torchaudio and librosa are absent from this image: their filterbank definitions
(torchaudio.functional.melscale_fbanks, librosa.filters.mel) are restated from their published formulas, and so is the polyphase
windowed-sinc resampler of `torchaudio.functional.resample` (emb/mel.py:67,86), which runs as one FIR kernel (`ttk_resample_fir`).
`format_autoregressive_conditioning` / `format_diffusion_conditioning` / `encode` mirror emb/mel.py:50-109 on tensors (file IO stays outside).
"""
from __future__ import annotations

import ctypes as C
import functools
import math
from typing import Optional

import numpy as np
import torch

from . import _lib


class MelConfigC(C.Structure):
	_fields_ = [(n, C.c_int) for n in ("n_fft", "hop", "n_mels", "power", "clip", "has_norms")]


def hann_periodic(n: int) -> np.ndarray:
	"""scipy.signal.get_window('hann', n, fftbins=True) == torch.hann_window(n, periodic=True)."""
	return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n, dtype=np.float64) / n)


def dft_basis(n_fft: int, window: np.ndarray) -> np.ndarray:
	"""[2 * (n_fft/2 + 1), n_fft]: rows k = Re, rows n_fft/2 + 1 + k = Im of the windowed forward DFT (arch_utils.py:571-590: real and
	imaginary parts of fft(eye) stacked, times the window)."""
	nb = n_fft // 2 + 1
	ang = 2.0 * np.pi * np.outer(np.arange(nb), np.arange(n_fft)) / n_fft
	return np.concatenate([np.cos(ang), -np.sin(ang)], axis=0) * window[None, :]


def melscale_fbanks_htk(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
	"""torchaudio.functional.melscale_fbanks(..., norm="slaney", mel_scale="htk") transposed to [n_mels, n_freqs]: triangles with corners
	equally spaced on m = 2595 log10(1 + f / 700), each scaled by 2 / (its width in Hz)."""
	all_freqs = np.linspace(0, sample_rate // 2, n_freqs)
	m_pts = np.linspace(2595.0 * math.log10(1.0 + f_min / 700.0), 2595.0 * math.log10(1.0 + f_max / 700.0), n_mels + 2)
	f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
	f_diff = f_pts[1:] - f_pts[:-1]
	slopes = f_pts[None, :] - all_freqs[:, None]
	fb = np.maximum(0.0, np.minimum(-slopes[:, :-2] / f_diff[:-1], slopes[:, 2:] / f_diff[1:]))
	fb = fb * (2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels]))[None, :]
	return fb.T.copy()


def _slaney_hz_to_mel(f):
	f = np.asarray(f, dtype=np.float64)
	lin = f / (200.0 / 3)
	return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-300) / 1000.0) / (math.log(6.4) / 27.0), lin)


def _slaney_mel_to_hz(m):
	m = np.asarray(m, dtype=np.float64)
	return np.where(m >= 15.0, 1000.0 * np.exp((math.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))


def mel_basis_slaney(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
	"""librosa.filters.mel(sr=, n_fft=, n_mels=, fmin=, fmax=) with its defaults htk=False, norm='slaney': [n_mels, n_fft/2 + 1]; corners
	equally spaced on the Slaney scale (linear below 1 kHz, logarithmic above), area normalisation."""
	fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
	mel_f = _slaney_mel_to_hz(np.linspace(_slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax), n_mels + 2))
	fdiff = np.diff(mel_f)
	ramps = mel_f[:, None] - fftfreqs[None, :]
	w = np.maximum(0.0, np.minimum(-ramps[:-2] / fdiff[:-1, None], ramps[2:] / fdiff[1:, None]))
	return w * (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]


@functools.lru_cache(maxsize=16)
def sinc_resample_kernel(orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99):
	"""torchaudio.functional.resample's `_get_sinc_resample_kernel` (sinc_interp_hann) in f32, as `functional.resample` computes it for f32
	audio: (kernels [gnew, 2 * width + gorig], width, gorig, gnew) for the gcd-reduced rates."""
	g = math.gcd(int(orig_freq), int(new_freq))
	orig, new = int(orig_freq) // g, int(new_freq) // g
	base_freq = min(orig, new) * rolloff
	width = math.ceil(lowpass_filter_width * orig / base_freq)
	idx = torch.arange(-width, width + orig, dtype=torch.float32)[None, None] / orig
	t = torch.arange(0, -new, -1, dtype=torch.float32)[:, None, None] / new + idx
	t = t * base_freq
	t = t.clamp_(-lowpass_filter_width, lowpass_filter_width)
	window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
	t = t * math.pi
	scale = base_freq / orig
	kernels = torch.where(t == 0, torch.tensor(1.0), t.sin() / t) * window * scale
	return kernels.reshape(new, 2 * width + orig).contiguous(), width, orig, new


@functools.lru_cache(maxsize=16)
def _device_kernels(orig_freq: int, new_freq: int, device: str) -> torch.Tensor:
	return sinc_resample_kernel(orig_freq, new_freq)[0].to(device)


@torch.inference_mode()
def resample(wav: torch.Tensor, orig_freq: int, new_freq: int, device: str = "cuda:0") -> torch.Tensor:
	"""`torchaudio.functional.resample(wav, orig_freq, new_freq)` (defaults: sinc_interp_hann, width 6, rolloff 0.99) on the GPU:
	[..., n] -> [..., ceil(new * n / orig)] f32."""
	if int(orig_freq) == int(new_freq):
		return wav
	dev = torch.device(device)
	if dev.type != "cuda":
		raise _lib.TTKError("tortoise_tts_amd runs on an MI355X only (device must be cuda:N)")
	kernels, width, orig, new = sinc_resample_kernel(orig_freq, new_freq)
	shape = wav.shape
	x = wav.reshape(-1, shape[-1]).to(dev, torch.float32).contiguous()
	b, n = x.shape
	if b == 0 or n == 0:
		raise _lib.TTKError("empty clip")
	n_out = int(math.ceil(new * n / orig))
	out = torch.empty(b, n_out, device=dev, dtype=torch.float32)
	k = _device_kernels(int(orig_freq), int(new_freq), str(dev))
	with torch.cuda.device(dev):
		_lib.check(_lib.load().ttk_resample_fir(x.data_ptr(), b, n, k.data_ptr(), orig, new, width, out.data_ptr(), n_out, _lib.stream_ptr()), "ttk_resample_fir")
	return out.reshape(*shape[:-1], n_out)


def pad_or_truncate(t: torch.Tensor, length: int) -> torch.Tensor:
	"""emb/mel.py:20-29."""
	if t.shape[-1] == length:
		return t
	if t.shape[-1] < length:
		return torch.nn.functional.pad(t, (0, length - t.shape[-1]))
	return t[..., :length]


class _MelFrontEnd:
	def __init__(self, n_fft: int, hop: int, n_mels: int, power: int, clip: bool, mel_matrix: np.ndarray, mel_norms: Optional[torch.Tensor], device: str):
		self.device = torch.device(device)
		if self.device.type != "cuda":
			raise _lib.TTKError("tortoise_tts_amd runs on an MI355X only (device must be cuda:N)")
		self.lib = _lib.load()
		self.n_fft, self.hop_length, self.n_mel_channels = n_fft, hop, n_mels
		sd = {"basis": torch.from_numpy(dft_basis(n_fft, hann_periodic(n_fft))).float(), "mel_basis": torch.from_numpy(mel_matrix).float()}
		if mel_norms is not None:
			if mel_norms.numel() != n_mels:
				raise _lib.TTKError(f"mel_norms has {mel_norms.numel()} entries, expected {n_mels}")
			sd["mel_norms"] = mel_norms.detach().float().reshape(n_mels).cpu()
		names = list(sd.keys())
		views, keep = _lib.weight_views(sd, names)
		c = MelConfigC(n_fft, hop, n_mels, power, int(clip), int(mel_norms is not None))
		self._h = C.c_void_p()
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_mel_create(C.byref(self._h), C.byref(c), views, len(names)), "ttk_mel_create")
		del keep

	def __del__(self):
		h = getattr(self, "_h", None)
		if h:
			self.lib.ttk_mel_destroy(h)
			self._h = None

	def to(self, *a, **k):
		return self

	def eval(self):
		return self

	@torch.inference_mode()
	def _run(self, wav: torch.Tensor) -> torch.Tensor:
		if wav.dim() == 3:
			wav = wav.squeeze(1)
		if wav.dim() != 2:
			raise _lib.TTKError(f"audio must be [b, samples] (or [b, 1, samples]), got {tuple(wav.shape)}")
		b, n = wav.shape
		if b == 0 or n <= self.n_fft // 2:
			raise _lib.TTKError(f"a clip needs more than {self.n_fft // 2} samples (reflect padding), got {n}")
		wav = wav.to(self.device, torch.float32).contiguous()
		out = torch.empty(b, self.n_mel_channels, n // self.hop_length + 1, device=self.device, dtype=torch.float32)
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_mel_forward(self._h, wav.data_ptr(), b, n, out.data_ptr(), _lib.stream_ptr()), "ttk_mel_forward")
		return out


class TorchMelSpectrogram(_MelFrontEnd):
	"""arch_utils.py:361-395.  `mel_norms`: the per-band divisor of `mel_norms.pth` (a download upstream), or None."""

	def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, mel_fmin=0, mel_fmax=8000, sampling_rate=22050,
				 normalize=False, mel_norms: Optional[torch.Tensor] = None, device: str = "cuda:0"):
		if win_length != filter_length or normalize:
			raise NotImplementedError("win_length == filter_length and normalize=False (the reference's only use)")
		fb = melscale_fbanks_htk(filter_length // 2 + 1, float(mel_fmin), float(mel_fmax), n_mel_channels, sampling_rate)
		super().__init__(filter_length, hop_length, n_mel_channels, 2, False, fb, mel_norms, device)

	def forward(self, inp: torch.Tensor) -> torch.Tensor:
		"""[b, samples] or [b, 1, samples] -> log-mel [b, 80, samples // 256 + 1]."""
		return self._run(inp)

	__call__ = forward


class TacotronSTFT(_MelFrontEnd):
	"""arch_utils.py:662-700; `load_model("stft", sr=24000)` builds TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000) (models/__init__.py:147-152)."""

	def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, sampling_rate=22050, mel_fmin=0.0, mel_fmax=8000.0,
				 device: str = "cuda:0"):
		if win_length != filter_length:
			raise NotImplementedError("win_length == filter_length (the reference's only use)")
		fb = mel_basis_slaney(sampling_rate, filter_length, n_mel_channels, float(mel_fmin), float(mel_fmax))
		super().__init__(filter_length, hop_length, n_mel_channels, 1, True, fb, None, device)

	def mel_spectrogram(self, y: torch.Tensor) -> torch.Tensor:
		"""[b, samples] in [-10, 10] (asserted upstream :692-693), clipped to [-1, 1] -> log-mel [b, n_mels, samples // hop + 1]."""
		if y.numel() and (float(y.min()) < -10 or float(y.max()) > 10):
			raise _lib.TTKError("audio outside [-10, 10] (arch_utils.py:692-693 asserts the same)")
		return self._run(y)


def format_autoregressive_conditioning(wav: torch.Tensor, tms: TorchMelSpectrogram, cond_length: int = 132300, rng=None) -> torch.Tensor:
	"""emb/mel.py:50-65: wav [1, n] at 22.05 kHz, padded or randomly cropped to `cond_length` samples (0: whole clip) -> mel [1, 80, F].
	`rng`: a `random.Random` for the crop offset (the reference draws from the global `random`)."""
	import random
	if cond_length > 0:
		gap = wav.shape[-1] - cond_length
		if gap < 0:
			wav = torch.nn.functional.pad(wav, pad=(0, abs(gap)))
		elif gap > 0:
			start = (rng or random).randint(0, gap)
			wav = wav[:, start:start + cond_length]
	return tms(wav.unsqueeze(0)).squeeze(0).unsqueeze(0)


def format_diffusion_conditioning(sample: torch.Tensor, stft: TacotronSTFT) -> torch.Tensor:
	"""emb/mel.py:67-82: wav [1, n] at 22.05 kHz -> 24 kHz, padded / truncated to 102400 samples -> mel [1, 100, 401]."""
	sample = resample(sample, 22050, 24000, device=str(stft.device))
	sample = pad_or_truncate(sample, 102400)
	return stft.mel_spectrogram(sample)


@torch.inference_mode()
def encode(wav: torch.Tensor, sr: int, *, tms: TorchMelSpectrogram, stft: TacotronSTFT, conditioning_encoder, contextual_embedder, rng=None) -> dict:
	"""emb/mel.py:84-109 without the dvae codes (training data, not needed to speak): one mono clip [1, n] at `sr` ->
	{"conds": (ar mel [1, 1, 80, F], diffusion mel [1, 1, 100, F']), "latent": ([1, 1024], [1, 2048]), "metadata": {...}}."""
	n = wav.shape[-1]
	wav = resample(wav, sr, 22050, device=str(tms.device))
	ar_conds = torch.stack([format_autoregressive_conditioning(wav, tms, rng=rng)], dim=1)
	diff_conds = torch.stack([format_diffusion_conditioning(wav, stft)], dim=1)
	return {
		"conds": (ar_conds, diff_conds),
		"latent": (conditioning_encoder.get_conditioning(ar_conds), contextual_embedder.get_conditioning(diff_conds)),
		"metadata": {"original_length": n, "sample_rate": sr, "duration": n / sr},
	}
