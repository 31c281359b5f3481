"""Mel front-ends of the conditioning path on libttk: the two spectrogram modules `emb/mel.py:50-82` runs on a
reference clip before the conditioning encoders --

  TorchMelSpectrogram   models/arch_utils.py:361-395   AR side: torchaudio MelSpectrogram(n_fft 1024, hop 256, power 2, 80 HTK-spaced
                        slaney-normalised bands, 0-8 kHz, 22.05 kHz), log(clamp 1e-5), divided by the per-band `mel_norms` when given
  TacotronSTFT          models/arch_utils.py:662-700   diffusion side: clip to [-1, 1], hann-windowed DFT magnitudes (STFT :560-623),
                        librosa's Slaney mel basis (100 bands, 0-12 kHz, 24 kHz), log(clamp 1e-5)

Both are "reflect-padded frames x windowed DFT matrix -> |.|^p -> mel matrix -> log" and run as two f32 GEMMs behind `ttk_mel_*`; this
module builds the two matrices on the host in float64.  torchaudio and librosa are absent from this image: their filterbank definitions
(torchaudio.functional.melscale_fbanks, librosa.filters.mel) are restated from their published formulas.  Resampling
(`torchaudio.functional.resample`, emb/mel.py:67,86) is not provided: callers hand in 22.05 kHz / 24 kHz audio.
"""
from __future__ import annotations

import ctypes as C
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
