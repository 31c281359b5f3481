"""TEST INFRASTRUCTURE -- CPU restatement of the two mel front-ends of the conditioning path.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path never does.

  stft_magnitude          STFT.transform                 models/arch_utils.py:560-623  (conv1d with the windowed DFT basis, as the reference)
  tacotron_mel            TacotronSTFT.mel_spectrogram   models/arch_utils.py:662-700
  torch_mel_spectrogram   TorchMelSpectrogram.forward    models/arch_utils.py:361-395  (torchaudio MelSpectrogram = torch.stft power + melscale_fbanks)
  librosa_mel / torchaudio_fbanks   the two filterbank definitions, written band by band

Pinning: stft_magnitude is pinned by tests/golden/stft_ref.npz, produced by the reference's own STFT class.  The filterbanks belong to
librosa 0.x `filters.mel` and torchaudio `functional.melscale_fbanks`, both ABSENT from this image (no version is pinned by the reference's
setup.py either): they are restated from the published definitions and are **parity unpinned**; tests cross-check the two spectrogram
routes (conv basis vs torch.stft) and known properties of the banks (Slaney area normalisation, band edges).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor


def hann(n: int) -> np.ndarray:
	"""scipy.signal.get_window('hann', n, fftbins=True)."""
	return np.array([0.5 - 0.5 * math.cos(2 * math.pi * i / n) for i in range(n)])


def forward_basis(n_fft: int) -> Tensor:
	"""arch_utils.py:571-590: real and imaginary halves of fft(eye(n_fft))[:n_fft/2 + 1], windowed, as conv filters [2 * nb, 1, n_fft]."""
	fb = np.fft.fft(np.eye(n_fft))
	cut = n_fft // 2 + 1
	fb = np.vstack([np.real(fb[:cut]), np.imag(fb[:cut])])
	return torch.FloatTensor(fb[:, None, :]) * torch.from_numpy(hann(n_fft)).float()


def stft_magnitude(y: Tensor, n_fft: int = 1024, hop: int = 256) -> Tensor:
	"""[b, n] -> [b, n_fft/2 + 1, n // hop + 1]."""
	x = F.pad(y[:, None, None, :], (n_fft // 2, n_fft // 2, 0, 0), mode="reflect").squeeze(1)
	t = F.conv1d(x, forward_basis(n_fft), stride=hop)
	cut = n_fft // 2 + 1
	return torch.sqrt(t[:, :cut] ** 2 + t[:, cut:] ** 2)


def _slaney_hz_to_mel(f: float) -> float:
	return f / (200.0 / 3) if f < 1000.0 else 15.0 + math.log(f / 1000.0) / (math.log(6.4) / 27.0)


def _slaney_mel_to_hz(m: float) -> float:
	return m * (200.0 / 3) if m < 15.0 else 1000.0 * math.exp((math.log(6.4) / 27.0) * (m - 15.0))


def librosa_mel(sr: int, n_fft: int, n_mels: int, fmin: float, fmax: float) -> np.ndarray:
	"""librosa.filters.mel defaults (htk=False, norm='slaney'), one band at a time."""
	nb = n_fft // 2 + 1
	freqs = [i * (sr / 2.0) / (nb - 1) for i in range(nb)]
	lo, hi = _slaney_hz_to_mel(fmin), _slaney_hz_to_mel(fmax)
	edges = [_slaney_mel_to_hz(lo + (hi - lo) * i / (n_mels + 1)) for i in range(n_mels + 2)]
	w = np.zeros((n_mels, nb))
	for m in range(n_mels):
		left, centre, right = edges[m], edges[m + 1], edges[m + 2]
		for k, f in enumerate(freqs):
			w[m, k] = max(0.0, min((f - left) / (centre - left), (right - f) / (right - centre))) * 2.0 / (right - left)
	return w


def torchaudio_fbanks(n_freqs: int, f_min: float, f_max: float, n_mels: int, sample_rate: int) -> np.ndarray:
	"""torchaudio.functional.melscale_fbanks(norm='slaney', mel_scale='htk') as [n_mels, n_freqs], one band at a time."""
	freqs = [i * (sample_rate // 2) / (n_freqs - 1) for i in range(n_freqs)]
	lo, hi = 2595.0 * math.log10(1.0 + f_min / 700.0), 2595.0 * math.log10(1.0 + f_max / 700.0)
	edges = [700.0 * (10.0 ** ((lo + (hi - lo) * i / (n_mels + 1)) / 2595.0) - 1.0) for i in range(n_mels + 2)]
	w = np.zeros((n_mels, n_freqs))
	for m in range(n_mels):
		left, centre, right = edges[m], edges[m + 1], edges[m + 2]
		for k, f in enumerate(freqs):
			w[m, k] = max(0.0, min((f - left) / (centre - left), (right - f) / (right - centre))) * 2.0 / (right - left)
	return w


def tacotron_mel(y: Tensor, n_fft=1024, hop=256, n_mels=100, sr=24000, fmin=0.0, fmax=12000.0) -> Tensor:
	"""arch_utils.py:691-700: clip, magnitudes, mel_basis @ magnitudes, log(clamp 1e-5)."""
	mag = stft_magnitude(torch.clip(y, min=-1, max=1), n_fft, hop)
	mel = torch.matmul(torch.from_numpy(librosa_mel(sr, n_fft, n_mels, fmin, fmax)).float(), mag)
	return torch.log(torch.clamp(mel, min=1e-5))


def torch_mel_spectrogram(wav: Tensor, mel_norms: Tensor = None, n_fft=1024, hop=256, n_mels=80, sr=22050, fmin=0.0, fmax=8000.0) -> Tensor:
	"""arch_utils.py:384-395 with torchaudio's Spectrogram(power=2, center=True, reflect, periodic hann) + MelScale."""
	spec = torch.stft(wav, n_fft, hop, n_fft, window=torch.hann_window(n_fft), center=True, pad_mode="reflect", normalized=False, onesided=True,
					  return_complex=True).abs().pow(2.0)
	mel = torch.matmul(torch.from_numpy(torchaudio_fbanks(n_fft // 2 + 1, fmin, fmax, n_mels, sr)).float(), spec)
	mel = torch.log(torch.clamp(mel, min=1e-5))
	if mel_norms is not None:
		mel = mel / mel_norms[None, :, None]
	return mel


def resample(waveform: Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> Tensor:
	"""torchaudio.functional.resample (sinc_interp_hann), restated from its published source layout: gcd-reduced rates, windowed-sinc
	kernel bank [new, 1, 2 * width + orig], zero padding (width, width + orig), conv1d with stride orig, interleave, crop to
	ceil(new * n / orig).  torchaudio is absent here: **parity unpinned**; tests check it against exact band-limited interpolation."""
	g = math.gcd(int(orig_freq), int(new_freq))
	orig, new = int(orig_freq) // g, int(new_freq) // g
	base_freq = min(orig, new) * rolloff
	width = math.ceil(lowpass_filter_width * orig / base_freq)
	rows = []
	for p in range(new):
		row = []
		for j in range(-width, width + orig):
			t = (-p / new + j / orig) * base_freq
			t = max(-lowpass_filter_width, min(lowpass_filter_width, t))
			win = math.cos(t * math.pi / lowpass_filter_width / 2) ** 2
			x = t * math.pi
			row.append((1.0 if x == 0 else math.sin(x) / x) * win * base_freq / orig)
		rows.append(row)
	kernel = torch.tensor(rows, dtype=torch.float64)[:, None, :]
	shape = waveform.shape
	w = waveform.reshape(-1, shape[-1]).double()
	n = w.shape[-1]
	w = F.pad(w, (width, width + orig))
	out = F.conv1d(w[:, None], kernel, stride=orig).transpose(1, 2).reshape(w.shape[0], -1)
	return out[..., : int(math.ceil(new * n / orig))].float().reshape(*shape[:-1], -1)
