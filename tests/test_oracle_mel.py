"""The mel front-end oracle (oracle/mel_oracle.py): its STFT against the reference's STFT class run in the build container
(tests/golden/stft_ref.npz, arch_utils.py:560-623), the two spectrogram routes against each other, and the host-side matrices of
tortoise_tts_amd/mel.py against the oracle's band-by-band filterbanks."""
import numpy as np
import torch

import mel_oracle as MO
from tortoise_tts_amd import mel as M


def t(a):
	return torch.from_numpy(np.asarray(a))


def test_stft_equals_reference(golden):
	g = golden("stft_ref")
	y, ref = t(g["y"]), t(g["magnitude"])
	got = MO.stft_magnitude(y)
	assert got.shape == ref.shape == (2, 513, 6000 // 256 + 1)
	assert (got - ref).abs().max().item() < 2e-4 * ref.abs().max().item()


def test_conv_route_equals_fft_route():
	"""|conv with the windowed DFT basis|^2 == |torch.stft|^2: the TacotronSTFT and torchaudio routes frame, pad and window alike"""
	g = torch.Generator().manual_seed(3)
	y = torch.randn(2, 5000, generator=g) * 0.3
	a = MO.stft_magnitude(y) ** 2
	b = torch.stft(y, 1024, 256, 1024, window=torch.hann_window(1024), center=True, pad_mode="reflect", return_complex=True).abs() ** 2
	assert a.shape == b.shape and (a - b).abs().max().item() < 1e-3 * b.abs().max().item()


def test_host_matrices_equal_oracle_banks():
	lib = MO.librosa_mel(24000, 1024, 100, 0.0, 12000.0)
	assert np.abs(M.mel_basis_slaney(24000, 1024, 100, 0.0, 12000.0) - lib).max() < 1e-12
	ta = MO.torchaudio_fbanks(513, 0.0, 8000.0, 80, 22050)
	assert np.abs(M.melscale_fbanks_htk(513, 0.0, 8000.0, 80, 22050) - ta).max() < 1e-12
	# Slaney area normalisation: every band integrates to ~1 over frequency (bin width sr / n_fft); bands tile [fmin, fmax]
	area = lib.sum(axis=1) * (24000 / 1024)
	assert np.all(np.abs(area - 1.0) < 0.12) and np.all(np.abs(area[-20:] - 1.0) < 0.01), area     # low bands are a few bins wide
	assert lib[:, 0].sum() == 0 and lib[-1, -1] == 0 and (lib >= 0).all()
	assert np.all(np.diff(lib.argmax(axis=1)) > 0)                               # band centres ascend
	assert ta[:, -1].sum() == 0 and np.count_nonzero(ta[:, 373:].sum(axis=0)) == 0   # nothing above f_max = 8 kHz (bin 372 at 21.5 Hz / bin)
	basis = M.dft_basis(1024, M.hann_periodic(1024))
	assert np.abs(basis - MO.forward_basis(1024)[:, 0].double().numpy()).max() < 1e-6
	assert np.abs(M.hann_periodic(1024) - MO.hann(1024)).max() < 1e-15 and abs(M.hann_periodic(1024) - torch.hann_window(1024, dtype=torch.float64).numpy()).max() < 1e-12


def test_mel_outputs_shapes_and_floor():
	g = torch.Generator().manual_seed(4)
	y = torch.randn(1, 3000, generator=g) * 0.2
	m = MO.tacotron_mel(y)
	assert m.shape == (1, 100, 3000 // 256 + 1) and torch.isfinite(m).all()
	assert abs(MO.tacotron_mel(torch.zeros(1, 2000)).max().item() - np.log(1e-5)) < 1e-6     # silence sits on the clamp: TACOTRON_MEL_MIN, arch_utils.py:533
	big = MO.tacotron_mel(torch.full((1, 2000), 5.0))
	assert torch.equal(big, MO.tacotron_mel(torch.ones(1, 2000)))                            # clipped to [-1, 1] first
	norms = torch.rand(80, generator=g) + 0.5
	a, b = MO.torch_mel_spectrogram(y, None), MO.torch_mel_spectrogram(y, norms)
	assert a.shape == (1, 80, 3000 // 256 + 1) and torch.allclose(a / norms[None, :, None], b)
