"""Mel front-ends on libttk against the oracle (oracle/mel_oracle.py; its STFT is pinned by the reference's STFT
class, tests/golden/stft_ref.npz) and, through a magnitude-only probe, against that fixture directly.  GPU only; through `ttk_mel_*`.
Tolerance: |log-mel difference| < 2e-3 on bands whose energy is above 1e-4 of the clip's peak band (two f32 DFTs of 1024 points; bands
at the noise floor of the f32 sum are compared through the linear mel energy, relative to the peak, < 1e-5)."""
import numpy as np
import pytest
import torch

import mel_oracle as MO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def t(a):
	return torch.from_numpy(np.asarray(a))


def clip(seed, b, n, sr):
	g = torch.Generator().manual_seed(seed)
	tt = torch.arange(n) / sr
	y = 0.3 * torch.sin(2 * np.pi * 220 * tt)[None] + 0.1 * torch.sin(2 * np.pi * 3100 * tt)[None] + 0.05 * torch.randn(b, n, generator=g)
	return y * torch.linspace(0.2, 1.0, n)[None]


def check(got, ref):
	assert got.shape == ref.shape and got.dtype == torch.float32
	lin_g, lin_r = got.double().exp(), ref.double().exp()
	peak = lin_r.max()
	assert ((lin_g - lin_r).abs().max() / peak).item() < 1e-5
	loud = lin_r > 1e-4 * peak
	assert loud.float().mean().item() > 0.05
	assert (got - ref)[loud].abs().max().item() < 2e-3


@pytest.mark.parametrize("b,n", [(1, 513), (2, 6000), (1, 102400), (3, 132300)])
def test_tacotron_stft_vs_oracle(b, n):
	from tortoise_tts_amd.mel import TacotronSTFT
	y = clip(n + b, b, n, 24000.0)
	fe = TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000, device=DEV)
	got = fe.mel_spectrogram(y.to(DEV)).cpu()
	check(got, MO.tacotron_mel(y))
	assert got.shape[-1] == n // 256 + 1


@pytest.mark.parametrize("b,n,norms", [(1, 513, False), (2, 132300, True), (1, 40000, False)])
def test_torch_mel_spectrogram_vs_oracle(b, n, norms):
	from tortoise_tts_amd.mel import TorchMelSpectrogram
	y = clip(n + 7, b, n, 22050.0)
	mn = (torch.rand(80, generator=torch.Generator().manual_seed(1)) * 2 + 0.5) if norms else None
	fe = TorchMelSpectrogram(mel_norms=mn, device=DEV)
	got = fe(y[:, None, :].to(DEV)).cpu()                        # [b, 1, n] form, squeezed like arch_utils.py:385-386
	ref = MO.torch_mel_spectrogram(y, mn)
	if norms:
		got, ref = got * mn[None, :, None], ref * mn[None, :, None]
	check(got, ref)


def test_magnitudes_equal_reference_stft(golden):
	"""an identity 'mel' matrix turns the handle into the bare STFT: its magnitudes against the reference's STFT.transform fixture"""
	import ctypes as C
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.mel import MelConfigC, dft_basis, hann_periodic
	g = golden("stft_ref")
	y, ref = t(g["y"]), t(g["magnitude"])
	lib = _lib.load()
	sd = {"basis": torch.from_numpy(dft_basis(1024, hann_periodic(1024))).float(), "mel_basis": torch.eye(513)}
	names = list(sd)
	views, keep = _lib.weight_views(sd, names)
	h = C.c_void_p()
	cfg = MelConfigC(1024, 256, 513, 1, 0, 0)
	_lib.check(lib.ttk_mel_create(C.byref(h), C.byref(cfg), views, 2), "ttk_mel_create")
	out = torch.empty(2, 513, ref.shape[-1], device=DEV)
	yd = y.to(DEV).contiguous()
	_lib.check(lib.ttk_mel_forward(h, yd.data_ptr(), 2, y.shape[1], out.data_ptr(), _lib.stream_ptr()), "ttk_mel_forward")
	mag = out.exp().cpu()
	lib.ttk_mel_destroy(h)
	floor = ref.clamp_min(1e-5)                                    # the handle always clamps at 1e-5 before the log
	assert (mag - floor).abs().max().item() < 2e-4 * ref.abs().max().item()


def test_silence_clipping_and_errors():
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.mel import TacotronSTFT
	fe = TacotronSTFT(1024, 256, 1024, 100, 24000, 0, 12000, device=DEV)
	z = fe.mel_spectrogram(torch.zeros(1, 4096).to(DEV))
	assert abs(z.max().item() - np.log(1e-5)) < 1e-6 and abs(z.min().item() - np.log(1e-5)) < 1e-6
	assert torch.equal(fe.mel_spectrogram(torch.full((1, 4096), 5.0).to(DEV)), fe.mel_spectrogram(torch.ones(1, 4096).to(DEV)))
	with pytest.raises(_lib.TTKError, match="more than 512 samples"):
		fe.mel_spectrogram(torch.zeros(1, 512))
	with pytest.raises(_lib.TTKError, match=r"\[-10, 10\]"):
		fe.mel_spectrogram(torch.full((1, 4096), 11.0))
	with pytest.raises(_lib.TTKError, match=r"\[b, samples\]"):
		fe.mel_spectrogram(torch.zeros(4096))
