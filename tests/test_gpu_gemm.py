"""The dense GEMM kernel (csrc/gemm.hip) on caller-provided operands through `ttk_gemm_nt`, against a plain PyTorch f32 matmul of the same
operands: exact-f32 mode, bf16 operands, and fp8-e4m3 operands on the fp8 MFMA.  The operands are exactly representable and every product
is exact in f32, so f32 and bf16 differ from the f64 reference by f32 accumulation only: 1e-5 relative to the row's magnitude sum|a||w|.
v_mfma_f32_16x16x32_fp8_fp8 aligns the 32 products of a step with fewer bits than an f32 adder keeps (observed 4.4e-5 of sum|a||w| with
+-448 operands in the row): stated bound 2e-4."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
F32, BF16, FP8 = 0, 1, 3


def gemm(dtype, A, Wt, out_scale=0.0, bias=None):
	from tortoise_tts_amd import _lib
	lib = _lib.load()
	M, K = A.shape
	N = Wt.shape[0]
	out = torch.full((M, N), float("nan"), device=DEV, dtype=torch.float32)
	_lib.check(lib.ttk_gemm_nt(dtype, A.data_ptr(), Wt.data_ptr(), M, N, K, C.c_float(out_scale), bias.data_ptr() if bias is not None else None,
							   out.data_ptr(), _lib.stream_ptr()), "ttk_gemm_nt")
	torch.cuda.synchronize()
	return out


@pytest.mark.parametrize("M,N,K", [(1, 128, 128), (77, 256, 384), (2176, 1024, 1024), (300, 3072, 1024), (5000, 128, 256)])
def test_fp8_gemm_equals_matmul_of_the_decoded_operands(M, N, K):
	g = torch.Generator().manual_seed(M + N + K)
	a8 = (torch.randn(M, K, generator=g) * 1.5).to(torch.float8_e4m3fn)
	w8 = (torch.randn(N, K, generator=g) * 20).to(torch.float8_e4m3fn)       # as stored: weight / scale, up to +-448
	a8[0, :8] = torch.tensor([0.0, -0.0, 448.0, -448.0, 2.0 ** -9, -2.0 ** -9, 2.0 ** -6, 1.0]).to(torch.float8_e4m3fn)   # zeros, extremes, subnormals
	bias = torch.randn(N, generator=g)
	scale = 2.0 ** -7
	got = gemm(FP8, a8.to(DEV), w8.to(DEV), scale, bias.to(DEV)).cpu()
	ref = (a8.double() @ w8.double().t()) * scale + bias.double()
	mag = (a8.double().abs() @ w8.double().abs().t()) * scale + bias.abs().double()
	assert torch.isfinite(got).all()
	assert ((got.double() - ref).abs() / mag.clamp_min(1e-30)).max().item() < 2e-4


@pytest.mark.parametrize("dtype,K", [(BF16, 64), (BF16, 1024), (F32, 32), (F32, 1024)])
def test_bf16_and_f32_gemm_equal_matmul(dtype, K):
	M, N = 333, 256
	g = torch.Generator().manual_seed(K)
	A, Wt = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
	if dtype == BF16:
		A, Wt = A.bfloat16(), Wt.bfloat16()
	got = gemm(dtype, A.to(DEV).contiguous(), Wt.to(DEV).contiguous()).cpu()
	ref = A.double() @ Wt.double().t()
	mag = A.double().abs() @ Wt.double().abs().t()
	assert ((got.double() - ref).abs() / mag).max().item() < 1e-5


def test_argument_errors():
	from tortoise_tts_amd import _lib
	a = torch.zeros(4, 128, device=DEV)
	with pytest.raises(_lib.TTKError, match="N % 128"):
		gemm(F32, a, torch.zeros(100, 128, device=DEV))
	with pytest.raises(_lib.TTKError, match="K % 128"):
		gemm(FP8, torch.zeros(4, 64, device=DEV, dtype=torch.uint8), torch.zeros(128, 64, device=DEV, dtype=torch.uint8))
	with pytest.raises(_lib.TTKError, match="dtype"):
		gemm(2, a, torch.zeros(128, 128, device=DEV))
