"""The role-specialised GEMM instantiations of the DDIM loop (csrc/gemm.hip: GemmRole -- the 1x1 conv, the k = 3 conv + residual, the QKV and the
proj + residual GEMMs at 1024 channels with everything but M and the frame count fixed at compile time) give the GENERIC kernel's bits: same tiles,
same k order, same epilogue operations; only address arithmetic and argument loads differ.  TTK_GEMM_ROLE=0 (read at handle creation) switches the
roles off.  Full-size model: the roles exist for model_channels = 1024 only.  GPU only; calls go through the C ABI."""
import pytest
import torch

from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
	return torch.Generator().manual_seed(seed)


@pytest.fixture(scope="module")
def diff_sd():
	return W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 1)


def build(diff_sd, dtype, roles, monkeypatch):
	from tortoise_tts_amd.diffusion import DiffusionTTS
	monkeypatch.setenv("TTK_GEMM_ROLE", roles)
	return DiffusionTTS(diff_sd, W.DIFF_FULL, dtype=dtype, device=DEV)


@pytest.mark.parametrize("dtype,T", [("bf16", 1088), ("bf16", 1000), ("f16", 1088), ("bf16", 320), ("bf16", 1216), ("bf16", 2176), ("fp8", 1088), ("fp8", 1000)])
def test_role_kernels_equal_the_generic_kernel_bit_for_bit(diff_sd, monkeypatch, dtype, T):
	"""T = 1088: the benchmarked shape (M = 2176: the MIXED grid of k_gemm_mixed -- 256 full 128 x 64 tiles + the 17th tile row as 32 half-height tiles on two
	waves -- for the three statistics roles, 256 x 128 tiles for QKV); T = 1216: M = 2432, 48 half-height tiles; T = 1000: M = 2000 is not a multiple of the
	tile height (guarded epilogue, rows beyond M); T = 320: 128 x 64 tiles for QKV as well; T = 2176 (a configs[3] line): 256 x 128 tiles for every role, statistics included; dtype fp8: the same roles on fp8 operands (tensor scale in the epilogue).  Three DDIM steps each, and one plain evaluation."""
	from tortoise_tts_amd.diffusion import get_diffuser
	noise = torch.randn(1, 100, T, generator=gen(3)).to(DEV)
	E = torch.randn(1, 1024, T, generator=gen(4)).to(DEV)
	t = torch.tensor([900], device=DEV)
	out = {}
	for roles in ("1", "0"):
		m = build(diff_sd, dtype, roles, monkeypatch)
		with torch.inference_mode():
			y = m(noise, t, precomputed_aligned_embeddings=E)
			mel = get_diffuser(steps=3, cond_free=True).sample_loop(m, (1, 100, T), sampler="ddim", noise=noise, model_kwargs={"precomputed_aligned_embeddings": E})
		torch.cuda.synchronize()
		out[roles] = (y, mel)
		del m
	assert torch.isfinite(out["1"][1]).all()
	assert torch.equal(out["1"][0], out["0"][0]) and torch.equal(out["1"][1], out["0"][1])


def test_role_kernels_in_a_ragged_line_batch(diff_sd, monkeypatch):
	"""two lines as one batch (M = 4 x 1088 rows: 128 x 128 tiles for the 1024-wide roles; the shorter line's statistics come from the separate launch)"""
	from tortoise_tts_amd.diffusion import get_diffuser
	Ts = (1088, 1000)
	noises = [torch.randn(1, 100, t, generator=gen(5 + i)).to(DEV) for i, t in enumerate(Ts)]
	Es = [torch.randn(1, 1024, t, generator=gen(7 + i)).to(DEV) for i, t in enumerate(Ts)]
	out = {}
	for roles in ("1", "0"):
		m = build(diff_sd, "bf16", roles, monkeypatch)
		with torch.inference_mode():
			out[roles] = get_diffuser(steps=2, cond_free=True).sample_loop_lines(m, noises, Es)
		torch.cuda.synchronize()
		del m
	for a, b in zip(out["1"], out["0"]):
		assert torch.isfinite(a).all() and torch.equal(a, b)
