"""DDIM loops of several utterances of DIFFERENT length as one batch (include/ttk.h: ttk_diff_sample_ddim_lines).

The reference diffuses one line at a time (inference.py:237-422; its loop is the b = 1 oracle, diffusion.py:768-810): the batch must not change
what any line gets.  Element e of a batch -- a slot of Tp frames holding T_e real ones -- equals its own `sample_loop` run (b = 1, T = T_e) BIT
FOR BIT: attention masks keys / skips query blocks beyond T_e, GroupNorm statistics cover exactly its frames chunked from its first frame
(from the GEMM epilogues where a batch of its own would take them there, from the statistics launch otherwise), the k = 3 convolutions read
zeros beyond its last frame, and every other launch is row-wise.  Checked on the small model (statistics always from the separate launch) and
on the full-size model in bf16 and f32, where lengths that are / are not multiples of the epilogue's 64-row blocks, shorter than one GEMM tile
row and equal to the slot length sit in ONE batch -- and against the oracle, so the single run it is compared with is itself pinned.
GPU only; calls go through the C ABI."""
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def gen(seed):
	return torch.Generator().manual_seed(seed)


def run_both(model, Ts, steps, seed):
	from tortoise_tts_amd.diffusion import get_diffuser
	C = model.cfg.model_channels
	noises = [torch.randn((1, 100, T), generator=gen(seed + i)).to(DEV) for i, T in enumerate(Ts)]
	Es = [torch.randn((1, C, T), generator=gen(seed + 100 + i)).to(DEV) for i, T in enumerate(Ts)]
	d = get_diffuser(steps=steps, cond_free=True)
	single = [d.sample_loop(model, (1, 100, T), sampler="ddim", noise=n, model_kwargs={"precomputed_aligned_embeddings": e}, consume_rng=False)
			  for T, n, e in zip(Ts, noises, Es)]
	batch = d.sample_loop_lines(model, noises, Es)
	torch.cuda.synchronize()
	return single, batch, noises, Es


@pytest.mark.parametrize("Ts", [[30, 17], [64, 64], [129, 40, 200], [5, 300, 64, 1]])
def test_small_model_line_batches_equal_the_single_runs(Ts):
	from tortoise_tts_amd.diffusion import DiffusionTTS
	sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32)
	model = DiffusionTTS(sd, W.DIFF_SMALL, dtype="f32", device=DEV)
	with torch.inference_mode():
		single, batch, noises, Es = run_both(model, Ts, 6, 500)
		for i, (a, b) in enumerate(zip(single, batch)):
			assert a.shape == b.shape == (1, 100, Ts[i]) and torch.isfinite(b).all()
			assert torch.equal(a, b), (Ts, i, (a - b).abs().max().item())
		# ... and the single run is the reference's loop (oracle), so the batch is too
		dor = O.DiffusionOracle(sd, W.DIFF_SMALL)
		ref = O.SpacedSchedule(steps=6, cond_free=True).sample_loop(dor, noises[0].cpu(), Es[0].cpu(), sampler="ddim")
		assert (batch[0].cpu() - ref).abs().max() < 2e-3


@pytest.fixture(scope="module")
def full_sd():
	return W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 0)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_full_size_line_batches_equal_the_single_runs(full_sd, dtype):
	"""1088 (the benchmark's length: 17 blocks of 64, statistics from the GEMM epilogues), 1000 (not a multiple of 64: the statistics launch), 512 (a
	multiple of 64 but, alone, below the row count at which the GEMMs take the tile that emits statistics) and 70 in one batch of four; 3 DDIM steps"""
	from tortoise_tts_amd.diffusion import DiffusionTTS
	model = DiffusionTTS(full_sd, W.DIFF_FULL, dtype=dtype, device=DEV)
	with torch.inference_mode():
		# (784 / 1152: the shortest / longest single runs that take the balanced attention form -- 6 and 9 sixteen-query tiles per workgroup; 1150 / 800: tile
		# counts that do not divide by the workgroups per head, last key tile partly valid)
		for Ts in ([1088, 1000, 512, 70], [1088, 1088], [960, 1088, 1024], [784, 1152], [1150, 800]):
			single, batch, _, _ = run_both(model, Ts, 3, 900)
			for i, (a, b) in enumerate(zip(single, batch)):
				assert torch.isfinite(b).all() and torch.equal(a, b), (dtype, Ts, i, (a - b).abs().max().item())


def test_line_batch_argument_errors(full_sd):
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
	model = DiffusionTTS(W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 32), W.DIFF_SMALL, dtype="f32", device=DEV)
	d = get_diffuser(steps=2, cond_free=True)
	x = torch.zeros((1, 100, 64), device=DEV)
	E = torch.zeros((1, 128, 64), device=DEV)
	steps = (_lib.StepC * 2)(*[d.step_coefs(i, "ddim") for i in range(2)])
	bad = (_lib.C.c_int * 1)(65)
	assert model.lib.ttk_diff_sample_ddim_lines(model._h, x.data_ptr(), E.data_ptr(), 1, 64, bad, steps, 2, None) != 0       # longer than its slot
	ok = (_lib.C.c_int * 1)(64)
	assert model.lib.ttk_diff_sample_ddim_lines(model._h, x.data_ptr(), E.data_ptr(), 1, 60, ok, steps, 2, None) != 0        # slot not a multiple of 64
	with pytest.raises(NotImplementedError):
		get_diffuser(steps=2, cond_free=False).sample_loop_lines(model, [x], [E])
	with pytest.raises(ValueError):
		d.sample_loop_lines(model, [x], [E, E])
