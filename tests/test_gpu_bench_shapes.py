"""Parity at the shapes that are benchmarked and claimed, in the arithmetic they are benchmarked in.

  configs[1]  bf16 (and f32): teacher-forced KV-cached decode of 16 candidates through ctx 68 -> 318 (64 text tokens, 250 mel
              tokens), logits against the oracle every ~50 steps; one conditioned + one conditioning-free network evaluation at
              T = 1088 frames; the final mel of the last 8 steps of the 80-step DDIM schedule.
  configs[3]  the per-GPU shard: 32 candidates, 256 text tokens, cache length up to 750 (logits at ctx ~ 388 and ~ 750), one
              evaluation at T = 2176 frames.
Full-size models, seeded synthetic weights.  The oracle side of the decode tests is `AROracle.teacher_forced_logits`: ONE dense
causal pass over [prefix | start_mel | forced tokens] with the decode path's position rule (the fed-back token k sits at mel
position k + 1, unified_voice.py:213-214); `tests/test_oracle_golden.py::test_dense_teacher_forced_pass_equals_cached_decode`
pins that this equals the oracle's KV-cached steps.  Tolerances (relative L2 unless said otherwise) are the ones of DESIGN.md section 2.
GPU only; calls go through the C ABI."""
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

BF16_LOGITS = 3e-2     # bf16 logits / latents / E vs the f32 oracle
BF16_EVAL = 5e-2       # one bf16 network evaluation
BF16_MEL = 8e-2        # bf16 mel after several DDIM steps
# fp16 operands carry 11 significand bits against bf16's 8: the same comparisons at an eighth of the bf16 bounds
F16 = 0.125
F32_LOGITS_ABS = 1e-3  # f32 mode, full size, long context (5e-4 at ctx <= 12 in test_gpu_parity.py)
F32_EVAL_ABS = 2e-3
F32_MEL_ABS = 5e-3


def gen(seed):
	return torch.Generator().manual_seed(seed)


def maxerr(a, b):
	return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


def relerr(a, b):
	a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
	return ((a - b).norm() / b.norm()).item()


def run_teacher_forced(model, cond, text, toks, steps):
	B, n = toks.shape
	logits = model._prefill(cond.to(DEV), text.to(DEV), B)
	toks = toks.to(DEV)
	out = {}
	if 0 in steps:
		out[0] = logits.clone()
	for k in range(1, n):
		model._decode(toks[:, k - 1].contiguous(), logits)
		if k in steps:
			out[k] = logits.clone()
	torch.cuda.synchronize()
	return torch.stack([out[j] for j in steps], 1)


# ------------------------------------------------------------------------------------------------ AR decode
@pytest.fixture(scope="module")
def ar_sd():
	return W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0)


@pytest.fixture(scope="module")
def cfg1_case(ar_sd):
	"""configs[1]: 64 text tokens, 16 candidates, 250 mel tokens (ctx 68 -> 318)."""
	text = torch.randint(1, 255, (1, 64), generator=gen(1234))
	cond = torch.randn(1, 1024, generator=gen(1235))
	toks = torch.randint(0, 8192, (16, 250), generator=gen(1236))
	steps = [0, 1, 50, 100, 150, 200, 249]
	with torch.inference_mode():
		ref = O.AROracle(ar_sd, W.AR_FULL).teacher_forced_logits(cond, text, toks, steps)
	return text, cond, toks, steps, ref


@pytest.fixture(scope="module")
def cfg3_case(ar_sd):
	"""configs[3] shard: 256 text tokens, 32 candidates, cache length up to 750; oracle on rows of both 16-row MFMA tiles."""
	text = torch.randint(1, 255, (1, 256), generator=gen(2234))
	cond = torch.randn(1, 1024, generator=gen(2235))
	toks = torch.randint(0, 8192, (32, 491), generator=gen(2236))
	steps = [0, 128, 300, 487, 488, 489, 490]         # ctx 260, 388 (past the 384 keys one request round covers), 560, 747..750
	rows = [0, 15, 16, 31]
	with torch.inference_mode():
		ref = O.AROracle(ar_sd, W.AR_FULL).teacher_forced_logits(cond, text, toks[rows], steps)
	return text, cond, toks, steps, rows, ref


@pytest.mark.parametrize("dtype", ["bf16", "f32", "f16"])
def test_config1_teacher_forced_decode_full_size(ar_sd, cfg1_case, dtype):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	text, cond, toks, steps, ref = cfg1_case
	model = UnifiedVoice(ar_sd, W.AR_FULL, dtype=dtype, device=DEV, max_batch=16, max_ctx=64 + 4 + 250 + 8)
	got = run_teacher_forced(model, cond, text, toks, steps)
	for i, j in enumerate(steps):
		if dtype == "f32":
			assert maxerr(got[:, i], ref[:, i]) < F32_LOGITS_ABS, (j, maxerr(got[:, i], ref[:, i]))
		else:
			tol = BF16_LOGITS * (F16 if dtype == "f16" else 1.0)
			assert relerr(got[:, i], ref[:, i]) < tol, (j, relerr(got[:, i], ref[:, i]))
			# per candidate too: one bad row must not hide in the batch norm
			worst = max(relerr(got[b, i], ref[b, i]) for b in range(got.shape[0]))
			assert worst < 2 * tol, (j, worst)


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_config3_shard_teacher_forced_decode_full_size(ar_sd, cfg3_case, dtype):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	text, cond, toks, steps, rows, ref = cfg3_case
	model = UnifiedVoice(ar_sd, W.AR_FULL, dtype=dtype, device=DEV, max_batch=32, max_ctx=256 + 4 + 491 + 8)
	got = run_teacher_forced(model, cond, text, toks, steps)[rows]
	for i, j in enumerate(steps):
		if dtype == "f32":
			assert maxerr(got[:, i], ref[:, i]) < F32_LOGITS_ABS, (j, maxerr(got[:, i], ref[:, i]))
		else:
			assert relerr(got[:, i], ref[:, i]) < BF16_LOGITS, (j, relerr(got[:, i], ref[:, i]))


# ------------------------------------------------------------------------------------------------ diffusion
@pytest.fixture(scope="module")
def diff_sd():
	return W.synth_state_dict(W.diffusion_shapes(W.DIFF_FULL), 1)


@pytest.fixture(scope="module")
def cfg1_diff_case(golden):
	"""configs[1]: 250 mel tokens -> T = 1088 frames.  Expected values come from the REFERENCE run at this size in the build container
	(tests/golden/diff_cfg1.npz, oracle/make_golden.py: diff_cfg1_case): E and the two evaluations on every 8th frame, and the final
	mel of the last 8 steps of the 80-step DDIM schedule started from x as x_8.  Inputs are regenerated from the same seeds."""
	g = golden("diff_cfg1")
	M, T = int(g["M"]), int(g["T"])
	assert T == O.mel_frames_for(M) == 1088
	lat = torch.randn(1, M, 1024, generator=gen(11))
	dcond = torch.randn(1, 2048, generator=gen(12))
	x = torch.randn(1, 100, T, generator=gen(13))
	t = torch.tensor([1500])
	tt = lambda k: torch.from_numpy(g[k])
	return lat, dcond, x, t, int(g["stride"]), tt("E_sub"), tt("y_cond_sub"), tt("y_uncond_sub"), tt("mel")


@pytest.mark.parametrize("dtype", ["bf16", "f32", "f16"])
def test_config1_evaluation_and_ddim_slice_at_T1088(diff_sd, cfg1_diff_case, dtype):
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.diffusion import DiffusionTTS, get_diffuser
	lat, dcond, x, t, st, E_sub, yc_sub, yu_sub, xm = cfg1_diff_case
	T = x.shape[-1]
	model = DiffusionTTS(diff_sd, W.DIFF_FULL, dtype=dtype, device=DEV)
	gE = model.timestep_independent(lat.to(DEV), dcond.to(DEV), T, False)
	# the evaluations and the loop take the oracle's E (f32, equal to the reference's) as their input in both arithmetic modes, so each stage
	# is compared on its own
	with torch.inference_mode():
		E = O.DiffusionOracle(diff_sd, W.DIFF_FULL).timestep_independent(lat, dcond, T).to(DEV)
	assert maxerr(E[:, :, ::st], E_sub) < 1e-4
	gc = model(x.to(DEV), t.to(DEV), precomputed_aligned_embeddings=E)
	gu = model(x.to(DEV), t.to(DEV), conditioning_free=True)
	# last 8 steps of the 80-step schedule through the whole-loop entry (two-stream, cond + cond-free as one batch of 2)
	d = get_diffuser(steps=80, cond_free=True)
	steps = (_lib.StepC * 8)(*[d.step_coefs(i, "ddim") for i in range(8)])
	gx = x.to(DEV).clone()
	Ed = E.contiguous()
	_lib.check(model.lib.ttk_diff_sample_ddim(model._h, gx.data_ptr(), Ed.data_ptr(), 1, T, steps, 8, _lib.stream_ptr()), "ttk_diff_sample_ddim")
	torch.cuda.synchronize()
	if dtype == "f32":
		assert maxerr(gE[:, :, ::st], E_sub) < F32_EVAL_ABS and maxerr(gc[:, :, ::st], yc_sub) < F32_EVAL_ABS and maxerr(gu[:, :, ::st], yu_sub) < F32_EVAL_ABS
		assert maxerr(gx, xm) < F32_MEL_ABS
	else:
		k = F16 if dtype == "f16" else 1.0
		assert relerr(gE[:, :, ::st], E_sub) < k * BF16_LOGITS
		assert relerr(gc[:, :, ::st], yc_sub) < k * BF16_EVAL and relerr(gu[:, :, ::st], yu_sub) < k * BF16_EVAL, (relerr(gc[:, :, ::st], yc_sub), relerr(gu[:, :, ::st], yu_sub))
		assert relerr(gx, xm) < k * BF16_MEL, relerr(gx, xm)
	assert gx.abs().max() <= 1.0 + 1e-5        # step 0: alpha_bar_prev = 1, so the result is the clamped x0


@pytest.fixture(scope="module")
def cfg3_diff_case(diff_sd):
	T = O.mel_frames_for(500)
	assert T == 2176
	x = torch.randn(1, 100, T, generator=gen(21))
	E = torch.randn(1, 1024, T, generator=gen(22))
	t = torch.tensor([3200])
	with torch.inference_mode():
		dor = O.DiffusionOracle(diff_sd, W.DIFF_FULL)
		yc = dor.forward(x, t, E)
		yu = dor.forward(x, t, None, conditioning_free=True)
	return x, E, t, yc, yu


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_config3_evaluation_at_T2176(diff_sd, cfg3_diff_case, dtype):
	"""500 mel tokens -> T = 2176 frames: 34 key tiles, the relative-position bias saturated over most of them, M = 4352 GEMM rows."""
	from tortoise_tts_amd.diffusion import DiffusionTTS
	x, E, t, yc, yu = cfg3_diff_case
	model = DiffusionTTS(diff_sd, W.DIFF_FULL, dtype=dtype, device=DEV)
	gc = model(x.to(DEV), t.to(DEV), precomputed_aligned_embeddings=E.to(DEV))
	gu = model(x.to(DEV), t.to(DEV), conditioning_free=True)
	if dtype == "f32":
		assert maxerr(gc, yc) < F32_EVAL_ABS and maxerr(gu, yu) < F32_EVAL_ABS, (maxerr(gc, yc), maxerr(gu, yu))
	else:
		assert relerr(gc, yc) < BF16_EVAL and relerr(gu, yu) < BF16_EVAL, (relerr(gc, yc), relerr(gu, yu))
