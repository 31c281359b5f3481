"""The compile-time-specialised decode launches (csrc/gemv.hip: k_gemv) against the generic kernel they replace at the benchmarked
geometry (csrc/skinny.hip: k_skinny, selected with TTK_AR_LEAN=0): same operands, same token loop, full-size model.

  f32   the two kernel families run the same operations in the same order (same fragments, same k order per wave, same wave order in the
        cross-wave sum, same folded-LayerNorm statistics): logits, appended K / V rows and sampled ids are equal BIT FOR BIT.
  bf16  the projections and the head are still the same operations; the folded launches sum a lane's row statistics with
        v_dot2c_f32_bf16 instead of an unpack / add / fma chain (a few ulp of mean and rstd), which moves some gelu outputs to the
        neighbouring bf16 value: logits within 3e-3 relative L2 of the generic kernel's (the bf16 bar against the f32 oracle is 3e-2).
  rows  M = 1, 16, 17 (second m-tile, rows 17..31 padding), 48 (third): each row's logits do not depend on what else is in the batch.
GPU only; calls go through the C ABI."""
import os

import pytest
import torch

from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def relerr(a, b):
	a, b = a.double().cpu(), b.double().cpu()
	return ((a - b).norm() / b.norm()).item()


@pytest.fixture(scope="module")
def ar_sd():
	return W.synth_state_dict(W.ar_shapes(W.AR_FULL), 0)


def build(ar_sd, dtype, lean, max_batch=16, max_ctx=64 + 4 + 40):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	old = os.environ.get("TTK_AR_LEAN")
	os.environ["TTK_AR_LEAN"] = "1" if lean else "0"       # read by ttk_ar_create
	try:
		return UnifiedVoice(ar_sd, W.AR_FULL, dtype=dtype, device=DEV, max_batch=max_batch, max_ctx=max_ctx)
	finally:
		if old is None:
			os.environ.pop("TTK_AR_LEAN")
		else:
			os.environ["TTK_AR_LEAN"] = old


def forced(model, cond, text, toks):
	B, n = toks.shape
	logits = model._prefill(cond.to(DEV), text.to(DEV), B)
	out = [logits.clone()]
	toks = toks.to(DEV)
	for k in range(1, n):
		model._decode(toks[:, k - 1].contiguous(), logits)
		out.append(logits.clone())
	torch.cuda.synchronize()
	return torch.stack(out, 1)


@pytest.mark.parametrize("dtype,B", [("f32", 16), ("f32", 5), ("bf16", 16), ("f16", 16), ("bf16", 48), ("fp8w", 16)])
def test_lean_decode_launches_equal_the_generic_kernel(ar_sd, dtype, B):
	g = torch.Generator().manual_seed(77)
	text = torch.randint(1, 255, (1, 64), generator=g)
	cond = torch.randn(1, 1024, generator=g)
	toks = torch.randint(0, 8192, (B, 12), generator=g)
	with torch.inference_mode():
		a = forced(build(ar_sd, dtype, True, max_batch=B), cond, text, toks)
		b = forced(build(ar_sd, dtype, False, max_batch=B), cond, text, toks)
	assert torch.isfinite(a).all()
	if dtype == "f32":
		assert torch.equal(a, b), (a - b).abs().max().item()
	else:
		assert torch.equal(a[:, 0], b[:, 0])                 # the prefill does not run the decode kernels
		for k in range(1, a.shape[1]):
			assert relerr(a[:, k], b[:, k]) < 3e-3, (k, relerr(a[:, k], b[:, k]))


def test_lean_sampling_loop_equals_the_generic_kernel_in_f32(ar_sd):
	"""the whole captured token loop (graph, head-drawn noise, fused sampler) on either kernel family: equal ids, equal generator position"""
	g = torch.Generator().manual_seed(78)
	text = torch.randint(1, 255, (1, 40), generator=g).to(DEV)
	cond = torch.randn(1, 1024, generator=g).to(DEV)
	kw = dict(do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16, max_generate_length=24)
	with torch.inference_mode():
		ma, mb = build(ar_sd, "f32", True), build(ar_sd, "f32", False)
		ia = ma.inference_speech(cond, text, **kw)
		off_a = torch.cuda.default_generators[0].get_offset()
		ib = mb.inference_speech(cond, text, **kw)
		off_b = torch.cuda.default_generators[0].get_offset()
	assert torch.equal(ia, ib) and off_a == off_b


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_rows_do_not_depend_on_the_batch_they_ride_in(ar_sd, dtype):
	"""M = 1 / 16 / 17 / 32 (f32: the largest) / 48 candidates fed the same tokens: row 0's logits are the same bits in every batch size (one
	m-tile, two with padding rows, three) -- what candidate shards and line batches rely on"""
	g = torch.Generator().manual_seed(79)
	text = torch.randint(1, 255, (1, 30), generator=g)
	cond = torch.randn(1, 1024, generator=g)
	row = torch.randint(0, 8192, (1, 8), generator=g)
	sizes = [1, 16, 17, 32] + ([48] if dtype != "f32" else [])
	ref = None
	with torch.inference_mode():
		for B in sizes:
			other = torch.randint(0, 8192, (B, 8), generator=g)
			other[0] = row[0]
			out = forced(build(ar_sd, dtype, True, max_batch=B, max_ctx=30 + 4 + 16), cond, text, other)[0]
			if ref is None:
				ref = out
			else:
				assert torch.equal(out, ref), (B, (out - ref).abs().max().item())


def _with_env(name, value, fn):
	old = os.environ.get(name)
	os.environ[name] = value
	try:
		return fn()
	finally:
		if old is None:
			os.environ.pop(name)
		else:
			os.environ[name] = old


@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_folded_layernorm_on_outlier_channels_and_on_a_common_offset(ar_sd, dtype):
	"""the folded launches multiply the UN-normalised residual row (T-typed) and finish the norm in the epilogue, so their rounding
	error scales with |x|, not |x - mean|.  (a) outlier channels -- what trained GPT-2 streams have: a few channels hundreds of times the rest --
	inflate the row's std with them, the normalised value carries the same relative error either way: fold == LayerNorm-prologue form within the
	16-bit bar.  (b) a COMMON offset of 40 std is what the fold cannot carry: the launch says so (ttk_ar_health -> RuntimeWarning) instead of
	drifting silently, and the prologue form (TTK_AR_LNFOLD=0) still matches the f32 arithmetic."""
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	import warnings
	g = torch.Generator().manual_seed(91)
	text = torch.randint(1, 255, (1, 20), generator=g)
	cond = torch.randn(1, 1024, generator=g)
	toks = torch.randint(0, 8192, (16, 6), generator=g)
	mk = lambda sd, dt, fold: _with_env("TTK_AR_LNFOLD", "1" if fold else "0", lambda: UnifiedVoice(sd, W.AR_FULL, dtype=dt, device=DEV, max_batch=16, max_ctx=20 + 4 + 16))
	tol = 3e-2 if dtype == "bf16" else 4e-3
	# (a) outlier channels in the embeddings the decode rows are built from
	sd = dict(ar_sd)
	emb = sd["mel_embedding.weight"].clone()
	scale = emb.std().item()
	emb[:, 7] += 300 * scale
	emb[:, 500] -= 150 * scale
	sd["mel_embedding.weight"] = emb
	with torch.inference_mode(), warnings.catch_warnings():
		warnings.simplefilter("error")                                   # no health warning on this stream
		ref = forced(mk(sd, "f32", False), cond, text, toks)
		a = forced(mk(sd, dtype, True), cond, text, toks)
		b = forced(mk(sd, dtype, False), cond, text, toks)
		m = mk(sd, dtype, True)
		m.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16, max_generate_length=6)
		assert m.last_health == 0
	for k in range(1, a.shape[1]):
		assert relerr(a[:, k], ref[:, k]) < tol and relerr(b[:, k], ref[:, k]) < tol, (k, relerr(a[:, k], ref[:, k]), relerr(b[:, k], ref[:, k]))
	# (b) a common offset: every channel of every mel embedding row shifted by 40 std
	sd = dict(ar_sd)
	sd["mel_embedding.weight"] = ar_sd["mel_embedding.weight"] + 40 * scale
	with torch.inference_mode():
		ref = forced(mk(sd, "f32", False), cond, text, toks)
		b = forced(mk(sd, dtype, False), cond, text, toks)
		assert relerr(b[:, 1], ref[:, 1]) < tol                           # the prologue form normalises in f32 first: unaffected
		m = mk(sd, dtype, True)
		with pytest.warns(RuntimeWarning, match="TTK_AR_LNFOLD=0"):
			m.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, temperature=0.8, top_k=0, num_return_sequences=16, max_generate_length=4)
		assert m.last_health & 1
