"""TTK_F16: the bf16 design with IEEE-half operands (v_mfma_f32_16x16x32_f16) -- the other dtype the reference's autocast region can be given
(inference.py:331, config.py:625-637).  Same comparisons as the bf16 mode against the reference's golden vectors / the f32 oracle, at an
eighth of the bf16 tolerances (11 significand bits against 8); the full-size shapes are covered in tests/test_gpu_bench_shapes.py.  GPU only."""
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import _lib
from tortoise_tts_amd import weights as W
from test_gpu_parity import DEV, _ar_golden_run, make_ar, make_diff, relerr, t

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("K", [64, 1024])
def test_f16_gemm_equals_matmul(K):
	M, N = 333, 256
	g = torch.Generator().manual_seed(K)
	A, Wt = torch.randn(M, K, generator=g).half(), torch.randn(N, K, generator=g).half()
	C = torch.empty(M, N, device=DEV)
	a, w = A.to(DEV).contiguous(), Wt.to(DEV).contiguous()
	_lib.check(_lib.load().ttk_gemm_nt(_lib.TTK_F16, a.data_ptr(), w.data_ptr(), M, N, K, 0.0, None, C.data_ptr(), _lib.stream_ptr()), "ttk_gemm_nt")
	ref = A.double() @ Wt.double().t()
	mag = A.double().abs() @ Wt.double().abs().t()
	assert ((C.cpu().double() - ref).abs() / mag).max().item() < 1e-5       # products of halves are exact in f32: accumulation order only


def test_ar_small_f16_tolerance(golden):
	g = golden("ar_small")
	model, _ = make_ar(W.AR_SMALL, int(g["seed"]), "f16", max_batch=4, max_ctx=64)
	out = _ar_golden_run(model, g, int(g["B"]))
	assert relerr(out["prefill"], g["prefill_logits"]) < 4e-3
	assert relerr(out["decode"], g["decode_logits"]) < 4e-3
	assert relerr(out["latents"], g["latents"]) < 4e-3


def test_diff_small_f16_tolerance(golden):
	from tortoise_tts_amd.diffusion import get_diffuser
	g = golden("diff_small")
	model, _ = make_diff(W.DIFF_SMALL, int(g["seed"]), "f16")
	T = int(g["T"])
	x, ts, Eg = t(g["x"]).to(DEV), t(g["t"]).to(DEV), t(g["E"]).to(DEV)
	assert relerr(model.timestep_independent(t(g["latents"]).to(DEV), t(g["cond"]).to(DEV), T, False), g["E"]) < 4e-3
	assert relerr(model(x, ts, precomputed_aligned_embeddings=Eg), g["y_cond"]) < 7e-3
	mel = get_diffuser(steps=4, cond_free=True).sample_loop(model, (1, 100, T), sampler="ddim", noise=t(g["noise"]).to(DEV),
															model_kwargs={"precomputed_aligned_embeddings": Eg[:1]})
	assert relerr(mel, g["ddim_cf1"]) < 1e-2


def test_f16_sampling_is_repeatable_and_graph_equals_eager():
	"""the token loop in f16: the captured step replays what the eager step computes, and the candidates stay the oracle's for as long as no
	near-tie in the probabilities is resolved differently (checked on the first tokens only: an f16 logit differs from f32 by ~1e-3)"""
	cfg = W.AR_SMALL
	text = torch.randint(1, 255, (1, 9), generator=torch.Generator().manual_seed(1))
	cond = torch.randn(1, cfg.model_dim, generator=torch.Generator().manual_seed(2))
	kw = dict(num_return_sequences=3, max_generate_length=24, temperature=0.8, top_k=0, top_p=1.0, repetition_penalty=1.0)
	outs = []
	for use_graph in (False, True, True):
		model, sd = make_ar(cfg, 11, "f16", max_batch=4, max_ctx=96, use_graph=use_graph)
		with torch.inference_mode():
			outs.append(model.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, **kw))
	assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
	with torch.inference_mode():
		ref = O.inference_speech(O.AROracle(sd, cfg), cond, text, sample_device="cuda", **kw)
	assert torch.equal(outs[0][:, :4].cpu(), ref[:, :4])


def test_other_handles_reject_f16():
	from tortoise_tts_amd.vocoder import BigVGAN
	with pytest.raises(_lib.TTKError, match="bf16"):
		BigVGAN(W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), 0), W.VOC_SMALL, dtype="f16", device=DEV)
