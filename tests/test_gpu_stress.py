"""The HIP path in the numerical regime of a trained checkpoint.

tests/golden/stress_*.npz were produced by the REFERENCE's classes (oracle/make_golden.py: stress_*_case) on `tortoise_tts_amd.weights.stress_*` weights:
  AR         `mel_head` x 8 and the q / k columns of every `c_attn` x 4 -> logits of std 8 (median max token probability 0.6-0.8 at T = 0.8), attention rows
             close to one-hot (`peaked`); plus two channels of the embedding tables x 300 -> massive activations in fixed residual channels (`outlier`).
  diffusion  q / k rows of every `qkv` x 4, `relative_attention_bias` x 10 -> scores of +-90 (small) / +-113 (full size), softmax rows of median top weight
             0.9998 / 0.25; `emb_layers` x 4; the `in_layers` conv rows of one GroupNorm group x 1e-2 with bias 3.0 -> a group of 3.0 +- 1e-2.
What is asserted: ids bit for bit in f32 (free-running, the fused sampler with `hf_exact_top_p` off AND on, under the CLI's warpers and the reference's
TypicalLogitsWarper), the kernel's kept sets on the reference's own peaked rows, logits / latents / E / evaluations / DDIM x per arithmetic mode with the bounds
of DESIGN.md section 2 ("stress" table), and the folded-LayerNorm health word staying silent (outlier CHANNELS are carried; only a common offset is not).
Measured values go to gpurun_out/stress_errors.json.  GPU only; calls go through the C ABI."""
import json
import os
import warnings

import numpy as np
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ALL_MODES = os.environ.get("TTK_TEST_ALL_MODES") == "1"       # the builder's runs sweep every arithmetic mode; the default run keeps f32 + bf16 (+ the cheap small-model ones)
COLS = torch.cat([torch.arange(0, 96), torch.arange(8100, 8194)])


def t(a):
	return torch.from_numpy(np.asarray(a))


def gen(seed):
	return torch.Generator().manual_seed(seed)


def maxerr(a, b):
	return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


def relerr(a, b):
	a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
	return ((a - b).norm() / b.norm()).item()


def record(name, values):
	path = os.path.join(ROOT, "gpurun_out", "stress_errors.json")
	try:
		os.makedirs(os.path.dirname(path), exist_ok=True)
		data = json.load(open(path)) if os.path.exists(path) else {}
		data[name] = values
		json.dump(data, open(path, "w"), indent=1, sort_keys=True)
	except OSError:
		pass


def ar_sd(variant, cfg=W.AR_SMALL, seed=14):
	return W.stress_ar(W.synth_state_dict(W.ar_shapes(cfg), seed), cfg, variant)


def build_ar(sd, cfg, dtype, **kw):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	return UnifiedVoice(sd, cfg, dtype=dtype, device=DEV, **kw)


# ------------------------------------------------------------------------------------------------ AR: logits and latents per arithmetic mode
# f32: absolute, on logits of std 8 (|logit| to 35).  16-bit / fp8w: relative L2 against the reference's f32 values.
# Measured on MI355X (round 5, profiles/r05_stress_errors.json -> DESIGN.md section 2), worst of the two variants and of prefill / decode / the reference's own sampled sequence:
#   f32  logits 1.2e-4 abs, latents 6.3e-6 abs        f16  logits 2.2e-3, latents 1.4e-3        bf16 logits 1.7e-2, latents 1.7e-2        fp8w logits 1.7e-1, latents 9.3e-2
# Bounds = 3-4x the measurement.  Independent criteria: (a) bf16 -- the REFERENCE'S OWN deviation under the autocast(bf16) region of inference.py:331, stored in the fixture
# (peaked: prefill 8.1e-3, decode 9.4e-3; outlier 2.4e-3 / 2.9e-3): the product's bf16 error must stay within BF16_VS_REFERENCE_AUTOCAST x that (measured 0.5-1.4x);
# (b) fp8w -- 16 x the bf16 bound would be four lost significand bits; the bound below is tighter than that.
AR_BOUNDS = {
	"f32": dict(kind="abs", logits=5e-4, latents=5e-5),
	"f16": dict(kind="rel", logits=8e-3, latents=5e-3),
	"bf16": dict(kind="rel", logits=6e-2, latents=6e-2),
	"fp8w": dict(kind="rel", logits=5e-1, latents=3e-1),
}
BF16_VS_REFERENCE_AUTOCAST = 2.5


@pytest.mark.parametrize("variant", ["peaked", "outlier"])
@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "fp8w"])
def test_ar_logits_and_latents_against_the_reference(golden, variant, dtype):
	g = golden("stress_ar")
	p = variant + "::"
	cfg = W.AR_SMALL
	model = build_ar(ar_sd(variant), cfg, dtype, max_batch=4, max_ctx=64)
	text, cond, toks, codes = t(g[p + "text"]).to(DEV), t(g[p + "cond"]).to(DEV), t(g[p + "dec_tokens"]).to(DEV), t(g[p + "codes"]).to(DEV)
	B = toks.shape[0]
	b = AR_BOUNDS[dtype]
	fn = maxerr if b["kind"] == "abs" else relerr
	logits = model._prefill(cond, text, B)
	e_pre = fn(logits, g[p + "prefill_logits"])
	dec = []
	for k in range(toks.shape[1]):
		model._decode(toks[:, k].contiguous(), logits)
		dec.append(logits.clone())
	e_dec = fn(torch.stack(dec, 1), g[p + "decode_logits"])
	lat = model.forward(cond.repeat(B, 1), text.repeat(B, 1), torch.tensor([text.shape[1]] * B, dtype=torch.int32), codes,
						torch.tensor([codes.shape[1] * 1024] * B), return_latent=True, clip_inputs=False)
	e_lat = fn(lat, g[p + "latents"])
	# ... and along the sequence the reference itself sampled under top-k 16 / top-p 0.8 / penalty 2 (the logits every token was drawn from)
	q = p + "topk16_topp_pen::"
	meta = json.loads(str(g[q + "meta"]))
	ids = t(g[q + "ids"]).to(DEV)
	lg = model._prefill(t(g[q + "cond"]).to(DEV), t(g[q + "text"]).to(DEV), meta["B"])
	rows = [lg[:, COLS.to(DEV)].clone()]
	for k in range(ids.shape[1] - 1):
		model._decode(ids[:, k].contiguous(), lg)
		rows.append(lg[:, COLS.to(DEV)].clone())
	e_seq = fn(torch.stack(rows, 1), g[q + "logits_sub"])
	record(f"ar_{variant}_{dtype}", dict(kind=b["kind"], prefill=e_pre, decode=e_dec, latents=e_lat, sampled_sequence=e_seq, health=model._check_health()))
	assert e_pre < b["logits"] and e_dec < b["logits"] and e_seq < b["logits"] and e_lat < b["latents"], (variant, dtype, e_pre, e_dec, e_seq, e_lat)
	if dtype == "bf16":      # no worse than the reference's own 16-bit mode in this regime
		r_pre, r_dec = relerr(g[p + "prefill_logits_autocast_bf16"], g[p + "prefill_logits"]), relerr(g[p + "decode_logits_autocast_bf16"], g[p + "decode_logits"])
		record(f"ar_{variant}_reference_autocast_bf16", dict(kind="rel", prefill=r_pre, decode=r_dec))
		assert e_pre < BF16_VS_REFERENCE_AUTOCAST * r_pre and e_dec < BF16_VS_REFERENCE_AUTOCAST * r_dec, (variant, e_pre, r_pre, e_dec, r_dec)


# ------------------------------------------------------------------------------------------------ AR: ids
@pytest.mark.parametrize("variant", ["peaked", "outlier"])
@pytest.mark.parametrize("hf_exact", [False, True])
def test_free_running_ids_f32_under_the_cli_warpers(golden, variant, hf_exact):
	"""every warper combination of the fixture (top-k 16 alone, + top-p 0.8 + repetition penalty 2, + typical mass 0.9, top-p alone, typical alone): the product's
	free-running f32 ids equal the oracle's own loop bit for bit -- the oracle's CPU-generator ids are the reference's (tests/test_oracle_stress.py), here it
	samples on the device so both sides consume the same Philox stream.  With `hf_exact_top_p` the cumulative-mass warpers run as HF's torch ops in front of the
	kernel, without it inside the kernel (exact fixed-point masses): both must give the oracle's ids.  The folded LayerNorm's health word stays silent: outlier
	channels inflate the row's std with them (|mean| / std <= sqrt(2 / d)), which the fold carries (docs/history/r03.md)."""
	g = golden("stress_ar")
	cfg = W.AR_SMALL
	sd = ar_sd(variant)
	model = build_ar(sd, cfg, "f32", max_batch=4, max_ctx=96, hf_exact_top_p=hf_exact)
	oracle = O.AROracle(sd, cfg)
	names = sorted({k.split("::")[1] for k in g if k.startswith(variant + "::") and k.count("::") == 2})
	assert len(names) == 5
	for name in names:
		q = f"{variant}::{name}::"
		meta = json.loads(str(g[q + "meta"]))
		kw = dict(meta["kw"])
		kw.setdefault("top_k", 0)
		tm = kw.pop("typical_mass", None)
		typ = dict(typical_sampling=tm is not None, typical_mass=tm if tm is not None else 0.9)
		text, cond = t(g[q + "text"]), t(g[q + "cond"])
		with torch.inference_mode():
			want = O.inference_speech(oracle, cond, text, num_return_sequences=meta["B"], max_generate_length=meta["max_new"], sample_device="cuda", **typ, **kw)
		with warnings.catch_warnings():
			warnings.simplefilter("error")                                   # a health warning would be an error here
			got = model.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, num_return_sequences=meta["B"], max_generate_length=meta["max_new"], **typ, **kw)
		assert got.shape == want.shape and torch.equal(got.cpu(), want), (variant, name, hf_exact, (got.cpu() != want).nonzero()[:4].tolist())
		assert model.last_health == 0


# measured (profiles/r05_stress_errors.json): bf16 0.969 / 0.979 (peaked: top-k 16 / top-p), 0.991 / 0.986 (outlier); f16 0.995-0.999.  On random weights at full size bf16 agrees on
# 0.9978 of the draws (tests/test_gpu_ids.py): a logit error of 1.3-1.7 % of a std-8 row is ~0.1 in the exponent, which moves a peaked row's probabilities by ~10 %
SIXTEEN_BIT_DRAW_AGREEMENT_FLOOR = {"bf16": 0.94, "f16": 0.985}


@pytest.mark.parametrize("variant", ["peaked", "outlier"])
@pytest.mark.parametrize("dtype", ["bf16", "f16"])
def test_sixteen_bit_draws_agree_with_the_oracle_in_the_peaked_regime(variant, dtype):
	"""north_star's "bit-exact ids" cannot hold in a 16-bit mode against an f32 reference (SURVEY section 7 asks for the agreement instead).  In the regime a trained
	checkpoint produces the question is sharper than on random weights -- a flipped draw now needs the top of a PEAKED distribution or a top-k / top-p cut to move:
	the product is teacher-forced on the oracle's ids (16 candidates x 48 tokens, top-k 16 at T = 0.8, and top-p 0.8 alone) and every draw is repeated on ITS logits with
	the oracle's noise through the oracle's warper chain; the fraction of draws that pick the oracle's token has an asserted floor."""
	cfg = W.AR_SMALL
	sd = ar_sd(variant)
	oracle = O.AROracle(sd, cfg)
	model = build_ar(sd, cfg, dtype, max_batch=16, max_ctx=96)
	B, N = 16, 48
	text = torch.randint(1, 255, (1, 9), generator=gen(101))
	cond = torch.randn(1, cfg.model_dim, generator=gen(102))
	out = {}
	for name, kw in (("topk16", dict(temperature=0.8, top_k=16)), ("topp", dict(temperature=1.0, top_k=0, top_p=0.8))):
		with torch.inference_mode():
			ids, ref_logits = O.inference_speech(oracle, cond, text, num_return_sequences=B, max_generate_length=N, sample_device="cuda", suppress_tokens=[cfg.stop_mel_token],
												 return_logits=True, **kw)
		assert ids.shape == (B, N)
		lg = model._prefill(cond.to(DEV), text.to(DEV), B)
		rows = [lg.clone()]
		idd = ids.to(DEV)
		for k in range(N - 1):
			model._decode(idd[:, k].contiguous(), lg)
			rows.append(lg.clone())
		got_logits = torch.stack(rows, 1)
		torch.manual_seed(0); torch.cuda.manual_seed_all(0)
		q = torch.empty((B, cfg.number_mel_codes), device=DEV)
		agree = 0
		for k in range(N):
			q.exponential_(1)
			po = torch.softmax(O.process_logits(None, ref_logits[:, k].to(DEV), suppress_tokens=[cfg.stop_mel_token], **kw), dim=-1)
			assert torch.equal(torch.argmax(po / q, dim=-1), idd[:, k]), k      # the replayed noise IS the oracle's
			pp = torch.softmax(O.process_logits(None, got_logits[:, k], suppress_tokens=[cfg.stop_mel_token], **kw), dim=-1)
			agree += int((torch.argmax(pp / q, dim=-1) == idd[:, k]).sum())
		out[name] = agree / (B * N)
	record(f"ar_{variant}_{dtype}_draw_agreement", out)
	assert min(out.values()) >= SIXTEEN_BIT_DRAW_AGREEMENT_FLOOR[dtype], out


def _probe(lib, lg, probe, *, temp, top_k, top_p, typical=None):
	"""is token probe[b] of row b still there after the kernel's warpers?  Noise 1 everywhere except a vanishing value at the probe: a kept probe wins
	argmax(p / q) whatever its probability, a removed one (p = 0) cannot (tests/test_gpu_parity.py::_probe_kept)."""
	from tortoise_tts_amd import _lib
	B, V = lg.shape
	q = torch.ones((B, V), device=DEV)
	q[torch.arange(B, device=DEV), probe] = 1e-30
	unf, tok = torch.ones(B, dtype=torch.long, device=DEV), torch.empty(B, dtype=torch.long, device=DEV)
	ids, col = torch.full((B, 1), -1, dtype=torch.long, device=DEV), torch.zeros(B, dtype=torch.long, device=DEV)
	a = _lib.SampleArgs()
	a.scores, a.ld, a.B, a.V, a.q, a.ldq = lg.data_ptr(), V, B, V, q.data_ptr(), V
	a.temperature, a.top_k, a.top_p, a.repetition_penalty = temp, top_k, top_p, 1.0
	if typical is not None:
		a.typical_mass = typical
	a.stop_token, a.unfinished, a.tok, a.ids, a.ids_ld, a.ids_cols, a.col = V + 5, unf.data_ptr(), tok.data_ptr(), ids.data_ptr(), 1, 1, col.data_ptr()
	_lib.check(lib.ttk_sample_step_warped(_lib.C.byref(a), _lib.stream_ptr()), "ttk_sample_step_warped")
	torch.cuda.synchronize()
	return tok == probe


@pytest.mark.parametrize("variant", ["peaked", "outlier"])
def test_kept_sets_of_the_sampling_kernel_on_the_references_own_rows(golden, variant):
	"""the logit rows the REFERENCE drew from (whole rows stored at two steps of two cases), through the kernel's top-k 16 / top-p 0.8 and typical 0.9 cuts:
	every token HF's chain keeps is kept, the best token it removes is removed.  Rows where the cut sits on an f32 tie of the cumulative mass are excused only
	if the f64 mass is within 2e-6 of the threshold (none expected on rows this peaked)."""
	from tortoise_tts_amd import _lib
	lib = _lib.load()
	g = golden("stress_ar")
	rows = torch.cat([t(g[f"{variant}::{c}::logit_rows"]).reshape(-1, 8194) for c in ("topk16_topp_pen", "typical_only")]).to(DEV)
	B, V = rows.shape
	assert float(torch.softmax(rows / 0.8, -1).max(-1)[0].median()) > 0.4
	checked = 0
	for kw in (dict(temperature=0.8, top_k=16, top_p=0.8), dict(temperature=0.8, top_k=16, top_p=1.0), dict(temperature=1.0, top_k=0, top_p=0.8),
			   dict(temperature=1.0, top_k=0, top_p=1.0, typical_mass=0.9), dict(temperature=0.8, top_k=16, top_p=1.0, typical_mass=0.9)):
		sc = O.process_logits(None, rows, **kw)
		kept = torch.isfinite(sc)
		n_kept = kept.sum(-1)
		assert int(n_kept.min()) >= 1 and int(n_kept.max()) < 200
		pk = dict(temp=kw["temperature"], top_k=kw["top_k"], top_p=kw["top_p"], typical=kw.get("typical_mass"))
		order = torch.argsort(torch.where(kept, rows, torch.full_like(rows, float("-inf"))), dim=-1, descending=True)
		for j in range(int(n_kept.max())):                              # the j-th best kept token of every row that has one
			has = n_kept > j
			ok = _probe(lib, rows, order[:, j], **pk)
			assert bool((ok | ~has).all()), (variant, kw, j, (~ok & has).nonzero().flatten().tolist())
			checked += int(has.sum())
		removed_best = torch.where(~kept, rows, torch.full_like(rows, float("-inf"))).argmax(-1)
		assert not bool(_probe(lib, rows, removed_best, **pk).any()), (variant, kw)
		if kw.get("typical_mass") is not None and kw["top_k"] == 0:      # typical sampling drops the MOST likely token when it is far from the entropy: probe the removed token closest to it as well
			lp = torch.log_softmax(rows, -1)
			H = -(lp * lp.exp()).nansum(-1, keepdim=True)
			near = torch.where(~kept, (-lp - H).abs(), torch.full_like(lp, float("inf"))).argmin(-1)
			assert not bool(_probe(lib, rows, near, **pk).any()), (variant, kw)
	assert checked > 5 * B


def test_ar_full_size_outlier_weights_logits_and_ids(golden):
	"""full-size UnifiedVoice (the benchmarked geometry: k_gemv with the folded LayerNorm) on the outlier stress weights: f32 logits / latents against the
	reference, the bf16 handle's within its bound, free-running f32 ids equal to the oracle's loop under top-k 16 + repetition penalty 2"""
	g = golden("stress_ar_full")
	cfg = W.AR_FULL
	sd = ar_sd("outlier", cfg, int(g["seed"]))
	text, cond, toks, codes = t(g["text"]), t(g["cond"]), t(g["dec_tokens"]), t(g["codes"])
	B, cols = int(g["B"]), t(g["logit_cols"]).to(DEV)
	meta = json.loads(str(g["stream_meta"]))
	with torch.inference_mode():
		want = O.inference_speech(O.AROracle(sd, cfg), cond, text, num_return_sequences=meta["B"], max_generate_length=meta["max_new"], sample_device="cuda", **meta["kw"])
		# f32 noise is what this regime amplifies (300x channels through 30 layers): the yardstick for the f32 handle is the reference's OWN distance from the f64 value of the
		# same function (the oracle on f64 weights): 3.6e-4 on the prefill logits, 3.1e-4 on the latents -- the product must stay within F32_VS_TRUTH x that
		o64 = O.AROracle({k: v.double() for k, v in sd.items()}, cfg)
		lg64, _, _ = o64.prefill(o64.prefix_embeddings(cond.double(), text), B)
		lg64 = lg64[:, -1][:, cols.cpu()]
		lat64 = o64.forward_latents(cond.double().repeat(B, 1), text.repeat(B, 1), codes)[:, :, :128]
		ref_pre64, ref_lat64 = maxerr(g["prefill_logits"], lg64), maxerr(g["latents"], lat64)
		del o64
	F32_VS_TRUTH = 4.0
	for dtype in ("f32", "bf16"):
		model = build_ar(sd, cfg, dtype, max_batch=2, max_ctx=64)
		b = AR_BOUNDS[dtype]
		fn = maxerr if b["kind"] == "abs" else relerr
		logits = model._prefill(cond.to(DEV), text.to(DEV), B)
		logits_pre = logits.clone()
		e_pre = fn(logits[:, cols], g["prefill_logits"])
		dec = []
		for k in range(toks.shape[1]):
			model._decode(toks[:, k].contiguous().to(DEV), logits)
			dec.append(logits[:, cols].clone())
		e_dec = fn(torch.stack(dec, 1), g["decode_logits"])
		lat = model.forward(cond.repeat(B, 1).to(DEV), text.repeat(B, 1).to(DEV), torch.tensor([text.shape[1]] * B, dtype=torch.int32), codes.to(DEV),
							torch.tensor([codes.shape[1] * 1024] * B), return_latent=True, clip_inputs=False)
		e_lat = fn(lat[:, :, :128], g["latents"])
		with warnings.catch_warnings():
			warnings.simplefilter("error")
			got = model.inference_speech(cond.to(DEV), text.to(DEV), do_sample=True, num_return_sequences=meta["B"], max_generate_length=meta["max_new"], **meta["kw"]).cpu()
		agree = float((got == want).float().mean())
		rec = dict(kind=b["kind"], prefill=e_pre, decode=e_dec, latents=e_lat, free_running_id_agreement=agree, health=model.last_health)
		if dtype == "f32":               # measured: logits 8.9e-4 / 8.7e-4 abs, latents 7.0e-4 abs (|latent| to 33); vs the f64 value 9.6e-4 / 7.3e-4 against the reference's own 3.6e-4 / 3.1e-4
			t_pre, t_lat = maxerr(logits_pre[:, cols], lg64), maxerr(lat[:, :, :128], lat64)
			rec.update(prefill_vs_f64=t_pre, latents_vs_f64=t_lat, reference_prefill_vs_f64=ref_pre64, reference_latents_vs_f64=ref_lat64)
			assert e_pre < 3e-3 and e_dec < 3e-3 and e_lat < 2.5e-3, (dtype, e_pre, e_dec, e_lat)
			assert t_pre < F32_VS_TRUTH * ref_pre64 and t_lat < F32_VS_TRUTH * ref_lat64, (t_pre, ref_pre64, t_lat, ref_lat64)
		else:                            # measured 6.2e-2 / 4.0e-2 on the logits, 2.2e-2 on the latents; the reference's own autocast(bf16) deviation here: 7.4e-2 / 3.9e-2
			r_pre, r_dec = relerr(g["prefill_logits_autocast_bf16"], g["prefill_logits"]), relerr(g["decode_logits_autocast_bf16"], g["decode_logits"])
			rec.update(reference_autocast_prefill=r_pre, reference_autocast_decode=r_dec)
			assert e_pre < 1.5 * r_pre and e_dec < 1.5 * r_dec and e_lat < b["latents"], (dtype, e_pre, r_pre, e_dec, r_dec, e_lat)
		record(f"ar_full_outlier_{dtype}", rec)
		assert model.last_health == 0
		if dtype == "f32":
			assert torch.equal(got, want), (got != want).nonzero()[:4].tolist()
		del model


# ------------------------------------------------------------------------------------------------ diffusion
# f32: absolute (|E| to 11 / 18, |y| to 2.5 / 3, x in [-1, 1] at the end); 16-bit / fp8: relative L2 against the reference's f32 values.
# Measured on MI355X (round 5, profiles/r05_stress_errors.json -> DESIGN.md section 2), small model / full size at T = 1088:
#   f32  E 1.3e-4 / 6.0e-4, evaluation 3.4e-4 / 1.2e-3, x 3.1e-4 / 2.7e-4 abs          f16  E 7.9e-3 / 7.5e-3, evaluation 2.0e-3 / 1.8e-2, x 2.8e-3 / 3.0e-3
#   bf16 E 2.8e-2 / 5.9e-2, evaluation 1.6e-2 / 1.2e-1, x 1.9e-2 / 2.0e-2              fp8w E 0.17 / 0.26, evaluation 0.12 / 0.37, x 0.13 / 0.06      fp8 E 0.20 / 0.32, evaluation 0.17 / 0.42, x 0.17 / 0.07
# Bounds = 2.5-4x the measurement (full size: FULL_K x the small model's bound for f32).  Independent criterion for the 16-bit modes: the REFERENCE'S OWN 16-bit mode
# (DiffusionTTS(use_fp16=True): layers >= 1 under autocast, diffusion.py:1559-1561) deviates from its f32 evaluation by 5.5e-2 (small) / 2.2e-1 (full size) here -- scores of
# +-100 through 8-bit significands move softmax weights by tens of percent whoever computes them -- and the product's bf16 evaluation must not be further away than that
# (measured 0.29x / 0.55x).
# fp8 / fp8w, round 6: the q / k / v projection keeps 16-bit operands in both modes (csrc/diff.hip load_attn) -- round 5's evaluations lost 37-42 % (full
# size) / 12-17 % (small) to e4m3 on q and k; now measured (profiles/r06_stress_errors.json), small model / full size:
#   fp8w E 5.6e-2 / 9.6e-2, evaluation 4.9e-2 .. 8.2e-2 / 1.2e-1 .. 2.7e-1, x 1.1e-1 / 4.4e-2        fp8 E 9.4e-2 / 1.2e-1, evaluation 7.2e-2 .. 1.3e-1 / 1.7e-1 .. 3.2e-1, x 1.2e-1 / 5.3e-2
# Bounds = 1.5 x the larger of the two sizes' measurements (a 2 x regression fails), AND the independent criterion: the conditioned evaluation no further from the reference's
# f32 one than 2 x the reference's own 16-bit mode is (1.1e-1 small / 4.4e-1 full size; measured 1.3 x / 1.45 x for fp8).
DIFF_BOUNDS = {
	"f32": dict(kind="abs", E=5e-4, y=1e-3, x=1e-3),
	"f16": dict(kind="rel", E=2.5e-2, y=5e-2, x=1e-2),
	"bf16": dict(kind="rel", E=1.5e-1, y=3e-1, x=6e-2),
	"fp8w": dict(kind="rel", E=1.45e-1, y=4.0e-1, x=1.6e-1),
	"fp8": dict(kind="rel", E=1.8e-1, y=4.8e-1, x=1.75e-1),
}
FULL_K = 4.0


def build_diff(sd, cfg, dtype):
	from tortoise_tts_amd.diffusion import DiffusionTTS
	return DiffusionTTS(sd, cfg, dtype=dtype, device=DEV)


def ddim_chunks(model, noise, E, T, n_steps, lo_hi):
	"""x after each (lo, hi) slice of the n_steps schedule, run from the highest index down through the whole-loop entry"""
	from tortoise_tts_amd import _lib
	from tortoise_tts_amd.diffusion import get_diffuser
	d = get_diffuser(steps=n_steps, cond_free=True)
	x = noise.to(DEV).clone()
	Ed = E.to(DEV, torch.float32).contiguous()
	out = []
	for lo, hi in lo_hi:
		k = hi - lo
		steps = (_lib.StepC * k)(*[d.step_coefs(i, "ddim") for i in range(lo, hi)])
		_lib.check(model.lib.ttk_diff_sample_ddim(model._h, x.data_ptr(), Ed.data_ptr(), 1, T, steps, k, _lib.stream_ptr()), "ttk_diff_sample_ddim")
		torch.cuda.synchronize()
		out.append(x.clone())
	return out


@pytest.mark.parametrize("dtype", ["f32", "bf16", "f16", "fp8w", "fp8"])
def test_diffusion_small_against_the_reference(golden, dtype):
	g = golden("stress_diff")
	cfg = W.DIFF_SMALL
	sd = W.stress_diffusion(W.synth_state_dict(W.diffusion_shapes(cfg), int(g["seed"])), cfg)
	model = build_diff(sd, cfg, dtype)
	T = int(g["T"])
	b = DIFF_BOUNDS[dtype]
	fn = maxerr if b["kind"] == "abs" else relerr
	e_E = fn(model.timestep_independent(t(g["latents"]).to(DEV), t(g["cond"]).to(DEV), T, False), g["E"])
	x, ts, Eg = t(g["x"]).to(DEV), t(g["t"]).to(DEV), t(g["E"]).to(DEV)
	e_yc = fn(model(x, ts, precomputed_aligned_embeddings=Eg), g["y_cond"])
	e_yu = fn(model(x, ts, precomputed_aligned_embeddings=Eg, conditioning_free=True), g["y_uncond"])
	xs = ddim_chunks(model, t(g["noise"]), Eg[:1], T, 8, [(6, 8), (4, 6), (0, 4)])
	e_x = {n: fn(xs[i], g[f"x_after_{n}"]) for i, n in enumerate((2, 4, 8))}
	record(f"diff_small_{dtype}", dict(kind=b["kind"], E=e_E, y_cond=e_yc, y_uncond=e_yu, **{f"x_after_{n}": v for n, v in e_x.items()}))
	assert e_E < b["E"] and e_yc < b["y"] and e_yu < b["y"], (dtype, e_E, e_yc, e_yu)
	assert all(v < b["x"] for v in e_x.values()), (dtype, e_x)
	if dtype != "f32":      # 16-bit modes: no further from the reference's f32 evaluation than the reference's own 16-bit mode is; the fp8 modes: within twice that
		r = relerr(g["y_cond_ref_fp16mode"], g["y_cond"])
		record("diff_small_reference_fp16mode", dict(kind="rel", y_cond=r))
		assert e_yc < (r if dtype in ("bf16", "f16") else 2 * r), (dtype, e_yc, r)
	assert torch.isfinite(xs[-1]).all() and xs[-1].abs().max() <= 1.0 + 1e-5


@pytest.mark.parametrize("dtype", ["f32", "bf16", "fp8"] + (["f16", "fp8w"] if ALL_MODES else []))
def test_diffusion_full_size_at_T1088_against_the_reference(golden, dtype):
	"""configs[1]'s shape on the stress weights: E, one evaluation pair and the last 4 of the 80 DDIM steps, every 8th frame / the final mel whole"""
	g = golden("stress_diff_cfg1")
	cfg = W.DIFF_FULL
	sd = W.stress_diffusion(W.synth_state_dict(W.diffusion_shapes(cfg), 2), cfg)
	model = build_diff(sd, cfg, dtype)
	M, T, st = int(g["M"]), int(g["T"]), int(g["stride"])
	assert T == 1088
	lat = torch.randn(1, M, 1024, generator=gen(31))
	dcond = torch.randn(1, 2048, generator=gen(32))
	x = torch.randn(1, 100, T, generator=gen(33))
	ts = torch.tensor([1500])
	b = DIFF_BOUNDS[dtype]
	fn = maxerr if b["kind"] == "abs" else relerr
	E = model.timestep_independent(lat.to(DEV), dcond.to(DEV), T, False)
	e_E = fn(E[:, :, ::st], g["E_sub"])
	with torch.inference_mode():      # the evaluations take the oracle's f32 E (equal to the reference's: tests/test_oracle_stress.py) so each stage is compared on its own
		Eo = O.DiffusionOracle(sd, cfg).timestep_independent(lat, dcond, T).to(DEV)
	e_yc = fn(model(x.to(DEV), ts.to(DEV), precomputed_aligned_embeddings=Eo)[:, :, ::st], g["y_cond_sub"])
	e_yu = fn(model(x.to(DEV), ts.to(DEV), precomputed_aligned_embeddings=Eo, conditioning_free=True)[:, :, ::st], g["y_uncond_sub"])
	xm = ddim_chunks(model, x, Eo, T, 80, [(0, 4)])[0]
	e_x = fn(xm, g["mel"])
	record(f"diff_full_{dtype}", dict(kind=b["kind"], E=e_E, y_cond=e_yc, y_uncond=e_yu, mel_last4=e_x))
	k = FULL_K if dtype == "f32" else 1.0
	assert e_E < k * b["E"] and e_yc < k * b["y"] and e_yu < k * b["y"] and e_x < k * b["x"], (dtype, e_E, e_yc, e_yu, e_x)
	if dtype != "f32":
		r = relerr(g["y_cond_ref_fp16mode_sub"], g["y_cond_sub"])
		record("diff_full_reference_fp16mode", dict(kind="rel", y_cond=r))
		assert e_yc < (r if dtype in ("bf16", "f16") else 2 * r), (dtype, e_yc, r)


# ------------------------------------------------------------------------------------------------ the WHOLE loop in the trained-checkpoint regime
# stress_diff_cfg1_loop.npz: the reference's `ddim_sample_loop_progressive` (diffusion.py:765-810) on the full-size stress weights at T = 1088, all 80 steps from seeded noise,
# in f32 AND in the reference's own 16-bit mode (`enable_fp16`, :1559-1561); x after 8 / 16 / 40 / 72 steps on every 8th frame, the final mel whole.
# Criterion for the 16-bit product modes (independent of this build's own measurements): at every checkpoint, no further from the reference's f32 x than the reference's own
# 16-bit loop is.  f32: absolute.  fp8 modes: STRESS_LOOP_BOUNDS below (1.5 x the measurement, stated next to it).
STRESS_LOOP_CHECKPOINTS = (8, 16, 40, 72, 80)
# Measured on MI355X (round 6, profiles/r06_stress_errors.json) after 8 / 16 / 40 / 72 / 80 steps:
#   f32  abs 1.3e-4 / 2.0e-4 / 5.2e-4 / 2.6e-3 / 3.9e-3 (x in [-1, 1] at the end; the peaked regime amplifies rounding-level differences between two f32 implementations step by step)
#   f16  rel 3.2e-4 / 6.0e-4 / 2.5e-3 / 1.3e-2 / 1.9e-2        bf16 rel 1.1e-3 / 2.5e-3 / 1.2e-2 / 4.2e-2 / 5.6e-2
#   fp8w rel 1.7e-3 / 4.0e-3 / 2.3e-2 / 7.6e-2 / 1.0e-1        fp8  rel 1.8e-3 / 4.4e-3 / 2.5e-2 / 8.4e-2 / 1.1e-1
#   the reference's OWN 16-bit loop vs its f32 loop: 1.5e-3 / 3.7e-3 / 2.1e-2 / 7.1e-2 / 9.6e-2 -- bf16 ends at 0.58 x of it, f16 at 0.20 x, fp8 at 1.15 x
# Bounds: f32 / f16 2.5 x the measurement; bf16 1.5 x (and the criterion above, which is the tighter one); fp8 modes 1.5 x, and the final mel within 2 x the reference's own 16-bit loop.
STRESS_LOOP_BOUNDS = {
	"f32": dict(kind="abs", at={8: 4e-4, 16: 5e-4, 40: 1.5e-3, 72: 7e-3, 80: 1e-2}),
	"f16": dict(kind="rel", at={8: 8e-4, 16: 1.5e-3, 40: 6.5e-3, 72: 3.2e-2, 80: 4.8e-2}),
	"bf16": dict(kind="rel", at={8: 1.7e-3, 16: 3.7e-3, 40: 1.9e-2, 72: 6.3e-2, 80: 8.4e-2}),
	"fp8w": dict(kind="rel", at={8: 2.5e-3, 16: 6e-3, 40: 3.4e-2, 72: 1.15e-1, 80: 1.5e-1}),
	"fp8": dict(kind="rel", at={8: 2.7e-3, 16: 6.7e-3, 40: 3.7e-2, 72: 1.26e-1, 80: 1.67e-1}),
}


@pytest.mark.parametrize("dtype", ["f32", "bf16", "fp8"] + (["f16", "fp8w"] if ALL_MODES else []))
def test_whole_80_step_loop_full_size_in_the_peaked_regime_against_the_reference(golden, dtype):
	g = golden("stress_diff_cfg1_loop")
	cfg = W.DIFF_FULL
	sd = W.stress_diffusion(W.synth_state_dict(W.diffusion_shapes(cfg), 2), cfg)
	model = build_diff(sd, cfg, dtype)
	M, T, st = int(g["M"]), int(g["T"]), int(g["stride"])
	assert T == 1088 and int(g["steps"]) == 80 and tuple(g["checkpoints"]) == STRESS_LOOP_CHECKPOINTS
	lat = torch.randn(1, M, 1024, generator=gen(31))
	dcond = torch.randn(1, 2048, generator=gen(32))
	noise = torch.randn(1, 100, T, generator=gen(34))
	with torch.inference_mode():      # every mode starts from the oracle's f32 E (equal to the reference's: tests/test_oracle_stress.py), so the LOOP is what is compared
		Eo = O.DiffusionOracle(sd, cfg).timestep_independent(lat, dcond, T)
	lo_hi, done = [], 0
	for n in STRESS_LOOP_CHECKPOINTS:
		lo_hi.append((80 - n, 80 - done))
		done = n
	xs = ddim_chunks(model, noise, Eo, T, 80, lo_hi)
	b = STRESS_LOOP_BOUNDS[dtype]
	fn = maxerr if b["kind"] == "abs" else relerr
	errs, ref16 = {}, {}
	for n, x in zip(STRESS_LOOP_CHECKPOINTS, xs):
		want = g["mel"] if n == 80 else g[f"x_after_{n}_sub"]
		amp = g["mel_ref_fp16mode"] if n == 80 else g[f"x_after_{n}_ref_fp16mode_sub"]
		errs[n] = fn(x if n == 80 else x[:, :, ::st], want)
		ref16[n] = relerr(amp, want)
	record(f"stress_loop_{dtype}", dict(kind=b["kind"], **{str(n): errs[n] for n in STRESS_LOOP_CHECKPOINTS}))
	record("stress_loop_reference_fp16mode", dict(kind="rel", **{str(n): ref16[n] for n in STRESS_LOOP_CHECKPOINTS}))
	for n in STRESS_LOOP_CHECKPOINTS:
		assert errs[n] < b["at"][n], (dtype, n, errs)
		if dtype in ("bf16", "f16"):
			assert errs[n] < ref16[n], (dtype, n, errs, ref16)
	if dtype in ("fp8", "fp8w"):
		assert errs[80] < 2 * ref16[80], (dtype, errs, ref16)
	assert torch.isfinite(xs[-1]).all() and xs[-1].abs().max() <= 1.0 + 1e-5

