"""Generate tests/golden/*.npz from the REFERENCE itself, run in the build container.

    python oracle/make_golden.py            # needs /root/reference (never present on the GPU box)

The reference modules are imported by path (oracle/ref_shim.py), loaded with the repo's seeded
synthetic weights (tortoise_tts_amd/weights.py: the same numbers every test regenerates), and run on
seeded inputs; only inputs and outputs are stored.  A fixture is data: no reference source text,
bytecode or pickled module is written anywhere.

Fixtures
  ar_small.npz    UnifiedVoice(layers=2, dim=128, heads=2): prefill + 6 KV-cached decode steps (logits),
                  forward(return_latent=True) latents                       (unified_voice.py:178-254, :544-599)
  ar_full.npz     full-size UnifiedVoice(): prefill + 2 decode steps (logit slices), latents slice
  diff_small.npz  DiffusionTTS(128 ch, 2 layers, 2 heads): timestep_independent, forward cond/uncond,
                  DDIM 4 steps, p-sampler 4 steps                            (diffusion.py:1487-1574, :500-810)
  diff_full.npz   full-size DiffusionTTS(): one forward (cond + uncond) at T=26
  diff_cfg1.npz   full-size DiffusionTTS() at configs[1]'s T = 1088: E, one evaluation pair, the last 8 of the 80 DDIM steps
  diff_cfg1_loop.npz  the same model's WHOLE 80-step DDIM loop from seeded noise, x at 8 / 16 / 40 / 72 steps + final mel
  e2e_cfg1.npz    one configs[1] utterance end to end in f32: the reference's sample_stream codes -> latents -> E -> 80 DDIM steps -> mel
  cond_small.npz  UnifiedVoice.get_conditioning / DiffusionTTS.get_conditioning, small widths, 2 clips   (unified_voice.py:535-542,
  cond_full.npz   diffusion.py:1477-1485); full-size encoders at short clips
  stft_ref.npz    STFT(1024, 256, 1024).transform magnitudes on seeded audio (arch_utils.py:560-623)
  tokenizer.npz   VoiceBpeTokenizer.encode / decode / preprocess_text on the reference vocabulary, digit-free ASCII texts (tokenizer.py:154-177)
  schedule.npz    get_diffuser(steps) tables for steps in 4, 30, 80, 200    (diffusion.py:1576-1590)
  stress_ar.npz / stress_ar_full.npz / stress_diff.npz / stress_diff_cfg1.npz
                  the same reference classes on weights moved to a trained checkpoint's numerical regime (tortoise_tts_amd/weights.py stress_*):
                  peaked logits and attention rows, outlier residual channels, a near-constant GroupNorm group, large scale / shift; logits,
                  latents, sample_stream ids under the CLI's warpers (top-k 16, top-p, repetition penalty, the reference's TypicalLogitsWarper),
                  E / evaluations / DDIM steps, small models in full and full-size slices
"""
import math
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from tortoise_tts_amd import weights as W  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def gen(seed):
	g = torch.Generator(device="cpu")
	g.manual_seed(seed)
	return g


def load_into(module, sd):
	missing, unexpected = module.load_state_dict(sd, strict=False)
	assert not unexpected, unexpected
	return module.eval()


def ar_case(uv_mod, cfg, seed, B, Tt, n_dec, M, full):
	sd = W.synth_state_dict(W.ar_shapes(cfg), seed)
	m = uv_mod.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, checkpointing=False)
	load_into(m, sd)
	text = torch.randint(1, 255, (1, Tt), generator=gen(seed + 1))
	cond = torch.randn(1, cfg.model_dim, generator=gen(seed + 2))
	dec_tokens = torch.randint(0, 8192, (B, n_dec), generator=gen(seed + 3))
	codes = torch.randint(0, 8192, (B, M), generator=gen(seed + 4))
	out = dict(text=text.numpy(), cond=cond.numpy(), dec_tokens=dec_tokens.numpy(), codes=codes.numpy(),
				seed=np.int64(seed), B=np.int64(B))
	with torch.inference_mode():
		ids = m.compute_embeddings(cond, text)             # builds inference_model, stores prefix emb
		ids = ids.repeat(B, 1)
		im = m.inference_model
		P1 = ids.shape[1]
		r = im.forward(input_ids=ids, attention_mask=torch.ones(B, P1, dtype=torch.long), use_cache=True, return_dict=True)
		pre = r.logits[:, -1].float()
		past = r.past_key_values
		dec = []
		for k in range(1, n_dec + 1):
			r = im.forward(input_ids=dec_tokens[:, k - 1:k], past_key_values=past,
							attention_mask=torch.ones(B, P1 + k, dtype=torch.long), use_cache=True, return_dict=True)
			past = r.past_key_values
			dec.append(r.logits[:, -1].float())
		dec = torch.stack(dec, 1)
		lat = m.forward(cond.repeat(B, 1), text.repeat(B, 1), torch.tensor([Tt] * B, dtype=torch.int32), codes,
						torch.tensor([M * cfg.mel_length_compression] * B), return_latent=True, clip_inputs=False)
	if full:   # keep the fixture small: vocabulary slices + a latent slice
		sel = torch.cat([torch.arange(0, 96), torch.arange(8100, 8194)])
		out.update(logit_cols=sel.numpy(), prefill_logits=pre[:, sel].numpy(), decode_logits=dec[:, :, sel].numpy(),
					latents=lat[:, :, :128].numpy())
	else:
		out.update(prefill_logits=pre.numpy(), decode_logits=dec.numpy(), latents=lat.numpy())
	return out


def diff_case(d_mod, cfg, seed, b, M, full):
	sd = W.synth_state_dict(W.diffusion_shapes(cfg), seed)
	m = d_mod.DiffusionTTS(model_channels=cfg.model_channels, num_layers=cfg.num_layers,
							in_latent_channels=cfg.in_latent_channels, num_heads=cfg.num_heads)
	load_into(m, sd)
	T = M * 4 * 24000 // 22050
	lat = torch.randn(b, M, cfg.in_latent_channels, generator=gen(seed + 1))
	cond = torch.randn(b, 2 * cfg.model_channels, generator=gen(seed + 2))
	x = torch.randn(b, 100, T, generator=gen(seed + 3))
	t = torch.tensor([1333, 2666][:b])
	out = dict(latents=lat.numpy(), cond=cond.numpy(), x=x.numpy(), t=t.numpy(), T=np.int64(T), seed=np.int64(seed))
	with torch.inference_mode():
		E = m.timestep_independent(lat, cond, T, False)
		yc = m(x, t, precomputed_aligned_embeddings=E)
		yu = m(x, t, precomputed_aligned_embeddings=E, conditioning_free=True)
		out.update(E=E.numpy(), y_cond=yc.numpy(), y_uncond=yu.numpy())
		if not full:
			noise = torch.randn(1, 100, T, generator=gen(seed + 5))
			for sampler in ("ddim", "p"):
				for cf in (True, False):
					diffuser = d_mod.get_diffuser(steps=4, cond_free=cf)
					torch.manual_seed(seed + 6)
					mel = diffuser.sample_loop(m, (1, 100, T), sampler=sampler, noise=noise,
												model_kwargs={"precomputed_aligned_embeddings": E[:1]}, progress=False)
					out[f"{sampler}_cf{int(cf)}"] = mel.numpy()
			out["noise"] = noise.numpy()
			out["sampler_seed"] = np.int64(seed + 6)
	return out


def diff_cfg1_case(d_mod):
	"""configs[1] at its own size through the REFERENCE: full-size DiffusionTTS, 250 mel tokens -> T = 1088 frames; timestep_independent,
	one conditioned + one conditioning-free evaluation, and the LAST 8 steps of the 80-step DDIM schedule (`ddim_sample` :646-694 with the
	ramped conditioning-free weight) started from x as x_8.  Stored subsampled (the full tensors are megabytes of noise-like floats): E and
	the evaluations on every 8th frame, the final mel whole.  tests/test_gpu_bench_shapes.py regenerates the inputs from the same seeds."""
	cfg = W.DIFF_FULL
	sd = W.synth_state_dict(W.diffusion_shapes(cfg), 1)
	m = d_mod.DiffusionTTS(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_channels=cfg.in_channels,
						   in_latent_channels=cfg.in_latent_channels, out_channels=cfg.out_channels, num_heads=cfg.num_heads)
	load_into(m, sd)
	M, T = 250, 250 * 4 * 24000 // 22050
	lat = torch.randn(1, M, 1024, generator=gen(11))
	dcond = torch.randn(1, 2048, generator=gen(12))
	x = torch.randn(1, 100, T, generator=gen(13))
	t = torch.tensor([1500])
	diffuser = d_mod.get_diffuser(steps=80, cond_free=True)
	with torch.inference_mode():
		E = m.timestep_independent(lat, dcond, T, False)
		yc = m(x, t, precomputed_aligned_embeddings=E)
		yu = m(x, t, precomputed_aligned_embeddings=E, conditioning_free=True)
		xm = x
		torch.manual_seed(0)
		for i in reversed(range(8)):
			xm = diffuser.ddim_sample(m, xm, torch.tensor([i]), clip_denoised=True, model_kwargs={"precomputed_aligned_embeddings": E}, eta=0.0)["sample"]
	return dict(T=np.int64(T), M=np.int64(M), stride=np.int64(8), E_sub=E[:, :, ::8].numpy(), y_cond_sub=yc[:, :, ::8].numpy(),
				y_uncond_sub=yu[:, :, ::8].numpy(), mel=xm.numpy())


LOOP_CHECKPOINTS = (8, 16, 40, 72, 80)    # x after this many of the 80 DDIM steps is stored


def _ddim_loop_with_checkpoints(d_mod, m, noise, E, steps=80):
	"""The reference's whole DDIM loop (`sample_loop(sampler="ddim")` -> ddim_sample_loop :734-763 -> ddim_sample_loop_progressive :765-810, the
	generator driven here so the intermediate samples can be kept), conditioning-free guidance on with its ramp as `get_diffuser` builds it."""
	diffuser = d_mod.get_diffuser(steps=steps, cond_free=True)
	keep, done = {}, 0
	torch.manual_seed(0)                       # ddim_sample draws (and, with eta = 0, ignores) one randn_like per step
	t0 = time.time()
	for out in diffuser.ddim_sample_loop_progressive(m, tuple(noise.shape), noise=noise, clip_denoised=True,
													 model_kwargs={"precomputed_aligned_embeddings": E}, device="cpu", progress=False, eta=0.0):
		done += 1
		if done in LOOP_CHECKPOINTS:
			keep[done] = out["sample"].clone()
			print(f"  step {done}/{steps}: {time.time() - t0:.0f}s, |x|max {float(keep[done].abs().max()):.3f}", flush=True)
	assert done == steps
	return keep


def diff_cfg1_loop_case(d_mod):
	"""configs[1]'s WHOLE diffusion through the REFERENCE: diff_cfg1's model, latents and conditioning at T = 1088, the full
	80-step DDIM loop from seeded start noise.  Stored: x after 8 / 16 / 40 / 72 steps on every 8th frame and the final mel whole.
	tests/test_gpu_ddim_full.py regenerates the inputs from the same seeds and asserts the per-dtype bounds of DESIGN.md section 2 at every checkpoint."""
	cfg = W.DIFF_FULL
	sd = W.synth_state_dict(W.diffusion_shapes(cfg), 1)
	m = d_mod.DiffusionTTS(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_channels=cfg.in_channels,
						   in_latent_channels=cfg.in_latent_channels, out_channels=cfg.out_channels, num_heads=cfg.num_heads)
	load_into(m, sd)
	M, T = 250, 250 * 4 * 24000 // 22050
	lat = torch.randn(1, M, 1024, generator=gen(11))
	dcond = torch.randn(1, 2048, generator=gen(12))
	noise = torch.randn(1, 100, T, generator=gen(14))
	with torch.inference_mode():
		E = m.timestep_independent(lat, dcond, T, False)
		keep = _ddim_loop_with_checkpoints(d_mod, m, noise, E)
	out = dict(T=np.int64(T), M=np.int64(M), stride=np.int64(8), steps=np.int64(80), checkpoints=np.array(LOOP_CHECKPOINTS, dtype=np.int64), mel=keep[80].numpy())
	for n in LOOP_CHECKPOINTS[:-1]:
		out[f"x_after_{n}_sub"] = keep[n][:, :, ::8].numpy()
	return out


def e2e_cfg1_case(d_mod, uv_mod):
	"""One configs[1] utterance end to end through the REFERENCE's modules in f32, in the order of inference.py:334-413: full-size UnifiedVoice and
	DiffusionTTS on the repo's seeded weights, SURVEY section 8d's inputs (64 text tokens seed 1234, conditioning latents seed 1235, start noise seed
	1236); 250 mel tokens drawn by the reference's own `sample_stream` (one candidate, stop token suppressed, CPU generator) -> forward(return_latent=True)
	-> timestep_independent -> the 80-step DDIM loop -> mel.  Stored: the codes, latents / E subsampled, x at the loop's checkpoints on every 8th frame, the final mel whole."""
	import importlib
	from transformers import GenerationConfig, LogitsProcessorList, SuppressTokensLogitsProcessor
	sg = importlib.import_module("tortoise_tts.models.stream_generator")
	acfg, dcfg = W.AR_FULL, W.DIFF_FULL
	ar = uv_mod.UnifiedVoice(layers=acfg.layers, model_dim=acfg.model_dim, heads=acfg.heads, checkpointing=False)
	load_into(ar, W.synth_state_dict(W.ar_shapes(acfg), 0))
	text = torch.randint(1, 255, (1, 64), generator=gen(1234))
	g = gen(1235)
	cond = torch.randn(1, 1024, generator=g)
	dcond = torch.randn(1, 2048, generator=g)
	M = 250
	t0 = time.time()
	with torch.inference_mode():
		ids = ar.compute_embeddings(cond, text)
		im = ar.inference_model
		procs = LogitsProcessorList([SuppressTokensLogitsProcessor([acfg.stop_mel_token])])
		warpers = sg.NewGenerationMixin._get_logits_warper(im, GenerationConfig(do_sample=True, num_beams=1, temperature=0.8, top_k=0, top_p=1.0))
		sg.setup_seed(0)
		toks = [tok.clone() for tok, _ in sg.NewGenerationMixin.sample_stream(
			im, ids, logits_processor=procs, logits_warper=warpers, stopping_criteria=_ScalarMaxLength(ids.shape[1] + M),
			pad_token_id=acfg.stop_mel_token, eos_token_id=acfg.stop_mel_token, output_hidden_states=True, return_dict_in_generate=False,
			use_cache=True, attention_mask=torch.ones_like(ids))]
		codes = torch.stack(toks, 1)
		assert codes.shape == (1, M) and int(codes.max()) < acfg.stop_mel_token
		print(f"  sampled {M} codes in {time.time() - t0:.0f}s: {codes[0, :8].tolist()} ...", flush=True)
		lat = ar.forward(cond, text, torch.tensor([64], dtype=torch.int32), codes, torch.tensor([M * acfg.mel_length_compression]),
						 return_latent=True, clip_inputs=False)
		del ar, im
		df = d_mod.DiffusionTTS(model_channels=dcfg.model_channels, num_layers=dcfg.num_layers, in_channels=dcfg.in_channels,
								in_latent_channels=dcfg.in_latent_channels, out_channels=dcfg.out_channels, num_heads=dcfg.num_heads)
		load_into(df, W.synth_state_dict(W.diffusion_shapes(dcfg), 1))
		T = lat.shape[1] * 4 * 24000 // 22050
		E = df.timestep_independent(lat, dcond, T, False)
		noise = torch.randn(1, 100, T, generator=gen(1236))
		keep = _ddim_loop_with_checkpoints(d_mod, df, noise, E)
	out = dict(T=np.int64(T), M=np.int64(M), stride=np.int64(8), codes=codes.numpy(), latents_sub=lat[:, :, ::8].numpy(), E_sub=E[:, :, ::8].numpy(),
			   checkpoints=np.array(LOOP_CHECKPOINTS, dtype=np.int64), mel=keep[80].numpy())
	for n in LOOP_CHECKPOINTS[:-1]:
		out[f"x_after_{n}_sub"] = keep[n][:, :, ::8].numpy()
	return out


def schedule_case(d_mod):
	out = {}
	for steps in (4, 30, 80, 200):
		df = d_mod.get_diffuser(steps=steps, cond_free=True)
		out[f"map_{steps}"] = np.array(df.timestep_map, dtype=np.int64)
		for name in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod",
					"sqrt_recipm1_alphas_cumprod", "posterior_log_variance_clipped", "posterior_mean_coef1",
					"posterior_mean_coef2"):
			out[f"{name}_{steps}"] = np.asarray(getattr(df, name), dtype=np.float64)
	return out


def lora_case(uv_mod, cfg, seed, rank, alpha):
	"""The reference's own LoRA attachment (models/lora.py apply_lora with the default parametrised pathway and the `gpt` policy,
	config.py:297-313) on a synthetic UnifiedVoice: the adapter tensors, the state_dict key names they get, the effective weights the
	reference's forward then uses, and prefill logits through the adapted model."""
	import importlib
	lora_mod = importlib.import_module("tortoise_tts.models.lora")
	sd = W.synth_state_dict(W.ar_shapes(cfg), seed)
	m = uv_mod.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, checkpointing=False)
	load_into(m, sd)
	m = lora_mod.apply_lora(m, rank=rank, alpha=alpha, policy=dict(include=["gpt"], exclude=[]), use_parametrize=True)
	m.eval()
	g = gen(seed + 7)
	out = dict(seed=np.int64(seed), rank=np.int64(rank), alpha=np.int64(alpha))
	with torch.no_grad():
		for name, p in m.named_parameters():
			if "lora_" in name:
				p.copy_(torch.randn(p.shape, generator=g) * 0.05)
	full = m.state_dict()
	lora, rest = lora_mod.lora_get_state_dict(full, split=True)
	out["lora_keys"] = np.array(sorted(lora.keys()))
	out["base_keys"] = np.array(sorted(k for k in rest.keys() if "parametrizations" in k))
	for k, v in lora.items():
		out["lora::" + k] = v.detach().numpy()
	for name, mod in m.named_modules():
		if hasattr(mod, "parametrizations"):
			out["eff::" + name + ".weight"] = mod.weight.detach().numpy().copy()
	Tt, B = 9, 2
	text = torch.randint(1, 255, (1, Tt), generator=gen(seed + 1))
	cond = torch.randn(1, cfg.model_dim, generator=gen(seed + 2))
	with torch.inference_mode():
		ids = m.compute_embeddings(cond, text).repeat(B, 1)
		r = m.inference_model.forward(input_ids=ids, attention_mask=torch.ones(B, ids.shape[1], dtype=torch.long), use_cache=True, return_dict=True)
	out.update(text=text.numpy(), cond=cond.numpy(), B=np.int64(B), prefill_logits=r.logits[:, -1].float().numpy())
	return out


def hf_sample_loop_case():
	"""ids of the INSTALLED HuggingFace generate() on the model-free stub of oracle/stub_lm.py (CPU generator), one entry per case"""
	import json
	import stub_lm
	import transformers
	out = dict(transformers_version=np.array(transformers.__version__))
	for name, seed, bias, B, N, kw in stub_lm.CASES:
		ids = stub_lm.hf_generate(stub_lm.make_table(seed, bias), B, N, kw)
		out["ids::" + name] = ids.numpy()
		out["kw::" + name] = np.array(json.dumps(kw))
	return out


def wrapper_case(uv_mod):
	"""The thin wrapper around the sampling loop: what the reference's `setup_seed(0)` leaves in the three host generators, and the fake
	id row `compute_embeddings` hands to generate()."""
	import importlib
	import random
	sg = importlib.import_module("tortoise_tts.models.stream_generator")
	out = {}
	for seed in (0, 7):
		sg.setup_seed(seed)
		out[f"torch_{seed}"] = torch.rand(4).numpy()
		out[f"numpy_{seed}"] = np.random.rand(4)
		out[f"python_{seed}"] = np.array([random.random() for _ in range(4)])
	torch.manual_seed(123); np.random.seed(123); random.seed(123)
	sg.setup_seed(-1)                                          # -1 leaves the generators alone
	out["torch_keep"] = torch.rand(4).numpy()
	cfg = W.AR_SMALL
	m = uv_mod.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, checkpointing=False)
	with torch.inference_mode():
		out["fake_ids"] = m.compute_embeddings(torch.zeros(2, cfg.model_dim), torch.randint(1, 255, (2, 5), generator=gen(1))).numpy()
	return out


SAMPLE_STREAM_CASES = (
	# name, weight seed, stop-token bias added to mel_head.bias, B, Tt, max new tokens, generation keywords
	("temperature", 11, 0.0, 3, 9, 10, dict(temperature=0.8, suppress_tokens=[8193])),
	("early_stop", 11, 4.5, 4, 7, 40, dict(temperature=0.8)),
	("cand16", 12, 0.0, 16, 5, 6, dict(temperature=1.0)),
	("warpers", 11, 0.0, 2, 6, 12, dict(temperature=0.7, top_k=16, top_p=0.9, repetition_penalty=2.0, suppress_tokens=[8193])),
	("stop_first", 11, 60.0, 2, 5, 8, dict(temperature=1.0)),
	("all_stop", 11, 7.0, 3, 7, 80, dict(temperature=0.8)),
	# a prompted continuation (inference_speech's input_tokens, unified_voice.py:651-656): 5 seeded mel tokens per row behind the fake prefix; "max new" counts them (:660)
	("prompted", 11, 0.0, 3, 8, 14, dict(temperature=0.8, repetition_penalty=1.5, suppress_tokens=[8193], prompt_len=5)),
)


class _ScalarMaxLength:
	"""`stopping_criteria(input_ids, scores)` as the transformers releases the reference targets answered it: ONE bool,
	`input_ids.shape[-1] >= max_length` (MaxLengthCriteria).  The installed 5.15 returns a per-row tensor there, whose truth value
	`sample_stream`'s `or` (stream_generator.py:1186) cannot take."""

	def __init__(self, max_length):
		self.max_length = max_length

	def __call__(self, input_ids, scores, **kw):
		return input_ids.shape[-1] >= self.max_length


def sample_stream_case(uv_mod):
	"""The reference's OWN sampling loop, `NewGenerationMixin.sample_stream` (stream_generator.py:911-1190), driven directly on the
	reference's GPT2InferenceModel: `generate()` cannot run on the installed transformers (AttributeError at :305), the loop can.  The
	processors / warpers are the HF classes `generate` would build (`_get_logits_processor`; `_get_logits_warper` :56-101, called here),
	max_length = trunc_index + max_generate_length (unified_voice.py:660), seed 0 (:296).  Stored per case: every yielded
	(tokens, latent) pair -- this pins a6's yield semantics (which hidden state goes with which token, whether the last token is
	yielded), the unfinished_sequences / padding rule, the stopping rule, and the CPU generator stream."""
	import importlib
	import json
	from transformers import (GenerationConfig, LogitsProcessorList, RepetitionPenaltyLogitsProcessor, StoppingCriteriaList,
							  SuppressTokensLogitsProcessor)
	sg = importlib.import_module("tortoise_tts.models.stream_generator")
	cfg = W.AR_SMALL
	out = {}
	for name, wseed, stop_bias, B, Tt, max_new, kw in SAMPLE_STREAM_CASES:
		sd = W.synth_state_dict(W.ar_shapes(cfg), wseed)
		if stop_bias:
			sd["mel_head.bias"] = sd["mel_head.bias"].clone()
			sd["mel_head.bias"][cfg.stop_mel_token] += stop_bias
		m = uv_mod.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, checkpointing=False)
		load_into(m, sd)
		text = torch.randint(1, 255, (1, Tt), generator=gen(wseed + 1))
		cond = torch.randn(1, cfg.model_dim, generator=gen(wseed + 2))
		prompt = None
		if kw.get("prompt_len"):
			prompt = torch.randint(0, 8192, (B, kw["prompt_len"]), generator=gen(wseed + 3))
		with torch.inference_mode():
			ids = m.compute_embeddings(cond, text).repeat(B, 1)
			n_fake = ids.shape[1]
			if prompt is not None:
				ids = torch.cat([ids, prompt], dim=1)
			im = m.inference_model
			procs = LogitsProcessorList()
			if kw.get("repetition_penalty", 1.0) != 1.0:
				procs.append(RepetitionPenaltyLogitsProcessor(penalty=kw["repetition_penalty"]))
			if kw.get("suppress_tokens"):
				procs.append(SuppressTokensLogitsProcessor(kw["suppress_tokens"]))
			gc = GenerationConfig(do_sample=True, num_beams=1, temperature=kw.get("temperature", 1.0), top_k=kw.get("top_k", 0), top_p=kw.get("top_p", 1.0))
			warpers = sg.NewGenerationMixin._get_logits_warper(im, gc)
			sg.setup_seed(0)
			toks, lats = [], []
			for tok, lat in sg.NewGenerationMixin.sample_stream(
					im, ids, logits_processor=procs, logits_warper=warpers,
					stopping_criteria=_ScalarMaxLength(n_fake + max_new), pad_token_id=cfg.stop_mel_token,
					eos_token_id=cfg.stop_mel_token, output_hidden_states=True, return_dict_in_generate=False, use_cache=True,
					attention_mask=torch.ones_like(ids)):
				toks.append(tok.clone())
				lats.append(lat.clone())
		out[f"{name}::text"], out[f"{name}::cond"] = text.numpy(), cond.numpy()
		if prompt is not None:
			out[f"{name}::prompt"] = prompt.numpy()
		out[f"{name}::ids"] = torch.stack(toks, 1).numpy()                      # [B, n]
		out[f"{name}::latents"] = torch.stack(lats, 1).numpy()                  # [B, n, d]
		out[f"{name}::meta"] = np.array(json.dumps(dict(weight_seed=wseed, stop_bias=stop_bias, B=B, Tt=Tt, max_new=max_new, kw=kw)))
		print(f"  {name}: {len(toks)} yields, ids[0] = {out[f'{name}::ids'][0].tolist()}")
	return out


# ------------------------------------------------------------------------------------------------------------------------------------
# The stress family: the same reference classes on `tortoise_tts_amd.weights.stress_*` weights -- peaked softmax rows,
# attention scores in the tens, outlier residual channels, a GroupNorm group of almost no variance, large scale / shift.
# ------------------------------------------------------------------------------------------------------------------------------------
STRESS_STREAM_CASES = (
	# name, B, Tt, max new tokens, generation keywords (top_k 16 / temperature 0.8 are the CLI's defaults, __main__.py:16-19; the other flags are the ones
	# it exposes :18-20 and unified_voice.py:633's typical_sampling / typical_mass)
	("topk16", 4, 9, 24, dict(temperature=0.8, top_k=16)),
	("topk16_topp_pen", 4, 9, 24, dict(temperature=0.8, top_k=16, top_p=0.8, repetition_penalty=2.0)),
	("topk16_typical", 4, 9, 24, dict(temperature=0.8, top_k=16, typical_mass=0.9)),
	("topp_only", 3, 7, 20, dict(temperature=1.0, top_p=0.8)),
	("typical_only", 3, 7, 20, dict(temperature=1.0, typical_mass=0.9, repetition_penalty=1.5)),
)


STRESS_LOGIT_COLS = list(range(0, 96)) + list(range(8100, 8194))
STRESS_ROW_STEPS = (1, 13)


def _stream_with_logits(uv_mod, m, cfg, cond, text, B, max_new, kw):
	"""The reference's sample_stream on model `m` with the processors / warpers generate() would build; the typical warper is the reference's OWN class
	(unified_voice.py:47-75), placed where generate() puts a caller's logits_processor: behind the default processors (HF `_merge_criteria_processor_list`),
	in front of the warpers.  Also returns the logits every token was drawn from (the model's output for the sequence the reference itself sampled)."""
	import importlib
	from transformers import GenerationConfig, LogitsProcessorList, RepetitionPenaltyLogitsProcessor, SuppressTokensLogitsProcessor
	sg = importlib.import_module("tortoise_tts.models.stream_generator")
	seen = []

	class _Tap:                                        # first "processor": records the raw next-token logits, changes nothing
		def __call__(self, input_ids, scores):
			seen.append(scores.clone())
			return scores
	with torch.inference_mode():
		ids = m.compute_embeddings(cond, text).repeat(B, 1)
		im = m.inference_model
		procs = LogitsProcessorList([_Tap()])
		if kw.get("repetition_penalty", 1.0) != 1.0:
			procs.append(RepetitionPenaltyLogitsProcessor(penalty=kw["repetition_penalty"]))
		if kw.get("suppress_tokens"):
			procs.append(SuppressTokensLogitsProcessor(kw["suppress_tokens"]))
		if kw.get("typical_mass") is not None:
			procs.append(uv_mod.TypicalLogitsWarper(mass=kw["typical_mass"]))
		gc = GenerationConfig(do_sample=True, num_beams=1, temperature=kw.get("temperature", 1.0), top_k=kw.get("top_k", 0), top_p=kw.get("top_p", 1.0))
		warpers = sg.NewGenerationMixin._get_logits_warper(im, gc)
		sg.setup_seed(0)
		toks, lats = [], []
		for tok, lat in sg.NewGenerationMixin.sample_stream(
				im, ids, logits_processor=procs, logits_warper=warpers, stopping_criteria=_ScalarMaxLength(ids.shape[1] + max_new),
				pad_token_id=cfg.stop_mel_token, eos_token_id=cfg.stop_mel_token, output_hidden_states=True, return_dict_in_generate=False,
				use_cache=True, attention_mask=torch.ones_like(ids)):
			toks.append(tok.clone())
			lats.append(lat.clone())
	return torch.stack(toks, 1), torch.stack(lats, 1), torch.stack(seen, 1)


def stress_ar_case(uv_mod):
	"""UnifiedVoice(small) on `stress_ar` weights, both variants: prefill + decode logits + latents as `ar_case` stores them, and the reference's own
	sample_stream under the CLI's warpers -- yielded ids, latents and the logits each token was drawn from."""
	import json
	cfg, seed = W.AR_SMALL, 14
	out = dict(seed=np.int64(seed))
	for variant in ("peaked", "outlier"):
		base = W.synth_state_dict(W.ar_shapes(cfg), seed)
		sd = W.stress_ar(base, cfg, variant)
		m = uv_mod.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, checkpointing=False)
		load_into(m, sd)
		B, Tt, n_dec, M = 2, 12, 4, 10
		text = torch.randint(1, 255, (1, Tt), generator=gen(seed + 1))
		cond = torch.randn(1, cfg.model_dim, generator=gen(seed + 2))
		dec_tokens = torch.randint(0, 8192, (B, n_dec), generator=gen(seed + 3))
		codes = torch.randint(0, 8192, (B, M), generator=gen(seed + 4))
		with torch.inference_mode():
			ids = m.compute_embeddings(cond, text).repeat(B, 1)
			im = m.inference_model
			P1 = ids.shape[1]
			r = im.forward(input_ids=ids, attention_mask=torch.ones(B, P1, dtype=torch.long), use_cache=True, return_dict=True)
			pre, past, dec = r.logits[:, -1].float(), r.past_key_values, []
			for k in range(1, n_dec + 1):
				r = im.forward(input_ids=dec_tokens[:, k - 1:k], past_key_values=past, attention_mask=torch.ones(B, P1 + k, dtype=torch.long), use_cache=True, return_dict=True)
				past = r.past_key_values
				dec.append(r.logits[:, -1].float())
			dec = torch.stack(dec, 1)
			lat = m.forward(cond.repeat(B, 1), text.repeat(B, 1), torch.tensor([Tt] * B, dtype=torch.int32), codes,
							torch.tensor([M * cfg.mel_length_compression] * B), return_latent=True, clip_inputs=False)
			# the REFERENCE'S OWN 16-bit deviation in this regime: the same calls under the autocast region inference.py:331 opens (bf16, on the CPU here)
			with torch.autocast("cpu", dtype=torch.bfloat16):
				r = im.forward(input_ids=ids, attention_mask=torch.ones(B, P1, dtype=torch.long), use_cache=True, return_dict=True)
				pre_amp, past, dec_amp = r.logits[:, -1].float(), r.past_key_values, []
				for k in range(1, n_dec + 1):
					r = im.forward(input_ids=dec_tokens[:, k - 1:k], past_key_values=past, attention_mask=torch.ones(B, P1 + k, dtype=torch.long), use_cache=True, return_dict=True)
					past = r.past_key_values
					dec_amp.append(r.logits[:, -1].float())
				dec_amp = torch.stack(dec_amp, 1)
		out.update({variant + "::prefill_logits_autocast_bf16": pre_amp.numpy(), variant + "::decode_logits_autocast_bf16": dec_amp.numpy()})
		print(f"  {variant}: the reference under autocast(bf16) vs itself in f32: rel L2 prefill {float((pre_amp - pre).norm() / pre.norm()):.3e}, decode {float((dec_amp - dec).norm() / dec.norm()):.3e}")
		pmax = torch.softmax(dec / 0.8, -1).max(-1)[0]
		print(f"  {variant}: logits std {float(dec.std()):.2f}, max token probability at T=0.8: median {float(pmax.median()):.3f}, max {float(pmax.max()):.3f}; |latent| max {float(lat.abs().max()):.1f}")
		p = variant + "::"
		out.update({p + "text": text.numpy(), p + "cond": cond.numpy(), p + "dec_tokens": dec_tokens.numpy(), p + "codes": codes.numpy(),
					p + "prefill_logits": pre.numpy(), p + "decode_logits": dec.numpy(), p + "latents": lat.numpy()})
		for name, B, Tt, max_new, kw in STRESS_STREAM_CASES:
			text = torch.randint(1, 255, (1, Tt), generator=gen(seed + 11))
			cond = torch.randn(1, cfg.model_dim, generator=gen(seed + 12))
			toks, lats, logits = _stream_with_logits(uv_mod, m, cfg, cond, text, B, max_new, kw)
			q = f"{variant}::{name}::"
			# the logits every token was drawn from: a vocabulary slice at every step, whole rows at two steps of the cases whose cut is a cumulative mass
			out.update({q + "text": text.numpy(), q + "cond": cond.numpy(), q + "ids": toks.numpy(), q + "latents": lats.numpy(), q + "logits_sub": logits[:, :, STRESS_LOGIT_COLS].numpy(),
						q + "meta": np.array(json.dumps(dict(B=B, Tt=Tt, max_new=max_new, kw=kw)))})
			if name in ("topk16_topp_pen", "typical_only"):
				out.update({q + "row_steps": np.array(STRESS_ROW_STEPS, dtype=np.int64), q + "logit_rows": logits[:, list(STRESS_ROW_STEPS)].numpy()})
			pm = torch.softmax(logits / kw.get("temperature", 1.0), -1).max(-1)[0]
			print(f"  {variant}/{name}: {toks.shape[1]} yields, max-prob median {float(pm.median()):.3f}, ids[0][:10] = {toks[0, :10].tolist()}")
	return out


def stress_ar_full_case(uv_mod):
	"""Full-size UnifiedVoice on the `outlier` stress weights: prefill + 2 decode steps (logit slices), a latent slice, and 12 tokens of the reference's
	sample_stream with top-k 16 + repetition penalty (ids, the logits they were drawn from on a vocabulary slice)."""
	import json
	cfg, seed = W.AR_FULL, 15
	sd = W.stress_ar(W.synth_state_dict(W.ar_shapes(cfg), seed), cfg, "outlier")
	m = uv_mod.UnifiedVoice(layers=cfg.layers, model_dim=cfg.model_dim, heads=cfg.heads, checkpointing=False)
	load_into(m, sd)
	B, Tt, n_dec, M = 2, 8, 2, 6
	text = torch.randint(1, 255, (1, Tt), generator=gen(seed + 1))
	cond = torch.randn(1, cfg.model_dim, generator=gen(seed + 2))
	dec_tokens = torch.randint(0, 8192, (B, n_dec), generator=gen(seed + 3))
	codes = torch.randint(0, 8192, (B, M), generator=gen(seed + 4))
	sel = torch.cat([torch.arange(0, 96), torch.arange(8100, 8194)])
	with torch.inference_mode():
		ids = m.compute_embeddings(cond, text).repeat(B, 1)
		im = m.inference_model
		P1 = ids.shape[1]
		r = im.forward(input_ids=ids, attention_mask=torch.ones(B, P1, dtype=torch.long), use_cache=True, return_dict=True)
		pre, past, dec = r.logits[:, -1].float(), r.past_key_values, []
		for k in range(1, n_dec + 1):
			r = im.forward(input_ids=dec_tokens[:, k - 1:k], past_key_values=past, attention_mask=torch.ones(B, P1 + k, dtype=torch.long), use_cache=True, return_dict=True)
			past = r.past_key_values
			dec.append(r.logits[:, -1].float())
		dec = torch.stack(dec, 1)
		lat = m.forward(cond.repeat(B, 1), text.repeat(B, 1), torch.tensor([Tt] * B, dtype=torch.int32), codes,
						torch.tensor([M * cfg.mel_length_compression] * B), return_latent=True, clip_inputs=False)
		with torch.autocast("cpu", dtype=torch.bfloat16):      # the reference's own 16-bit deviation (see stress_ar_case)
			r = im.forward(input_ids=ids, attention_mask=torch.ones(B, P1, dtype=torch.long), use_cache=True, return_dict=True)
			pre_amp, past, dec_amp = r.logits[:, -1].float(), r.past_key_values, []
			for k in range(1, n_dec + 1):
				r = im.forward(input_ids=dec_tokens[:, k - 1:k], past_key_values=past, attention_mask=torch.ones(B, P1 + k, dtype=torch.long), use_cache=True, return_dict=True)
				past = r.past_key_values
				dec_amp.append(r.logits[:, -1].float())
			dec_amp = torch.stack(dec_amp, 1)
	print(f"  full outlier: the reference under autocast(bf16) vs itself in f32: rel L2 prefill {float((pre_amp - pre).norm() / pre.norm()):.3e}, decode {float((dec_amp - dec).norm() / dec.norm()):.3e}")
	kw = dict(temperature=0.8, top_k=16, repetition_penalty=2.0)
	toks, lats, logits = _stream_with_logits(uv_mod, m, cfg, cond, text, 2, 12, kw)
	pm = torch.softmax(logits / 0.8, -1).max(-1)[0]
	print(f"  full outlier: logits std {float(dec.std()):.2f}, max-prob median {float(pm.median()):.3f}; ids[0] = {toks[0].tolist()}")
	return dict(seed=np.int64(seed), B=np.int64(B), text=text.numpy(), cond=cond.numpy(), dec_tokens=dec_tokens.numpy(), codes=codes.numpy(),
				logit_cols=sel.numpy(), prefill_logits=pre[:, sel].numpy(), decode_logits=dec[:, :, sel].numpy(), latents=lat[:, :, :128].numpy(),
				prefill_logits_autocast_bf16=pre_amp[:, sel].numpy(), decode_logits_autocast_bf16=dec_amp[:, :, sel].numpy(),
				stream_ids=toks.numpy(), stream_logits=logits[:, :, sel].numpy(), stream_latents=lats[:, :, :128].numpy(),
				stream_meta=np.array(json.dumps(dict(B=2, max_new=12, kw=kw))))


def _attn_score_probe(d_mod, m, x_in):
	"""largest |score + bias| inside the first main-layer AttentionBlock, for the log only"""
	import importlib
	au = importlib.import_module("tortoise_tts.models.arch_utils")
	blk = m.layers[0].attn
	with torch.inference_mode():
		qkv = blk.qkv(blk.norm(x_in))
		bs, width, length = qkv.shape
		ch = width // (3 * blk.num_heads)
		q, k, _ = qkv.reshape(bs * blk.num_heads, ch * 3, length).split(ch, dim=1)
		s = 1 / math.sqrt(math.sqrt(ch))
		wgt = torch.einsum("bct,bcs->bts", q * s, k * s)
		wgt = blk.relative_pos_embeddings(wgt.reshape(bs, blk.num_heads, length, length))
	return float(wgt.abs().max()), float(torch.softmax(wgt.float(), -1).max(-1)[0].median())


def stress_diff_case(d_mod):
	"""DiffusionTTS(small) on `stress_diffusion` weights: timestep_independent, one conditioned + one conditioning-free evaluation, and 8 DDIM steps
	(the reference's own loop, guidance ramp on) from seeded noise, x kept after 2 / 4 / 8 steps."""
	cfg, seed, b, M = W.DIFF_SMALL, 23, 2, 10
	sd = W.stress_diffusion(W.synth_state_dict(W.diffusion_shapes(cfg), seed), cfg)
	m = d_mod.DiffusionTTS(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_latent_channels=cfg.in_latent_channels, num_heads=cfg.num_heads)
	load_into(m, sd)
	T = M * 4 * 24000 // 22050
	lat = torch.randn(b, M, cfg.in_latent_channels, generator=gen(seed + 1))
	cond = torch.randn(b, 2 * cfg.model_channels, generator=gen(seed + 2))
	x = torch.randn(b, 100, T, generator=gen(seed + 3))
	t = torch.tensor([1333, 2666][:b])
	noise = torch.randn(1, 100, T, generator=gen(seed + 5))
	out = dict(latents=lat.numpy(), cond=cond.numpy(), x=x.numpy(), t=t.numpy(), T=np.int64(T), seed=np.int64(seed), noise=noise.numpy())
	with torch.inference_mode():
		E = m.timestep_independent(lat, cond, T, False)
		yc = m(x, t, precomputed_aligned_embeddings=E)
		yu = m(x, t, precomputed_aligned_embeddings=E, conditioning_free=True)
		# the reference's OWN 16-bit mode in this regime: DiffusionTTS(use_fp16=True) runs every main layer but the first under autocast (diffusion.py:1559-1561; bf16 is the
		# CPU autocast type).  (The outer autocast region of inference.py:331 cannot be used here: with use_fp16=False the model switches autocast OFF inside its layers and
		# then feeds them the bf16 tensor the region produced -- an ordinary dtype error of the reference's own code, CPU and GPU alike.)
		m.enable_fp16 = True
		yc_amp = m(x, t, precomputed_aligned_embeddings=E).float()
		m.enable_fp16 = False
		out.update(y_cond_ref_fp16mode=yc_amp.numpy())
		print(f"  small: the reference with use_fp16 (autocast bf16 in layers >= 1) vs itself in f32: rel L2 y_cond {float((yc_amp - yc).norm() / yc.norm()):.3e}")
		print(f"  small: |E| max {float(E.abs().max()):.1f}, |y| max {float(yc.abs().max()):.1f}, first-layer |score| max / median top weight: {_attn_score_probe(d_mod, m, E)}")
		diffuser = d_mod.get_diffuser(steps=8, cond_free=True)
		torch.manual_seed(0)
		done = 0
		for o in diffuser.ddim_sample_loop_progressive(m, (1, 100, T), noise=noise, clip_denoised=True, model_kwargs={"precomputed_aligned_embeddings": E[:1]},
														device="cpu", progress=False, eta=0.0):
			done += 1
			if done in (2, 4, 8):
				out[f"x_after_{done}"] = o["sample"].numpy().copy()
	out.update(E=E.numpy(), y_cond=yc.numpy(), y_uncond=yu.numpy())
	return out


def stress_diff_cfg1_case(d_mod):
	"""Full-size DiffusionTTS on `stress_diffusion` weights at configs[1]'s T = 1088: E, one evaluation pair (every 8th frame) and the LAST 4 steps of the
	80-step DDIM schedule from seeded x (final mel whole)."""
	cfg = W.DIFF_FULL
	sd = W.stress_diffusion(W.synth_state_dict(W.diffusion_shapes(cfg), 2), cfg)
	m = d_mod.DiffusionTTS(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_channels=cfg.in_channels,
						   in_latent_channels=cfg.in_latent_channels, out_channels=cfg.out_channels, num_heads=cfg.num_heads)
	load_into(m, sd)
	M, T = 250, 250 * 4 * 24000 // 22050
	lat = torch.randn(1, M, 1024, generator=gen(31))
	dcond = torch.randn(1, 2048, generator=gen(32))
	x = torch.randn(1, 100, T, generator=gen(33))
	t = torch.tensor([1500])
	diffuser = d_mod.get_diffuser(steps=80, cond_free=True)
	with torch.inference_mode():
		E = m.timestep_independent(lat, dcond, T, False)
		yc = m(x, t, precomputed_aligned_embeddings=E)
		yu = m(x, t, precomputed_aligned_embeddings=E, conditioning_free=True)
		m.enable_fp16 = True                    # the reference's own 16-bit mode (see stress_diff_case)
		yc_amp = m(x, t, precomputed_aligned_embeddings=E).float()
		m.enable_fp16 = False
		amp = dict(y_cond_ref_fp16mode_sub=yc_amp[:, :, ::8].numpy())
		print(f"  full: the reference with use_fp16 (autocast bf16 in layers >= 1) vs itself in f32: rel L2 y_cond {float((yc_amp - yc).norm() / yc.norm()):.3e}", flush=True)
		print(f"  full: |E| max {float(E.abs().max()):.1f}, |y| max {float(yc.abs().max()):.1f}, first-layer |score| max / median top weight: {_attn_score_probe(d_mod, m, E)}", flush=True)
		xm = x
		torch.manual_seed(0)
		for i in reversed(range(4)):
			xm = diffuser.ddim_sample(m, xm, torch.tensor([i]), clip_denoised=True, model_kwargs={"precomputed_aligned_embeddings": E}, eta=0.0)["sample"]
	return dict(T=np.int64(T), M=np.int64(M), stride=np.int64(8), E_sub=E[:, :, ::8].numpy(), y_cond_sub=yc[:, :, ::8].numpy(),
				y_uncond_sub=yu[:, :, ::8].numpy(), mel=xm.numpy(), **amp)


def stress_diff_cfg1_loop_case(d_mod):
	"""The WHOLE 80-step diffusion in the trained-checkpoint regime through the REFERENCE: stress_diff_cfg1's model (full-size
	`stress_diffusion` weights), latents and conditioning at T = 1088, `ddim_sample_loop_progressive` (diffusion.py:765-810) from seeded start noise, once in
	f32 (the reference's default, data/config.yaml:88-93) and once in the reference's OWN 16-bit mode (`enable_fp16`, diffusion.py:1559-1561; bf16 is the CPU
	autocast type) as the yardstick a 16-bit product mode is held against.  Stored: x after 8 / 16 / 40 / 72 steps on every 8th frame and the final mel whole,
	for both loops.  tests/test_gpu_stress.py regenerates the inputs from the same seeds."""
	cfg = W.DIFF_FULL
	sd = W.stress_diffusion(W.synth_state_dict(W.diffusion_shapes(cfg), 2), cfg)
	m = d_mod.DiffusionTTS(model_channels=cfg.model_channels, num_layers=cfg.num_layers, in_channels=cfg.in_channels,
						   in_latent_channels=cfg.in_latent_channels, out_channels=cfg.out_channels, num_heads=cfg.num_heads)
	load_into(m, sd)
	M, T = 250, 250 * 4 * 24000 // 22050
	lat = torch.randn(1, M, 1024, generator=gen(31))
	dcond = torch.randn(1, 2048, generator=gen(32))
	noise = torch.randn(1, 100, T, generator=gen(34))
	with torch.inference_mode():
		E = m.timestep_independent(lat, dcond, T, False)
		keep = _ddim_loop_with_checkpoints(d_mod, m, noise, E)
		m.enable_fp16 = True
		keep_amp = _ddim_loop_with_checkpoints(d_mod, m, noise, E)
		m.enable_fp16 = False
	out = dict(T=np.int64(T), M=np.int64(M), stride=np.int64(8), steps=np.int64(80), checkpoints=np.array(LOOP_CHECKPOINTS, dtype=np.int64),
			   mel=keep[80].numpy(), mel_ref_fp16mode=keep_amp[80].float().numpy())
	for n in LOOP_CHECKPOINTS:
		d = float((keep_amp[n].float() - keep[n]).norm() / keep[n].norm())
		print(f"  after {n} steps: the reference's 16-bit loop vs its f32 loop, rel L2 {d:.3e}", flush=True)
	for n in LOOP_CHECKPOINTS[:-1]:
		out[f"x_after_{n}_sub"] = keep[n][:, :, ::8].numpy()
		out[f"x_after_{n}_ref_fp16mode_sub"] = keep_amp[n][:, :, ::8].float().numpy()
	return out


def vocoder_case(cfg, seed, T):
	"""The reference BigVGAN generator (models/bigvgan.py) on synthetic weights: the anti-aliasing filter it builds, the weight-normed
	state_dict key names, one AMP block, `forward` internals and `inference` (waveform)."""
	import importlib
	bv = importlib.import_module("tortoise_tts.models.bigvgan")
	sd = W.synth_state_dict(W.vocoder_shapes(cfg), seed)
	torch.manual_seed(seed + 3)        # the weight-norm test tensors below are the constructor's own initialisation: seeded, so the fixture regenerates bit for bit
	m = bv.BigVGAN(data=cfg.as_json())
	out = dict(seed=np.int64(seed), wn_keys=np.array(sorted(k for k in m.state_dict().keys() if "filter" not in k)),
			   filter_up=m.activation_post.upsample.filter.reshape(-1).numpy().copy(),
			   filter_down=m.activation_post.downsample.lowpass.filter.reshape(-1).numpy().copy())
	# a weight-normed tensor pair for the ingest test (g, v and the weight they produce), before the norm is removed
	with torch.no_grad():
		gw = gen(seed + 2)
		m.conv_pre.weight_g.copy_(torch.rand(m.conv_pre.weight_g.shape, generator=gw) + 0.5)
		m.ups[0][0].weight_g.copy_(torch.rand(m.ups[0][0].weight_g.shape, generator=gw) + 0.5)
	out["wn_conv_pre_g"], out["wn_conv_pre_v"] = m.conv_pre.weight_g.detach().numpy().copy(), m.conv_pre.weight_v.detach().numpy().copy()
	out["wn_conv_pre_w"] = m.conv_pre.weight.detach().numpy().copy() if hasattr(m.conv_pre, "weight") else None
	out["wn_ups0_g"], out["wn_ups0_v"] = m.ups[0][0].weight_g.detach().numpy().copy(), m.ups[0][0].weight_v.detach().numpy().copy()
	m.remove_weight_norm()
	out["wn_conv_pre_w"] = m.conv_pre.weight.detach().numpy().copy()
	out["wn_ups0_w"] = m.ups[0][0].weight.detach().numpy().copy()
	missing, unexpected = m.load_state_dict(sd, strict=False)
	assert not unexpected and all("filter" in k for k in missing), (missing, unexpected)
	m.eval()
	mel = torch.randn(2, cfg.num_mels, T, generator=gen(seed + 1)) * 2.0 - 5.0
	out["mel"] = mel.numpy()
	with torch.inference_mode():
		x = m.conv_pre(mel)
		out["conv_pre"] = x.numpy().copy()
		x = m.ups[0][0](x)
		out["ups0"] = x.numpy().copy()
		out["act0"] = m.resblocks[0].activations[0](x).numpy().copy()
		out["amp0"] = m.resblocks[0](x).numpy().copy()
		out["forward"] = m.forward(mel, None).numpy().copy()
		out["audio"] = m.inference(mel).numpy().copy()
	return out


def clvp_case(cfg, seed):
	"""The reference CLVP (x-transformers branch) on synthetic weights: state_dict key names, one encoder's latent, the scores."""
	import importlib
	cl = importlib.import_module("tortoise_tts.models.clvp")
	sd = W.synth_state_dict(W.clvp_shapes(cfg), seed)
	m = cl.CLVP(dim_text=cfg.dim, dim_speech=cfg.dim, dim_latent=cfg.dim, num_text_tokens=cfg.num_text_tokens, text_enc_depth=cfg.depth,
				text_heads=cfg.heads, num_speech_tokens=cfg.num_speech_tokens, speech_enc_depth=cfg.depth, speech_heads=cfg.heads, use_xformers=True)
	missing, unexpected = m.load_state_dict(sd, strict=False)
	assert not unexpected and all("inv_freq" in k for k in missing), (missing, unexpected)
	m.eval()
	B, Tt, M = 5, 9, 23
	text = torch.randint(1, cfg.num_text_tokens, (1, Tt), generator=gen(seed + 1))
	codes = torch.randint(0, cfg.num_speech_tokens, (B, M), generator=gen(seed + 2))
	out = dict(seed=np.int64(seed), keys=np.array(sorted(k for k in m.state_dict().keys() if "inv_freq" not in k)), text=text.numpy(), codes=codes.numpy())
	with torch.inference_mode():
		out["scores"] = m(text.repeat(B, 1), codes, return_loss=False).numpy()
		enc = m.speech_transformer(m.speech_emb(codes), mask=torch.ones_like(codes).bool())
		out["speech_enc"] = enc.numpy()
	return out


def cond_case(d_mod, uv_mod, ar_cfg, diff_cfg, seed, b, n_clips, T_ar, T_diff, full):
	"""UnifiedVoice.get_conditioning / DiffusionTTS.get_conditioning on seeded synthetic weights (unified_voice.py:535-542,
	diffusion.py:1477-1485).  Only the sub-modules are built (full-size parents are not needed for their outputs)."""
	sd_ar = W.synth_state_dict(W.ar_conditioning_shapes(ar_cfg), seed)
	sd_df = W.synth_state_dict(W.diffusion_conditioning_shapes(diff_cfg), seed + 1)
	uv = uv_mod.UnifiedVoice(layers=1 if full else ar_cfg.layers, model_dim=ar_cfg.model_dim, heads=ar_cfg.heads, checkpointing=False)
	missing, unexpected = uv.load_state_dict(sd_ar, strict=False)
	assert not unexpected and not [k for k in missing if k.startswith("conditioning_encoder.")], unexpected
	df = d_mod.DiffusionTTS(model_channels=diff_cfg.model_channels, num_layers=1 if full else diff_cfg.num_layers,
							in_latent_channels=diff_cfg.in_latent_channels, num_heads=diff_cfg.num_heads)
	missing, unexpected = df.load_state_dict(sd_df, strict=False)
	assert not unexpected and not [k for k in missing if k.startswith("contextual_embedder.")], unexpected
	uv.eval(); df.eval()
	mel_ar = torch.randn(b, n_clips, 80, T_ar, generator=gen(seed + 2)) * 2 - 4
	mel_df = torch.randn(b, n_clips, diff_cfg.in_channels, T_diff, generator=gen(seed + 3)) * 2 - 4
	out = dict(seed=np.int64(seed), mel_ar=mel_ar.numpy(), mel_diff=mel_df.numpy())
	with torch.inference_mode():
		out["ar_latent"] = uv.get_conditioning(mel_ar).numpy()
		out["ar_latent_single"] = uv.get_conditioning(mel_ar[:, 0]).numpy()
		out["diff_latent"] = df.get_conditioning(mel_df).numpy()
		out["diff_latent_single"] = df.get_conditioning(mel_df[:, 0]).numpy()
		if not full:
			out["diff_embed"] = df.contextual_embedder(mel_df[:, 0]).numpy()
	return out


TOKENIZER_TEXTS = [
	"The quick brown fox jumps over the lazy dog.",
	"Hello, world!  This is   a test -- of the emergency broadcast system; please stand by...",
	"Mr. Smith and Mrs. Jones met Dr. Brown at St. Mary's, near Ft. Knox (Col. Mustard, Esq. was absent).",
	"\"Quoted\" text: isn't it?  Yes/no; a_b and x-y/z!",
	"supercalifragilisticexpialidocious antidisestablishmentarianism zzzz qqq xj",
	"UPPER lower MiXeD\ttabs\nnewlines  ",
	"",
	" ",
	"a",
	"???!!!...,,,",
	"[STOP] [UNK] [SPACE] [NOPE] ]x[ @#%^&*+=<>~`|{}",
]


def tokenizer_case():
	"""The reference's VoiceBpeTokenizer (tokenizer.py:154-177) on its own vocabulary (data/tokenizer.json), for ASCII texts without
	digits: `inflect` and `unidecode` are absent from this image, so they are stubbed at import (unidecode(ascii) is the identity by its
	specification; number_to_words is never reached for digit-free text -- the stub raises if it is).  Stores the vocabulary as arrays."""
	import importlib.util
	import json
	import types
	inflect = types.ModuleType("inflect")
	class _Engine:
		def number_to_words(self, *a, **k):
			raise AssertionError("digit-free fixtures only")
	inflect.engine = _Engine
	uni = types.ModuleType("unidecode")
	def unidecode(text):
		assert text.isascii(), "ASCII fixtures only"
		return text
	uni.unidecode = unidecode
	sys.modules.setdefault("inflect", inflect)
	sys.modules.setdefault("unidecode", uni)
	spec = importlib.util.spec_from_file_location("_ref_tokenizer", "/root/reference/tortoise_tts/tokenizer.py")
	mod = importlib.util.module_from_spec(spec)
	spec.loader.exec_module(mod)
	path = "/root/reference/data/tokenizer.json"
	tok = mod.VoiceBpeTokenizer(path)
	j = json.load(open(path))
	rng = np.random.default_rng(81)
	alphabet = list("abcdefghijklmnopqrstuvwxyz") + list("  ,.'-!?;:()/") + ["th", "the", "ing", "er", "ou", "and "]
	texts = list(TOKENIZER_TEXTS) + ["".join(rng.choice(alphabet, size=int(n))) for n in rng.integers(1, 200, size=40)]
	ids = [tok.encode(t) for t in texts]
	vocab = sorted(j["model"]["vocab"].items(), key=lambda kv: kv[1])
	assert [v for _, v in vocab] == list(range(len(vocab)))
	return dict(texts=np.array(texts), ids=np.array([i for row in ids for i in row], dtype=np.int64), offsets=np.cumsum([0] + [len(r) for r in ids]),
				cleaned=np.array([tok.preprocess_text(t) for t in texts]), decoded=np.array([tok.decode(np.array(r, dtype=np.int64)) for r in ids]),
				vocab=np.array([k for k, _ in vocab]), merges=np.array([m if isinstance(m, str) else " ".join(m) for m in j["model"]["merges"]]),
				special=np.array([a["content"] for a in j["added_tokens"]]))


def stft_case(seed):
	"""The reference's STFT.transform (arch_utils.py:560-623) -- the DFT-by-convolution TacotronSTFT sits on -- on seeded audio.
	`librosa.util.pad_center` is absent; with win_length == filter_length (the only use, arch_utils.py:676) it returns its input, so it
	is bound to that identity here.  Mel bases are not covered: librosa / torchaudio are absent (see oracle/mel_oracle.py)."""
	import importlib
	au = importlib.import_module("tortoise_tts.models.arch_utils")
	def pad_center(x, size):
		assert len(x) == size
		return x
	au.pad_center = pad_center
	stft = au.STFT(1024, 256, 1024)
	g = gen(seed)
	n = 6000
	t = torch.arange(n) / 24000.0
	y = 0.4 * torch.sin(2 * math.pi * 440 * t)[None] + 0.1 * torch.sin(2 * math.pi * 5300 * t)[None] + 0.05 * torch.randn(2, n, generator=g)
	with torch.inference_mode():
		mag, _ = stft.transform(y)
	return dict(seed=np.int64(seed), y=y.numpy(), magnitude=mag.numpy())


def main():
	os.makedirs(OUT, exist_ok=True)
	torch.set_num_threads(8)
	d_mod, uv_mod = ref_shim.load()
	jobs = [
		("schedule", lambda: schedule_case(d_mod)),
		("ar_small", lambda: ar_case(uv_mod, W.AR_SMALL, 11, B=2, Tt=12, n_dec=6, M=10, full=False)),
		("diff_small", lambda: diff_case(d_mod, W.DIFF_SMALL, 21, b=2, M=10, full=False)),
		("ar_full", lambda: ar_case(uv_mod, W.AR_FULL, 12, B=1, Tt=8, n_dec=2, M=6, full=True)),
		("diff_full", lambda: diff_case(d_mod, W.DIFF_FULL, 22, b=1, M=6, full=True)),
		("diff_cfg1", lambda: diff_cfg1_case(d_mod)),
		("diff_cfg1_loop", lambda: diff_cfg1_loop_case(d_mod)),
		("e2e_cfg1", lambda: e2e_cfg1_case(d_mod, uv_mod)),
		("lora_small", lambda: lora_case(uv_mod, W.AR_SMALL, 13, rank=4, alpha=8)),
		("hf_sample_loop", hf_sample_loop_case),
		("wrapper", lambda: wrapper_case(uv_mod)),
		("sample_stream", lambda: sample_stream_case(uv_mod)),
		("stress_ar", lambda: stress_ar_case(uv_mod)),
		("stress_ar_full", lambda: stress_ar_full_case(uv_mod)),
		("stress_diff", lambda: stress_diff_case(d_mod)),
		("stress_diff_cfg1", lambda: stress_diff_cfg1_case(d_mod)),
		("stress_diff_cfg1_loop", lambda: stress_diff_cfg1_loop_case(d_mod)),
		("vocoder_small", lambda: vocoder_case(W.VOC_SMALL, 51, T=13)),
		("clvp_small", lambda: clvp_case(W.CLVP_SMALL, 61)),
		("tokenizer", tokenizer_case),
		("stft_ref", lambda: stft_case(91)),
		("cond_small", lambda: cond_case(d_mod, uv_mod, W.AR_SMALL, W.DIFF_SMALL, 71, b=2, n_clips=2, T_ar=37, T_diff=45, full=False)),
		("cond_full", lambda: cond_case(d_mod, uv_mod, W.AR_FULL, W.DIFF_FULL, 72, b=1, n_clips=2, T_ar=70, T_diff=61, full=True)),
	]
	only = set(sys.argv[1:])
	for name, fn in jobs:
		if only and name not in only:
			continue
		t0 = time.time()
		data = fn()
		path = os.path.join(OUT, name + ".npz")
		np.savez_compressed(path, **data)
		print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB in {time.time() - t0:.1f}s")


if __name__ == "__main__":
	main()
