"""Checkpoint ingest (tortoise_tts_amd/checkpoint.py) against the reference's own file conventions and its own
LoRA attachment (tests/golden/lora_small.npz, produced by oracle/make_golden.py running models/lora.py apply_lora)."""
import numpy as np
import pytest
import torch

import tortoise_oracle as O
from tortoise_tts_amd import checkpoint as ck
from tortoise_tts_amd import weights as W


def t(a):
	return torch.from_numpy(np.asarray(a))


def _base(golden):
	g = golden("lora_small")
	return g, W.synth_state_dict(W.ar_shapes(W.AR_SMALL), int(g["seed"]))


def _lora_tensors(g):
	return {str(k): t(g["lora::" + str(k)]) for k in g["lora_keys"]}


def _parametrised(sd, g):
	"""the state_dict an adapted reference model saves: base weights renamed to `...parametrizations.weight.original`"""
	out = dict(sd)
	for k in g["base_keys"]:
		k = str(k)
		out[k] = out.pop(k[:-len(".parametrizations.weight.original")] + ".weight")
	return out


def test_materialize_lora_matches_reference_effective_weights(golden):
	g, sd = _base(golden)
	lora = _lora_tensors(g)
	scaling = float(g["alpha"]) / float(g["rank"])
	eff = {k[5:]: t(g[k]) for k in g if k.startswith("eff::")}
	assert len(eff) == 4 * W.AR_SMALL.layers
	# adapters in a separate file (the inference path), on plain and on parametrised base naming; and all in one state_dict
	for merged in (ck.materialize_lora(sd, lora, scaling=scaling), ck.materialize_lora(_parametrised(sd, g), lora, scaling=scaling),
				   ck.materialize_lora(_parametrised(sd, g) | lora, alpha=float(g["alpha"]))):
		assert not any("lora_" in k or "parametrizations" in k for k in merged)
		assert set(merged) == set(sd)
		for k, v in eff.items():
			assert (merged[k] - v).abs().max().item() <= 1e-6, k
		untouched = [k for k in sd if k not in eff]
		assert all(torch.equal(merged[k], sd[k]) for k in untouched)
	# default scaling is alpha == rank (config.py:320-323) => 1.0; here that must differ from the fixture's 2.0
	assert (ck.materialize_lora(sd, lora)["gpt.h.0.attn.c_attn.weight"] - eff["gpt.h.0.attn.c_attn.weight"]).abs().max() > 1e-3


def test_lora_key_names_are_the_reference_ones(golden):
	g, _ = _base(golden)
	keys = [str(k) for k in g["lora_keys"]]
	assert all(ck._PARAM_LORA.match(k) for k in keys) and len(keys) == 8 * W.AR_SMALL.layers
	lora, rest = ck.split_lora({k: torch.zeros(1) for k in keys} | {"gpt.ln_f.weight": torch.zeros(1)})
	assert set(lora) == set(keys) and set(rest) == {"gpt.ln_f.weight"}


def test_oracle_on_merged_weights_reproduces_adapted_reference_logits(golden):
	g, sd = _base(golden)
	merged = ck.materialize_lora(sd, _lora_tensors(g), scaling=float(g["alpha"]) / float(g["rank"]))
	ar = O.AROracle(ck.select_hot_path(merged, W.ar_shapes(W.AR_SMALL), "autoregressive"), W.AR_SMALL)
	with torch.inference_mode():
		logits, _, _ = ar.prefill(ar.prefix_embeddings(t(g["cond"]), t(g["text"])), int(g["B"]))
	assert (logits[:, -1] - t(g["prefill_logits"])).abs().max().item() <= 2e-5
	# and the adapters matter: the un-adapted model is far away
	ar0 = O.AROracle(sd, W.AR_SMALL)
	with torch.inference_mode():
		l0, _, _ = ar0.prefill(ar0.prefix_embeddings(t(g["cond"]), t(g["text"])), int(g["B"]))
	assert (l0[:, -1] - t(g["prefill_logits"])).abs().max().item() > 1e-2


@pytest.mark.parametrize("ext", [".pth", ".safetensors", ".sft"])
def test_file_round_trip_and_wrappers(tmp_path, ext):
	sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 3)
	p = tmp_path / ("diffusion" + ext)
	ck.save_state_dict(sd, p, metadata={"config": {"rank": 4, "alpha": 8}, "note": "x"})
	obj = ck.read_checkpoint(p)
	assert obj["config"] == {"rank": 4, "alpha": 8} and obj["note"] == "x"          # JSON-decoded like utils/io.py:117-123
	got = ck.unwrap_state_dict(obj)
	assert set(got) == set(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
	state, cfg = ck.load_diffusion_state(p)
	assert cfg == W.DIFF_SMALL and set(state) == set(sd)


def test_plain_pth_nested_key_and_errors(tmp_path):
	sd = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 5)
	extras = {"conditioning_encoder.init.weight": torch.zeros(4, 4), "gpt.h.0.attn.bias": torch.ones(1, 1, 8, 8), "text_head.weight": torch.zeros(3, 3)}
	torch.save({"generator": sd | extras}, tmp_path / "nested.pth")
	state, cfg = ck.load_autoregressive_state(tmp_path / "nested.pth", state_dict_key="generator")
	assert cfg == W.AR_SMALL and set(state) == set(sd)                                # strict=False: extras dropped
	with pytest.raises(ck.CheckpointError, match="no key"):
		ck.load_autoregressive_state(tmp_path / "nested.pth", state_dict_key="model_g")
	with pytest.raises(ck.CheckpointError, match="not found"):
		ck.read_checkpoint(tmp_path / "absent.pth")
	broken = dict(sd)
	del broken["gpt.h.1.mlp.c_fc.bias"]
	broken["mel_head.bias"] = torch.zeros(7)
	torch.save(broken, tmp_path / "broken.pth")
	with pytest.raises(ck.CheckpointError) as e:
		ck.load_autoregressive_state(tmp_path / "broken.pth", cfg=W.AR_SMALL)
	assert "missing gpt.h.1.mlp.c_fc.bias" in str(e.value) and "mel_head.bias" in str(e.value)
	with pytest.raises(ck.CheckpointError, match="not a DiffusionTTS"):
		ck.load_diffusion_state(tmp_path / "broken.pth")
	torch.save({"lora": {"gpt.h.0.attn.c_attn.parametrizations.weight.0.lora_A": torch.zeros(2, 384)}}, tmp_path / "half.pth")
	with pytest.raises(ck.CheckpointError, match="not both present"):
		ck.load_autoregressive_state(tmp_path / "nested.pth", tmp_path / "half.pth", state_dict_key="generator")


def test_lora_file_with_config_sets_scaling(golden, tmp_path):
	g, sd = _base(golden)
	torch.save(sd, tmp_path / "autoregressive.pth")
	lora = _lora_tensors(g)
	# the trainer's layout (engines/base.py:145-160): {'lora': tensors, 'config': {...}}; and the safetensors flavour of it
	torch.save({"lora": lora, "config": {"name": "lora", "rank": int(g["rank"]), "alpha": int(g["alpha"])}}, tmp_path / "lora.pth")
	ck.save_state_dict(lora, tmp_path / "lora.sft", metadata={"config": {"rank": int(g["rank"]), "alpha": int(g["alpha"])}})
	for lp in ("lora.pth", "lora.sft"):
		tensors, scaling = ck.read_lora(tmp_path / lp)
		assert scaling == 2.0 and set(tensors) == set(lora)
		state, cfg = ck.load_autoregressive_state(tmp_path / "autoregressive.pth", tmp_path / lp)
		assert cfg == W.AR_SMALL
		assert (state["gpt.h.1.mlp.c_proj.weight"] - t(g["eff::gpt.h.1.mlp.c_proj.weight"])).abs().max().item() <= 1e-6


@pytest.mark.gpu
def test_gpu_handle_from_checkpoint_files_with_lora(golden, tmp_path):
	"""files -> load_autoregressive -> libttk prefill == the adapted REFERENCE model's logits (f32 mode)."""
	g, sd = _base(golden)
	ck.save_state_dict(sd, tmp_path / "autoregressive.safetensors")
	torch.save({"lora": _lora_tensors(g), "config": {"rank": int(g["rank"]), "alpha": int(g["alpha"])}}, tmp_path / "lora.pth")
	ar = ck.load_autoregressive(tmp_path / "autoregressive.safetensors", tmp_path / "lora.pth", dtype="f32", device="cuda:0", max_batch=4)
	assert ar.cfg == W.AR_SMALL
	with torch.inference_mode():
		logits = ar._prefill(t(g["cond"]), t(g["text"]), int(g["B"]))
	assert (logits.cpu() - t(g["prefill_logits"])).abs().max().item() <= 2e-4


def test_vocoder_weight_norm_folding_equals_reference(golden):
	"""product-side folding (tortoise_tts_amd/vocoder.py) of the checkpoint's weight_g / weight_v pairs, both spellings, against the weights
	the reference's remove_weight_norm() produced (Conv1d: norm over [in, k] per output channel; ConvTranspose1d: over [out, k] per INPUT channel)"""
	from tortoise_tts_amd.vocoder import aa_filter, fold_weight_norm
	g = golden("vocoder_small")
	sd = {"conv_pre.weight_g": t(g["wn_conv_pre_g"]), "conv_pre.weight_v": t(g["wn_conv_pre_v"]), "conv_pre.bias": torch.zeros(4),
		  "ups.0.0.parametrizations.weight.original0": t(g["wn_ups0_g"]), "ups.0.0.parametrizations.weight.original1": t(g["wn_ups0_v"]),
		  "activation_post.upsample.filter": torch.zeros(1, 1, 12)}
	out = fold_weight_norm(sd)
	assert set(out) == {"conv_pre.weight", "conv_pre.bias", "ups.0.0.weight"}
	assert (out["conv_pre.weight"] - t(g["wn_conv_pre_w"])).abs().max().item() <= 1e-6
	assert (out["ups.0.0.weight"] - t(g["wn_ups0_w"])).abs().max().item() <= 1e-6
	assert np.array_equal(aa_filter().numpy(), g["filter_up"])          # the product's own filter restatement, bit for bit


def test_clvp_packing_and_shape_selection(golden):
	from tortoise_tts_amd.clvp import pack_state_dict
	cfg = W.CLVP_SMALL
	sd = W.synth_state_dict(W.clvp_shapes(cfg), 3)
	packed = pack_state_dict(sd | {"text_transformer.transformer.attn_layers.rotary_pos_emb.inv_freq": torch.zeros(16)}, cfg)
	p = "speech_transformer.transformer.attn_layers.layers.2.1.wrap."
	assert torch.equal(packed[p + "__qkv.weight"], torch.cat([sd[p + "to_q.weight"], sd[p + "to_k.weight"], sd[p + "to_v.weight"]]))
	assert not any("inv_freq" in k and not k.startswith("__") for k in packed) and packed["temperature"].shape == (1,)
	want = 1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32))
	assert torch.equal(packed["__rotary_inv_freq"], want) and packed["__rotary_inv_freq"].shape == (16,)
	# select_hot_path on a CLVP file with a missing tensor names it
	broken = {k: v for k, v in sd.items() if k != "to_speech_latent.weight"}
	with pytest.raises(ck.CheckpointError, match="to_speech_latent.weight"):
		ck.select_hot_path(broken, W.clvp_shapes(cfg), "clvp")


@pytest.mark.gpu
def test_gpu_conditioning_encoders_from_checkpoint_files(golden, tmp_path):
	"""whole-model files (hot-path + conditioning tensors side by side, wrapped under 'module') -> load_conditioning_encoder /
	load_contextual_embedder -> the REFERENCE's get_conditioning outputs (tests/golden/cond_small.npz)"""
	g = golden("cond_small")
	seed = int(g["seed"])
	ar_sd = W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 5) | W.synth_state_dict(W.ar_conditioning_shapes(W.AR_SMALL), seed)
	df_sd = W.synth_state_dict(W.diffusion_shapes(W.DIFF_SMALL), 6) | W.synth_state_dict(W.diffusion_conditioning_shapes(W.DIFF_SMALL), seed + 1)
	torch.save({"module": ar_sd, "config": {}}, tmp_path / "autoregressive.pth")
	ck.save_state_dict(df_sd, tmp_path / "diffusion.safetensors")
	enc = ck.load_conditioning_encoder(tmp_path / "autoregressive.pth", dtype="f32", device="cuda:0")
	ctx = ck.load_contextual_embedder(tmp_path / "diffusion.safetensors", dtype="f32", device="cuda:0")
	a = enc.get_conditioning(t(g["mel_ar"]).to("cuda:0"))
	d = ctx.get_conditioning(t(g["mel_diff"]).to("cuda:0"))
	assert (a.cpu() - t(g["ar_latent"])).abs().max().item() < 5e-4 and (d.cpu() - t(g["diff_latent"])).abs().max().item() < 5e-4
	with pytest.raises(ck.CheckpointError, match="conditioning_encoder"):
		ck.save_state_dict(W.synth_state_dict(W.ar_shapes(W.AR_SMALL), 5), tmp_path / "bare.safetensors")
		ck.load_conditioning_encoder(tmp_path / "bare.safetensors", dtype="f32", device="cuda:0")
