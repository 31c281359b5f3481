"""The BigVGAN oracle (oracle/bigvgan_oracle.py) against the reference generator run in the build container
(tests/golden/vocoder_small.npz, oracle/make_golden.py vocoder_small).  CPU only."""
import numpy as np
import torch

import bigvgan_oracle as BO
from tortoise_tts_amd import weights as W


def t(a):
	return torch.from_numpy(np.asarray(a))


def close(a, b, atol):
	err = (torch.as_tensor(a).double() - torch.as_tensor(b).double()).abs().max().item()
	assert err <= atol, f"max abs err {err:.3e}"


def _oracle(golden):
	g = golden("vocoder_small")
	sd = W.synth_state_dict(W.vocoder_shapes(W.VOC_SMALL), int(g["seed"]))
	return g, sd, BO.BigVGANOracle(sd, W.VOC_SMALL)


def test_antialiasing_filter_is_the_reference_buffer(golden):
	g = golden("vocoder_small")
	f = BO.aa_filter()
	assert f.shape == (12,) and np.array_equal(f.numpy(), g["filter_up"]) and np.array_equal(f.numpy(), g["filter_down"])     # bit for bit
	assert abs(float(f.sum()) - 1.0) < 1e-6


def test_state_dict_names_and_weight_norm_folding(golden):
	g, sd, _ = _oracle(golden)
	names = {str(k) for k in g["wn_keys"]}                      # what the reference model's state_dict holds (weight norm on)
	plain = {k[:-2] if k.endswith(("weight_g", "weight_v")) else k for k in names}
	plain = {k.replace(".weight_", ".weight") if k.endswith(".weight_") else k for k in plain}
	assert plain == set(W.vocoder_shapes(W.VOC_SMALL))
	folded = BO.fold_weight_norm({"conv_pre.weight_g": t(g["wn_conv_pre_g"]), "conv_pre.weight_v": t(g["wn_conv_pre_v"]),
								  "ups.0.0.weight_g": t(g["wn_ups0_g"]), "ups.0.0.weight_v": t(g["wn_ups0_v"]), "conv_pre.bias": torch.zeros(3)})
	assert set(folded) == {"conv_pre.weight", "ups.0.0.weight", "conv_pre.bias"}
	close(folded["conv_pre.weight"], g["wn_conv_pre_w"], 1e-6)
	close(folded["ups.0.0.weight"], g["wn_ups0_w"], 1e-6)       # ConvTranspose1d: the norm runs over [out, k] for each INPUT channel


def test_stages_and_waveform(golden):
	g, sd, voc = _oracle(golden)
	mel = t(g["mel"])
	cfg = W.VOC_SMALL
	with torch.inference_mode():
		x = torch.nn.functional.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
		close(x, g["conv_pre"], 1e-5)
		x = torch.nn.functional.conv_transpose1d(x, sd["ups.0.0.weight"], sd["ups.0.0.bias"], stride=4, padding=2)
		close(x, g["ups0"], 1e-5)
		close(voc._act(x, "resblocks.0.activations.0."), g["act0"], 1e-5)
		close(voc.amp_block(x, 0, cfg.resblock_kernel_sizes[0], cfg.resblock_dilation_sizes[0]), g["amp0"], 2e-5)
		close(voc.forward(mel), g["forward"], 2e-5)
		audio = voc.inference(mel)
	assert audio.shape == (2, 1, mel.shape[2] * cfg.hop_size) and float(audio.abs().max()) <= 1.0
	close(audio, g["audio"], 2e-5)
	assert float(t(g["audio"]).abs().max()) > 0.05              # the fixture is not a flat line


def test_resamplers_are_what_they_claim():
	f = BO.aa_filter()
	x = torch.ones(1, 3, 20)
	assert torch.allclose(BO.upsample2(x, f), torch.ones(1, 3, 40), atol=1e-6)        # DC gain 1 (replicate padding, x2, normalised filter)
	assert torch.allclose(BO.downsample2(torch.ones(1, 3, 40), f), torch.ones(1, 3, 20), atol=1e-6)
	assert BO.upsample2(torch.randn(2, 5, 7), f).shape == (2, 5, 14)
