"""Checkpoint ingest for the hot path: the reference's weight files -> the `state_dict` layout the
libttk handles consume (`weights.ar_shapes` / `weights.diffusion_shapes` name every tensor they read).

What the reference does, and what this module mirrors:

* `utils/io.py:106-127 torch_load`: `.pth`/`.pt` through `torch.load`; `.safetensors`/`.sft`/`.safetensor` through `safe_open`, the
  tensors wrapped as `{module_key: tensors} | json-decoded metadata`.
* `models/__init__.py:163-167 load_model`: `state_dict = torch.load(path)`, optionally `state_dict[state_dict_key]`, then
  `load_state_dict(strict=False)` for the autoregressive model (extra keys such as the conditioning encoder or HF's causal-mask
  buffers are ignored) and strict for the diffusion model.
* `inference.py:204-216` + `engines/__init__.py:102-104` + `models/lora.py:88-145`: a LoRA file holds the `lora_` tensors under the
  key `lora` (or `module`) and its `{rank, alpha}` under `config`; they are attached to every Linear / Conv1d / HF-Conv1D whose
  module name contains `gpt` as a weight *parametrization*, `W' = W + (lora_B @ lora_A).view(W.shape) * alpha / rank`
  (`ParameterizedLoRA.forward`; dropout is the identity in eval mode).  A parametrised module stores its tensors as
  `<m>.parametrizations.weight.original`, `<m>.parametrizations.weight.0.lora_A` ([rank, W.shape[1]]) and `...lora_B`
  ([W.shape[0], rank]).  The kernels want plain weights, so the adapters are folded in here once (`materialize_lora`), which is what
  the reference's forward computes on every call.

No CPU compute path hides in here: this is host-side file parsing and a few small matmuls at load time; the returned tensors go
straight to `UnifiedVoice` / `DiffusionTTS`, which upload and pack them on the device.
"""
from __future__ import annotations

import dataclasses
import json
import os
import re
from typing import Dict, Mapping, Optional, Tuple

import torch

from .weights import ARConfig, DiffusionConfig, ar_shapes, diffusion_shapes

SAFETENSORS_EXT = (".safetensor", ".safetensors", ".sft")
_PARAM_ORIG = ".parametrizations.weight.original"
_PARAM_LORA = re.compile(r"^(?P<mod>.+)\.parametrizations\.weight\.(?P<idx>\d+)\.lora_(?P<which>[AB])$")
_PLAIN_LORA = re.compile(r"^(?P<mod>.+)\.lora_(?P<which>[AB])$")


class CheckpointError(ValueError):
	pass


def _is_tensor_dict(d) -> bool:
	return isinstance(d, Mapping) and len(d) > 0 and all(isinstance(v, torch.Tensor) for v in d.values())


def read_checkpoint(path, module_key: str = "module"):
	"""The reference's `torch_load` (utils/io.py:106-127): returns whatever object the file holds for `.pth`, and
	`{module_key: tensors} | metadata` (values JSON-decoded where they parse) for safetensors.  Tensors land on the host."""
	path = os.fspath(path)
	if not os.path.exists(path):
		raise CheckpointError(f"checkpoint not found: {path}")
	if path.endswith(SAFETENSORS_EXT):
		from safetensors import safe_open
		tensors = {}
		with safe_open(path, framework="pt", device="cpu") as f:
			for k in f.keys():
				tensors[k] = f.get_tensor(k)
			metadata = dict(f.metadata() or {})
		for k, v in metadata.items():
			try:
				metadata[k] = json.loads(v)
			except Exception:
				pass
		return {module_key: tensors} | metadata
	try:
		return torch.load(path, map_location="cpu", weights_only=True)
	except Exception as e:          # pickled non-tensor payloads (old training checkpoints): the reference loads them with unsafe=True
		raise CheckpointError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {e}); re-save it as a plain "
							  f"state_dict or safetensors") from e


def unwrap_state_dict(obj, state_dict_key: Optional[str] = None) -> Dict[str, torch.Tensor]:
	"""Find the tensor dict inside a loaded checkpoint: `obj[state_dict_key]` when asked (models/__init__.py:164-165, e.g. 'generator'),
	the object itself when it already is one, else the usual wrappers ('module' from the trainer / safetensors path, 'state_dict',
	'model')."""
	if state_dict_key is not None:
		if not isinstance(obj, Mapping) or state_dict_key not in obj:
			raise CheckpointError(f"checkpoint has no key '{state_dict_key}'")
		obj = obj[state_dict_key]
	if _is_tensor_dict(obj):
		return dict(obj)
	if isinstance(obj, Mapping):
		for k in ("module", "state_dict", "model"):
			if k in obj and _is_tensor_dict(obj[k]):
				return dict(obj[k])
		# a state_dict with a few non-tensor entries (e.g. '_metadata'): keep the tensors
		tensors = {k: v for k, v in obj.items() if isinstance(v, torch.Tensor)}
		if tensors:
			return tensors
	raise CheckpointError("no tensor state_dict found in the checkpoint")


def split_lora(state_dict: Mapping[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor]]:
	"""`lora_get_state_dict(sd, split=True)` (models/lora.py:220-225): (adapter tensors, everything else)."""
	lora = {k: v for k, v in state_dict.items() if "lora_" in k}
	return lora, {k: v for k, v in state_dict.items() if "lora_" not in k}


def read_lora(path) -> Tuple[Dict[str, torch.Tensor], Optional[float]]:
	"""A LoRA file as `TTS` reads it (inference.py:214-215): tensors under 'lora' (else 'module'); scaling = alpha / rank from its
	'config' entry (config.py:122,140) when present."""
	obj = read_checkpoint(path)
	if isinstance(obj, Mapping) and not _is_tensor_dict(obj):
		tensors = obj.get("lora") if _is_tensor_dict(obj.get("lora")) else obj.get("module")
		if not _is_tensor_dict(tensors):
			raise CheckpointError(f"{path}: neither 'lora' nor 'module' holds a tensor dict")
		cfg = obj.get("config")
	else:
		tensors, cfg = obj, None
	scaling = None
	if isinstance(cfg, Mapping) and cfg.get("rank") and cfg.get("alpha") is not None:
		scaling = float(cfg["alpha"]) / float(cfg["rank"])
	lora, _ = split_lora(tensors)
	if not lora:
		raise CheckpointError(f"{path}: no 'lora_' tensors inside")
	return lora, scaling


def materialize_lora(state_dict: Mapping[str, torch.Tensor], lora: Optional[Mapping[str, torch.Tensor]] = None, *,
					 scaling: Optional[float] = None, alpha: Optional[float] = None) -> Dict[str, torch.Tensor]:
	"""Plain-weight state_dict with every adapter folded in.

	Handles, per adapted module `m`:
	  parametrised (the reference's default, models/lora.py:88-145): base `m.parametrizations.weight.original` or `m.weight`;
	      `W += (lora_B @ lora_A).view(W.shape) * s` for each `m.parametrizations.weight.<i>.lora_{A,B}`, in index order;
	  LoRALinear (models/lora.py:17-86): `m.lora_A [r, in]`, `m.lora_B [out, r]`, `W [out, in] += (lora_B @ lora_A) * s`.
	`s = scaling`, or `alpha / rank` with rank read off lora_A; the reference's default adapter has alpha == rank, i.e. s = 1
	(config.py:320-323), which is used when neither is given.  Adapter tensors may sit in `state_dict` itself or in `lora`."""
	merged = {}
	adapters: Dict[str, Dict[Tuple[int, str], torch.Tensor]] = {}
	sources = [state_dict] + ([lora] if lora else [])
	for src in sources:
		for k, v in src.items():
			m = _PARAM_LORA.match(k)
			if m:
				adapters.setdefault(m["mod"], {})[(int(m["idx"]), m["which"])] = v
				continue
			m = _PLAIN_LORA.match(k)
			if m:
				adapters.setdefault(m["mod"], {})[(-1, m["which"])] = v
				continue
			if src is state_dict:
				if k.endswith(_PARAM_ORIG):
					merged[k[:-len(_PARAM_ORIG)] + ".weight"] = v
				else:
					merged[k] = v
	for mod, parts in adapters.items():
		wkey = mod + ".weight"
		if wkey not in merged:
			raise CheckpointError(f"LoRA tensors for '{mod}' but the checkpoint has no '{wkey}'")
		W = merged[wkey].detach().to(torch.float32).clone()
		for idx in sorted({i for i, _ in parts}):
			A, Bm = parts.get((idx, "A")), parts.get((idx, "B"))
			if A is None or Bm is None:
				raise CheckpointError(f"'{mod}': lora_A / lora_B are not both present")
			A, Bm = A.detach().to(torch.float32), Bm.detach().to(torch.float32)
			rank = A.shape[0]
			if Bm.shape[1] != rank or Bm.shape[0] * A.shape[1] != W.numel():
				raise CheckpointError(f"'{mod}': lora shapes {tuple(Bm.shape)} x {tuple(A.shape)} do not match weight {tuple(W.shape)}")
			s = scaling if scaling is not None else (float(alpha) / rank if alpha is not None else 1.0)
			W += (Bm @ A).view(W.shape) * s
		merged[wkey] = W
	return merged


def _count_layers(sd: Mapping[str, torch.Tensor], pattern: str) -> int:
	rx = re.compile(pattern)
	idx = {int(m.group(1)) for k in sd for m in [rx.match(k)] if m}
	return max(idx) + 1 if idx else 0


def infer_ar_config(sd: Mapping[str, torch.Tensor], base: ARConfig = ARConfig()) -> ARConfig:
	"""Model sizes read off the tensors, inverting how the constructor arguments size them (unified_voice.py:337-350, 405-416):
	text_embedding has number_text_tokens + 1 rows, the position tables max_*_tokens + 2 (+ max_conditioning_inputs for mel) rows.
	Heads follow from the kernels' fixed head width of 64 (the reference's 1024 / 16)."""
	try:
		d = sd["mel_embedding.weight"].shape[1]
		return dataclasses.replace(
			base, layers=_count_layers(sd, r"^gpt\.h\.(\d+)\.ln_1\.weight$"), model_dim=d, heads=max(1, d // 64),
			number_mel_codes=sd["mel_embedding.weight"].shape[0], number_text_tokens=sd["text_embedding.weight"].shape[0] - 1,
			max_mel_tokens=sd["mel_pos_embedding.emb.weight"].shape[0] - 2 - base.max_conditioning_inputs,
			max_text_tokens=sd["text_pos_embedding.emb.weight"].shape[0] - 2)
	except KeyError as e:
		raise CheckpointError(f"not an autoregressive (UnifiedVoice) state_dict: missing {e}") from e


def infer_diffusion_config(sd: Mapping[str, torch.Tensor], base: DiffusionConfig = DiffusionConfig()) -> DiffusionConfig:
	"""diffusion.py:1390-1404: `layers` holds num_layers DiffusionLayers followed by 3 ResBlocks."""
	try:
		C = sd["inp_block.weight"].shape[0]
		return dataclasses.replace(base, model_channels=C, in_channels=sd["inp_block.weight"].shape[1], num_heads=max(1, C // 64),
								   num_layers=_count_layers(sd, r"^layers\.(\d+)\.") - 3, out_channels=sd["out.2.weight"].shape[0],
								   in_latent_channels=sd["latent_conditioner.0.weight"].shape[1])
	except KeyError as e:
		raise CheckpointError(f"not a DiffusionTTS state_dict: missing {e}") from e


def select_hot_path(sd: Mapping[str, torch.Tensor], shapes: Mapping[str, Tuple[int, ...]], what: str) -> Dict[str, torch.Tensor]:
	"""The tensors the handle reads, shape-checked; everything else in the file is ignored (the reference loads the autoregressive
	checkpoint with strict=False, models/__init__.py:104,167).  A missing or mis-shaped tensor is an error naming all of them."""
	out, problems = {}, []
	for name, shape in shapes.items():
		t = sd.get(name)
		if t is None:
			problems.append(f"missing {name}")
		elif tuple(t.shape) != tuple(shape):
			problems.append(f"{name}: shape {tuple(t.shape)} != expected {tuple(shape)}")
		else:
			out[name] = t.detach().to(torch.float32).contiguous()
	if problems:
		more = f" (+{len(problems) - 8} more)" if len(problems) > 8 else ""
		raise CheckpointError(f"{what} checkpoint does not fit the hot path: " + "; ".join(problems[:8]) + more)
	return out


def load_autoregressive_state(path, lora_path=None, *, cfg: Optional[ARConfig] = None, lora_scaling: Optional[float] = None,
							  state_dict_key: Optional[str] = None) -> Tuple[Dict[str, torch.Tensor], ARConfig]:
	"""`autoregressive.pth` (+ optional LoRA file) -> (hot-path state_dict with adapters folded in, config)."""
	sd = unwrap_state_dict(read_checkpoint(path), state_dict_key)
	lora = None
	if lora_path is not None:
		lora, file_scaling = read_lora(lora_path)
		lora_scaling = lora_scaling if lora_scaling is not None else file_scaling
	sd = materialize_lora(sd, lora, scaling=lora_scaling)
	cfg = cfg or infer_ar_config(sd)
	return select_hot_path(sd, ar_shapes(cfg), "autoregressive"), cfg


def load_diffusion_state(path, *, cfg: Optional[DiffusionConfig] = None, state_dict_key: Optional[str] = None
						 ) -> Tuple[Dict[str, torch.Tensor], DiffusionConfig]:
	sd = unwrap_state_dict(read_checkpoint(path), state_dict_key)
	sd = materialize_lora(sd)          # the reference only adapts modules under `gpt`; this just normalises parametrised keys
	cfg = cfg or infer_diffusion_config(sd)
	return select_hot_path(sd, diffusion_shapes(cfg), "diffusion"), cfg


def load_autoregressive(path, lora_path=None, *, dtype="bf16", device="cuda", max_batch=16, max_ctx=None, **kw):
	"""The counterpart of `load_model("autoregressive")` + the LoRA block of `TTS.__init__` (models/__init__.py:104-110,163-167;
	inference.py:204-216) returning the libttk-backed `UnifiedVoice`."""
	from .autoregressive import UnifiedVoice
	sd, cfg = load_autoregressive_state(path, lora_path, **kw)
	extra = {} if max_ctx is None else {"max_ctx": max_ctx}
	return UnifiedVoice(sd, cfg, dtype=dtype, device=device, max_batch=max_batch, **extra)


def load_diffusion(path, *, dtype="bf16", device="cuda", **kw):
	from .diffusion import DiffusionTTS
	sd, cfg = load_diffusion_state(path, **kw)
	return DiffusionTTS(sd, cfg, dtype=dtype, device=device)


def load_conditioning_encoder(path, lora_path=None, *, dtype="bf16", device="cuda", cfg: Optional[ARConfig] = None, state_dict_key: Optional[str] = None):
	"""The `conditioning_encoder.*` tensors of `autoregressive.pth` -> `ConditioningEncoder` (`UnifiedVoice.get_conditioning`,
	unified_voice.py:535-542).  The reference's LoRA only adapts modules under `gpt`; a LoRA file is accepted for symmetry and folded."""
	from .conditioning import ConditioningEncoder
	from .weights import ar_conditioning_shapes
	sd = unwrap_state_dict(read_checkpoint(path), state_dict_key)
	lora, scaling = read_lora(lora_path) if lora_path is not None else (None, None)
	sd = materialize_lora(sd, lora, scaling=scaling)
	cfg = cfg or infer_ar_config(sd)
	n_blocks = _count_layers(sd, r"^conditioning_encoder\.attn\.(\d+)\.")
	spec_dim = sd["conditioning_encoder.init.weight"].shape[1] if "conditioning_encoder.init.weight" in sd else 80
	picked = select_hot_path(sd, ar_conditioning_shapes(cfg, spec_dim, n_blocks), "autoregressive (conditioning_encoder)")
	return ConditioningEncoder(picked, cfg, dtype=dtype, device=device, spec_dim=spec_dim, attn_blocks=n_blocks)


def load_contextual_embedder(path, *, dtype="bf16", device="cuda", cfg: Optional[DiffusionConfig] = None, state_dict_key: Optional[str] = None):
	"""The `contextual_embedder.*` tensors of `diffusion.pth` -> `ContextualEmbedder` (`DiffusionTTS.get_conditioning`, diffusion.py:1477-1485)."""
	from .conditioning import ContextualEmbedder
	from .weights import diffusion_conditioning_shapes
	sd = materialize_lora(unwrap_state_dict(read_checkpoint(path), state_dict_key))
	cfg = cfg or infer_diffusion_config(sd)
	return ContextualEmbedder(select_hot_path(sd, diffusion_conditioning_shapes(cfg), "diffusion (contextual_embedder)"), cfg, dtype=dtype, device=device)


def load_bigvgan(path, *, cfg=None, dtype="bf16", device="cuda", state_dict_key: Optional[str] = "generator"):
	"""`load_model("bigvgan")` (models/__init__.py:128-140): the generator's tensors sit under 'generator' in the upstream file; weight norm
	is folded and the config defaults to the published bigvgan_24khz_100band values (weights.VocoderConfig: an assumption, the JSON is a
	download)."""
	from .vocoder import BigVGAN
	from .weights import VocoderConfig
	obj = read_checkpoint(path)
	if state_dict_key is not None and not (isinstance(obj, Mapping) and state_dict_key in obj):
		state_dict_key = None                      # a bare generator state_dict
	sd = unwrap_state_dict(obj, state_dict_key)
	return BigVGAN(sd, cfg or VocoderConfig(), dtype=dtype, device=device)


def load_clvp(path, *, cfg=None, dtype="bf16", device="cuda", state_dict_key: Optional[str] = None):
	"""`load_model("clvp")` (models/__init__.py:111-113): `clvp2.pth` is a plain state_dict of the x-transformers CLVP."""
	from .clvp import CLVP
	from .weights import CLVPConfig, clvp_shapes
	sd = unwrap_state_dict(read_checkpoint(path), state_dict_key)
	cfg = cfg or CLVPConfig()
	return CLVP(select_hot_path(sd, clvp_shapes(cfg), "clvp"), cfg, dtype=dtype, device=device)


def save_state_dict(state_dict: Mapping[str, torch.Tensor], path, metadata: Optional[Mapping[str, object]] = None):
	"""`torch_save` (utils/io.py:92-104) for a plain tensor dict: safetensors (metadata JSON-encoded) or `.pth` by extension."""
	path = os.fspath(path)
	if path.endswith(SAFETENSORS_EXT):
		from safetensors.torch import save_file
		md = {k: (v if isinstance(v, str) else json.dumps(v)) for k, v in (metadata or {}).items() if v is not None}
		return save_file({k: v.contiguous() for k, v in state_dict.items()}, path, md)
	obj = dict(state_dict) if not metadata else {"module": dict(state_dict), **metadata}
	return torch.save(obj, path)
