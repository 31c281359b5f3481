"""Drop-in for the reference's diffusion module object and its diffuser:
`DiffusionTTS` (timestep_independent / forward) and `get_diffuser(...).sample_loop(...)`
(/root/reference/tortoise_tts/models/diffusion.py:1389-1590, 188-810, 1110-1267), as `TTS.inference` calls them
(tortoise_tts/inference.py:183, 402-412), backed by libttk (HIP, gfx950).  No torch fallback exists.

Host logic kept here (it is host logic in the reference too): the float64 beta/alpha tables of the spaced schedule,
the timestep map, the per-step scalar coefficients, the ramped conditioning-free weight -- all numpy float64, converted to
f32 scalars exactly where the reference converts (`_extract_into_tensor(...).float()`, diffusion.py:1264).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib
from .weights import DiffusionConfig, diffusion_shapes

TACOTRON_MEL_MAX = 2.3143386840820312
TACOTRON_MEL_MIN = -11.512925148010254


def denormalize_tacotron_mel(norm_mel):
	"""/root/reference/tortoise_tts/models/arch_utils.py:532-537 (a17)."""
	return ((norm_mel + 1) / 2) * (TACOTRON_MEL_MAX - TACOTRON_MEL_MIN) + TACOTRON_MEL_MIN


# ------------------------------------------------------------------------------------------------ derived tables
def _relative_position_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 64) -> torch.Tensor:
	"""T5 bucket of rel = k - q, bidirectional (xtransformers.py:157-177 with causal=False, arch_utils.py:174)."""
	n = -rel
	nb = num_buckets // 2
	ret = (n < 0).long() * nb
	n = torch.abs(n)
	max_exact = nb // 2
	is_small = n < max_exact
	val = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (nb - max_exact)).long()
	val = torch.min(val, torch.full_like(val, nb - 1))
	return ret + torch.where(is_small, n, val)


def relbias_table(emb: torch.Tensor, head_dim: int) -> torch.Tensor:
	"""[H, 129] additive attention bias for clamp(k - q, -64, 64): the bucket is constant beyond +-64 (it saturates at
	max_distance), so this table reproduces `RelativePositionBias.forward` (xtransformers.py:179-188) for any length."""
	rel = torch.arange(-64, 65)
	assert int(_relative_position_bucket(torch.tensor([-64]))) == int(_relative_position_bucket(torch.tensor([-100000])))
	assert int(_relative_position_bucket(torch.tensor([64]))) == int(_relative_position_bucket(torch.tensor([100000])))
	return (emb.float().cpu()[_relative_position_bucket(rel)] * (head_dim ** 0.5)).t().contiguous()


def time_freqs(dim: int, max_period: int = 10000) -> torch.Tensor:
	"""diffusion.py:1287-1290."""
	half = dim // 2
	return torch.exp(-math.log(max_period) * torch.arange(start=0, end=half, dtype=torch.float32) / half)


def nearest_index(M: int, T: int) -> torch.Tensor:
	"""Source row of each output frame for F.interpolate(..., size=T, mode='nearest') (diffusion.py:1507):
	floor(dst * scale) with the f32 scale = M / T, clamped (ATen nearest_neighbor_compute_source_index)."""
	scale = np.float32(M) / np.float32(T)
	idx = np.floor(np.arange(T, dtype=np.float32) * scale).astype(np.int64)
	return torch.from_numpy(np.minimum(idx, M - 1).astype(np.int32))


def pack_state_dict(sd: Dict[str, torch.Tensor], cfg: DiffusionConfig) -> Dict[str, torch.Tensor]:
	"""Reference-layout tensors + the derived host tables ttk_diff_create expects (include/ttk.h)."""
	out = {k: sd[k] for k in diffusion_shapes(cfg)}
	out["__time_freqs"] = time_freqs(cfg.model_channels)
	res_prefixes = [f"conditioning_timestep_integrator.{i}.resblk." for i in range(3)]
	res_prefixes += [f"layers.{i}.resblk." for i in range(cfg.num_layers)]
	res_prefixes += [f"layers.{i}." for i in range(cfg.num_layers, cfg.num_layers + 3)]
	out["__emb_cat.weight"] = torch.cat([sd[p + "emb_layers.1.weight"].float().cpu() for p in res_prefixes], dim=0)
	out["__emb_cat.bias"] = torch.cat([sd[p + "emb_layers.1.bias"].float().cpu() for p in res_prefixes], dim=0)
	for k in list(sd.keys()):
		if k.endswith("relative_pos_embeddings.relative_attention_bias.weight") and k in out:
			prefix = k[: -len("relative_pos_embeddings.relative_attention_bias.weight")]
			out[prefix + "__relbias"] = relbias_table(sd[k], cfg.head_dim)
	return out


# ------------------------------------------------------------------------------------------------ the network
class DiffusionTTS:
	def __init__(self, state_dict: Dict[str, torch.Tensor], cfg: DiffusionConfig = DiffusionConfig(), dtype: str = "bf16",
				 device: str = "cuda:0"):
		self.cfg = cfg
		self.device = torch.device(device)
		if self.device.type != "cuda":
			raise _lib.TTKError("tortoise_tts_amd runs on an MI355X only (device must be cuda:N)")
		self.lib = _lib.load()
		self.dtype = _lib.DTYPES[dtype]
		self.in_channels, self.out_channels, self.model_channels = cfg.in_channels, cfg.out_channels, cfg.model_channels
		missing = [n for n in diffusion_shapes(cfg) if n not in state_dict]
		if missing:
			raise _lib.TTKError(f"state_dict lacks {len(missing)} hot-path tensors, e.g. {missing[:3]}")
		packed = pack_state_dict(state_dict, cfg)
		names = list(packed.keys())
		views, keep = _lib.weight_views(packed, names)
		c = _lib.DiffConfigC(cfg.model_channels, cfg.num_layers, cfg.in_channels, cfg.in_latent_channels, cfg.out_channels,
							 cfg.num_heads, self.dtype)
		self._h = _lib.C.c_void_p()
		with torch.cuda.device(self.device):
			_lib.check(self.lib.ttk_diff_create(_lib.C.byref(self._h), _lib.C.byref(c), views, len(names)), "ttk_diff_create")
		del keep
		self._idx_cache: Dict[tuple, torch.Tensor] = {}

	def __del__(self):
		h = getattr(self, "_h", None)
		if h:
			self.lib.ttk_diff_destroy(h)
			self._h = None

	def parameters(self):
		"""`next(model.parameters()).device` is queried by the sampler (diffusion.py:788)."""
		yield torch.empty(0, device=self.device)

	def to(self, *a, **k):
		return self

	def eval(self):
		return self

	def timestep_independent(self, aligned_conditioning, conditioning_latent, expected_seq_len, return_code_pred=False):
		"""diffusion.py:1487-1510, latent conditioning: [b, M, C_latent] f32, [b, 2C] -> [b, C, T] f32."""
		if return_code_pred:
			raise NotImplementedError("return_code_pred is a training-only branch")
		if aligned_conditioning.dtype != torch.float32:
			raise NotImplementedError("token conditioning (code_embedding/code_converter) is off the inference path")
		lat = aligned_conditioning.to(self.device, torch.float32).contiguous()
		cond = conditioning_latent.to(self.device, torch.float32).contiguous()
		b, M, _ = lat.shape
		if cond.shape[0] != b:
			cond = cond.expand(b, -1).contiguous()
		T = int(expected_seq_len)
		key = (M, T)
		if key not in self._idx_cache:
			self._idx_cache[key] = nearest_index(M, T).to(self.device)
		E = torch.empty((b, self.cfg.model_channels, T), device=self.device, dtype=torch.float32)
		_lib.check(self.lib.ttk_diff_precompute(self._h, lat.data_ptr(), cond.data_ptr(), self._idx_cache[key].data_ptr(), b, M, T,
												E.data_ptr(), _lib.stream_ptr()), "ttk_diff_precompute")
		return E

	def forward(self, x, timesteps, aligned_conditioning=None, conditioning_latent=None, precomputed_aligned_embeddings=None,
				conditioning_free=False, return_code_pred=False):
		"""diffusion.py:1517-1574 with precomputed embeddings: [b, 100, T], [b] -> [b, 200, T] f32."""
		if return_code_pred:
			raise NotImplementedError("return_code_pred is a training-only branch")
		x = x.to(self.device, torch.float32).contiguous()
		b, _, T = x.shape
		t = timesteps.to(self.device, torch.int64).contiguous()
		E = None
		if not conditioning_free:
			if precomputed_aligned_embeddings is None:
				if aligned_conditioning is None or conditioning_latent is None:
					raise ValueError("need precomputed_aligned_embeddings or (aligned_conditioning, conditioning_latent)")
				precomputed_aligned_embeddings = self.timestep_independent(aligned_conditioning, conditioning_latent, T)
			E = precomputed_aligned_embeddings.to(self.device, torch.float32).contiguous()
			if E.shape[0] != b:
				E = E.expand(b, -1, -1).contiguous()
		out = torch.empty((b, self.cfg.out_channels, T), device=self.device, dtype=torch.float32)
		_lib.check(self.lib.ttk_diff_forward(self._h, x.data_ptr(), t.data_ptr(), _lib.ptr(E), b, T, out.data_ptr(),
											 _lib.stream_ptr()), "ttk_diff_forward")
		return out

	__call__ = forward


# ------------------------------------------------------------------------------------------------ schedule + samplers
def get_named_beta_schedule(schedule_name: str, num_diffusion_timesteps: int) -> np.ndarray:
	"""diffusion.py:107-131 ('linear' is the only schedule get_diffuser uses)."""
	if schedule_name != "linear":
		raise NotImplementedError(f"unknown beta schedule: {schedule_name}")
	scale = 1000 / num_diffusion_timesteps
	return np.linspace(scale * 0.0001, scale * 0.02, num_diffusion_timesteps, dtype=np.float64)


def space_timesteps(num_timesteps: int, section_counts) -> set:
	"""diffusion.py:1169-1222."""
	if isinstance(section_counts, str):
		if section_counts.startswith("ddim"):
			desired = int(section_counts[len("ddim"):])
			for i in range(1, num_timesteps):
				if len(range(0, num_timesteps, i)) == desired:
					return set(range(0, num_timesteps, i))
			raise ValueError(f"cannot create exactly {num_timesteps} steps with an integer stride")
		section_counts = [int(x) for x in section_counts.split(",")]
	size_per = num_timesteps // len(section_counts)
	extra = num_timesteps % len(section_counts)
	start_idx = 0
	all_steps: List[int] = []
	for i, section_count in enumerate(section_counts):
		size = size_per + (1 if i < extra else 0)
		if size < section_count:
			raise ValueError(f"cannot divide section of {size} steps into {section_count}")
		frac_stride = 1 if section_count <= 1 else (size - 1) / (section_count - 1)
		cur_idx = 0.0
		for _ in range(section_count):
			all_steps.append(start_idx + round(cur_idx))
			cur_idx += frac_stride
		start_idx += size
	return set(all_steps)


class SpacedDiffusion:
	"""diffusion.py:1110-1166 over GaussianDiffusion.__init__ :205-262, epsilon model / learned-range variance."""

	def __init__(self, use_timesteps, betas, conditioning_free=False, conditioning_free_k=1, ramp_conditioning_free=True):
		self.use_timesteps = set(use_timesteps)
		self.original_num_steps = len(betas)
		base = np.cumprod(1.0 - np.array(betas, dtype=np.float64), axis=0)
		last = 1.0
		new_betas, self.timestep_map = [], []
		for i, ac in enumerate(base):
			if i in self.use_timesteps:
				new_betas.append(1 - ac / last)
				last = ac
				self.timestep_map.append(i)
		betas = np.array(new_betas, dtype=np.float64)
		self.betas = betas
		self.num_timesteps = int(betas.shape[0])
		self.conditioning_free = conditioning_free
		self.conditioning_free_k = conditioning_free_k
		self.ramp_conditioning_free = ramp_conditioning_free
		alphas = 1.0 - betas
		self.alphas_cumprod = np.cumprod(alphas, axis=0)
		self.alphas_cumprod_prev = np.append(1.0, self.alphas_cumprod[:-1])
		self.sqrt_recip_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod)
		self.sqrt_recipm1_alphas_cumprod = np.sqrt(1.0 / self.alphas_cumprod - 1)
		self.posterior_variance = betas * (1.0 - self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
		self.posterior_log_variance_clipped = np.log(np.append(self.posterior_variance[1], self.posterior_variance[1:]))
		self.posterior_mean_coef1 = betas * np.sqrt(self.alphas_cumprod_prev) / (1.0 - self.alphas_cumprod)
		self.posterior_mean_coef2 = (1.0 - self.alphas_cumprod_prev) * np.sqrt(alphas) / (1.0 - self.alphas_cumprod)

	def step_coefs(self, i: int, sampler: str) -> _lib.StepC:
		"""Scalars of step i.  float64 table -> f32 (`.float()`, :1264); the DDIM square roots are taken in f32 on the
		f32 alpha_bar_prev as `torch.sqrt` does on the extracted tensor (:677-688)."""
		f = np.float32
		ab_prev = f(self.alphas_cumprod_prev[i])
		if self.conditioning_free:
			cfk = self.conditioning_free_k * (1 - i / self.num_timesteps) if self.ramp_conditioning_free else self.conditioning_free_k
		else:
			cfk = -1.0
		s = _lib.StepC()
		s.t = int(self.timestep_map[i])
		s.sqrt_recip_ac = f(self.sqrt_recip_alphas_cumprod[i])
		s.sqrt_recipm1_ac = f(self.sqrt_recipm1_alphas_cumprod[i])
		s.sqrt_ac_prev = np.sqrt(ab_prev, dtype=f)
		s.sqrt_1m_ac_prev = np.sqrt(f(1) - ab_prev, dtype=f)
		s.coef1 = f(self.posterior_mean_coef1[i])
		s.coef2 = f(self.posterior_mean_coef2[i])
		s.min_log = f(self.posterior_log_variance_clipped[i])
		s.max_log = f(np.log(self.betas)[i])
		s.cfk = float(cfk)
		s.sampler = 0 if sampler == "ddim" else 1
		s.nonzero = int(i != 0)
		return s

	def sample_loop(self, model: DiffusionTTS, shape, noise=None, clip_denoised=True, denoised_fn=None, cond_fn=None,
					model_kwargs=None, device=None, progress=False, eta=0.0, sampler="ddim", consume_rng=True):
		"""diffusion.py:500-508 -> ddim_sample_loop :734-810 / p_sample_loop :556-644.  Returns f32 [b, 100, T]."""
		sampler = sampler.lower()
		if sampler not in ("ddim", "p"):
			raise RuntimeError(f"Sampler not implemented: {sampler}")
		if not clip_denoised or denoised_fn is not None or cond_fn is not None or eta != 0.0:
			raise NotImplementedError("only clip_denoised=True, eta=0, no denoised_fn/cond_fn (what inference.py:405-412 uses)")
		if not isinstance(model, DiffusionTTS):
			raise _lib.TTKError("sample_loop needs the libttk-backed DiffusionTTS (no fallback path)")
		E = (model_kwargs or {}).get("precomputed_aligned_embeddings")
		if E is None:
			raise ValueError("model_kwargs['precomputed_aligned_embeddings'] is required")
		dev = model.device
		b, C, T = shape
		if self.conditioning_free and self.ramp_conditioning_free:
			assert b == 1, "ramped conditioning-free guidance is batch-1 in the reference (diffusion.py:392)"
		x = (noise if noise is not None else torch.randn(*shape, device=dev)).to(dev, torch.float32).contiguous().clone()
		E = E.to(dev, torch.float32).contiguous()
		n = self.num_timesteps
		with torch.cuda.device(dev):
			if sampler == "ddim":
				steps = (_lib.StepC * n)(*[self.step_coefs(i, "ddim") for i in range(n)])
				_lib.check(model.lib.ttk_diff_sample_ddim(model._h, x.data_ptr(), E.data_ptr(), b, T, steps, n, _lib.stream_ptr()),
						   "ttk_diff_sample_ddim")
				# ddim_sample draws (and ignores) one randn_like(x) per step (:685); keep the generator stream aligned
				# (consume_rng=False: the caller already made these draws, see TTSHotPath.inference_lines)
				if consume_rng:
					for _ in range(n):
						torch.randn_like(x)
			else:
				# p_sample draws one randn_like(x) per step (:545), in loop order: drawn here, consumed by the whole-loop entry
				nz = torch.stack([torch.randn_like(x) for _ in range(n)])
				steps = (_lib.StepC * n)(*[self.step_coefs(i, "p") for i in range(n)])
				_lib.check(model.lib.ttk_diff_sample_p(model._h, x.data_ptr(), E.data_ptr(), b, T, steps, n, nz.data_ptr(), _lib.stream_ptr()),
						   "ttk_diff_sample_p")
		return x


def _sample_loop_lines(self, model: DiffusionTTS, noises, embeddings):
	"""DDIM loops of several utterances of different length as ONE batch (include/ttk.h: ttk_diff_sample_ddim_lines; no reference counterpart --
	the reference diffuses a text's lines one after the other, inference.py:237-422).  noises: list of [1, 100, T_i] start noises, embeddings:
	list of [1, C, T_i] precomputed aligned embeddings.  Returns a list of [1, 100, T_i] f32, element i bit for bit what
	`sample_loop(model, (1, 100, T_i), noise=noises[i], ...)` returns.  Draws nothing from the torch generator: the caller makes the reference's
	per-line draws (start noise, DDIM's ignored per-step randn_like) in line order, as TTSHotPath.inference_lines does."""
	if not self.conditioning_free:
		raise NotImplementedError("line batches run the conditioned and the conditioning-free evaluation as one [cond | uncond] batch")
	if not isinstance(model, DiffusionTTS):
		raise _lib.TTKError("sample_loop_lines needs the libttk-backed DiffusionTTS (no fallback path)")
	dev = model.device
	b = len(noises)
	if b == 0 or len(embeddings) != b:
		raise ValueError("one start noise and one embedding per line")
	Ts = [int(n.shape[-1]) for n in noises]
	if any(e.shape[-1] != t or n.shape[0] != 1 or e.shape[0] != 1 for n, e, t in zip(noises, embeddings, Ts)):
		raise ValueError("noises[i] [1, 100, T_i] and embeddings[i] [1, C, T_i] must agree in T_i")
	Tp = (max(Ts) + 63) // 64 * 64
	C = embeddings[0].shape[1]
	with torch.cuda.device(dev):
		x = torch.zeros((b, noises[0].shape[1], Tp), device=dev, dtype=torch.float32)
		E = torch.zeros((b, C, Tp), device=dev, dtype=torch.float32)
		for i, (n, e, t) in enumerate(zip(noises, embeddings, Ts)):
			x[i, :, :t] = n[0].to(dev, torch.float32)
			E[i, :, :t] = e[0].to(dev, torch.float32)
		n_steps = self.num_timesteps
		steps = (_lib.StepC * n_steps)(*[self.step_coefs(i, "ddim") for i in range(n_steps)])
		tlen = (_lib.C.c_int * b)(*Ts)
		_lib.check(model.lib.ttk_diff_sample_ddim_lines(model._h, x.data_ptr(), E.data_ptr(), b, Tp, tlen, steps, n_steps, _lib.stream_ptr()),
				   "ttk_diff_sample_ddim_lines")
		return [x[i:i + 1, :, :t].contiguous() for i, t in enumerate(Ts)]


SpacedDiffusion.sample_loop_lines = _sample_loop_lines


def get_diffuser(steps=80, cond_free=True, cond_free_k=2, trained_diffusion_steps=4000) -> SpacedDiffusion:
	"""diffusion.py:1576-1590."""
	return SpacedDiffusion(use_timesteps=space_timesteps(trained_diffusion_steps, [steps]),
						   betas=get_named_beta_schedule("linear", trained_diffusion_steps),
						   conditioning_free=cond_free, conditioning_free_k=cond_free_k)
