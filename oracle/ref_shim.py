"""Import the reference's hot-path modules by path, in THIS container only.

TEST INFRASTRUCTURE.  Only `oracle/make_golden.py` uses this file; nothing on the
product path, in `tests/`, `bench.py` or `smoke()` may import it (`/root/reference`
does not exist on the GPU box).

The reference package cannot be imported as a package here (ordinary
ModuleNotFoundError: torchaudio, h5py, inflect ... are not installed; SURVEY.md
section 8c).  Its hot-path files import cleanly once the third-party names they only
touch at import time are stubbed:

  tortoise_tts/models/diffusion.py      (sampler + DiffusionTTS)
  tortoise_tts/models/arch_utils.py     (AttentionBlock, GroupNorm32)
  tortoise_tts/models/xtransformers.py  (RelativePositionBias)
  tortoise_tts/models/unified_voice.py  (UnifiedVoice, GPT2InferenceModel)
  tortoise_tts/models/stream_generator.py
  tortoise_tts/models/lora.py, bigvgan.py, clvp.py              (imported on demand by make_golden.py)

No reference source is copied: the modules are executed where they lie.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("TTK_REFERENCE", "/root/reference")


def _stub(name, **attrs):
	m = types.ModuleType(name)
	m.__dict__.update(attrs)
	m.__spec__ = importlib.machinery.ModuleSpec(name, None)
	sys.modules[name] = m
	return m


def load():
	"""Returns (diffusion_module, unified_voice_module)."""
	if not os.path.isdir(REF_ROOT):
		raise RuntimeError(f"reference not present at {REF_ROOT}")
	sys.dont_write_bytecode = True
	import transformers
	from transformers import GPT2Config, GPT2Model  # noqa: F401  (resolve lazies before stubbing)
	import transformers.generation.utils as gu

	for name in ("torchaudio", "torchaudio.transforms", "librosa"):
		if name not in sys.modules:
			_stub(name)
	if "rotary_embedding_torch" not in sys.modules:   # models/transformer.py:10 (the non-x-transformers CLVP branch, never instantiated here)
		_stub("rotary_embedding_torch", RotaryEmbedding=None, broadcat=None)
	if "librosa.filters" not in sys.modules:
		_stub("librosa.filters", mel=None)        # models/bigvgan.py:14 imports the name; only its training-side mel_spectrogram() calls it
	if "librosa.util" not in sys.modules:
		_stub("librosa.util", pad_center=None, tiny=None)
	if "transformers.utils.model_parallel_utils" not in sys.modules:
		_stub("transformers.utils.model_parallel_utils", get_device_map=None, assert_device_map=None)
	for removed in ("DisjunctiveConstraint", "BeamSearchScorer", "PhrasalConstraint", "ConstrainedBeamSearchScorer",
					"LogitsWarper"):
		if removed not in sys.modules["transformers"].__dict__:
			try:
				getattr(transformers, removed)
			except Exception:
				sys.modules["transformers"].__dict__[removed] = type(removed, (), {})
	if not hasattr(gu, "SampleOutput"):
		gu.SampleOutput = object

	for pkg, rel in (("tortoise_tts", "tortoise_tts"), ("tortoise_tts.models", "tortoise_tts/models")):
		if pkg not in sys.modules:
			m = types.ModuleType(pkg)
			m.__path__ = [os.path.join(REF_ROOT, rel)]
			sys.modules[pkg] = m
	diffusion = importlib.import_module("tortoise_tts.models.diffusion")
	unified_voice = importlib.import_module("tortoise_tts.models.unified_voice")
	return diffusion, unified_voice


if __name__ == "__main__":
	d, u = load()
	print("ok", d.DiffusionTTS, u.UnifiedVoice)
