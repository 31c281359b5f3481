"""tortoise_tts_amd: MI355X-native (gfx950) inference hot path of e-c-k-e-r/tortoise-tts -- the UnifiedVoice KV-cached
mel-token decode and the DiffusionTTS DDIM mel decoder -- as hand-written HIP kernels behind a C ABI (include/ttk.h),
exposed under the reference's own method names.  See DESIGN.md and INTEGRATION.md."""
from .weights import ARConfig, CLVPConfig, DiffusionConfig, VocoderConfig  # noqa: F401

__all__ = ["ARConfig", "DiffusionConfig", "UnifiedVoice", "DiffusionTTS", "get_diffuser", "denormalize_tacotron_mel",
		   "load_autoregressive", "load_diffusion", "load_bigvgan", "load_clvp", "load_conditioning_encoder", "load_contextual_embedder", "BigVGAN", "CLVP",
		   "VocoderConfig", "CLVPConfig", "ConditioningEncoder", "ContextualEmbedder", "TorchMelSpectrogram", "TacotronSTFT", "VoiceBpeTokenizer", "TTS"]


def __getattr__(name):   # lazy: importing the package must not need the built library (CPU-side tools, oracle, weights)
	if name == "UnifiedVoice":
		from .autoregressive import UnifiedVoice
		return UnifiedVoice
	if name in ("DiffusionTTS", "get_diffuser", "denormalize_tacotron_mel", "SpacedDiffusion"):
		from . import diffusion
		return getattr(diffusion, name)
	if name == "CLVP":
		from .clvp import CLVP
		return CLVP
	if name in ("ConditioningEncoder", "ContextualEmbedder"):
		from . import conditioning
		return getattr(conditioning, name)
	if name in ("TorchMelSpectrogram", "TacotronSTFT"):
		from . import mel
		return getattr(mel, name)
	if name == "TTS":
		from .tts import TTS
		return TTS
	if name == "VoiceBpeTokenizer":
		from .tokenizer import VoiceBpeTokenizer
		return VoiceBpeTokenizer
	if name == "BigVGAN":
		from .vocoder import BigVGAN
		return BigVGAN
	if name == "mel":
		import importlib
		return importlib.import_module(".mel", __name__)
	if name in ("load_autoregressive", "load_diffusion", "load_bigvgan", "load_clvp", "load_conditioning_encoder", "load_contextual_embedder"):
		from . import checkpoint
		return getattr(checkpoint, name)
	raise AttributeError(name)
