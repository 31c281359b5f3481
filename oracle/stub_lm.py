"""TEST INFRASTRUCTURE: a model-free "language model" for pinning the sampling LOOP (not the network).

The reference's `generate()` (tortoise_tts/models/stream_generator.py:213-639) is a fork of HuggingFace's `GenerationMixin.generate`
/ `_sample` and cannot run on the transformers version installed here.  What it forks can: this file defines
next-token logits as a pure function of (last token, step) from a seeded table, once as a `transformers.PreTrainedModel` that the
INSTALLED `GenerationMixin.generate` drives (HF:generation/utils.py `_sample`, the code SURVEY.md a5 cites), and once behind the
interface `tortoise_oracle.inference_speech` drives.  Equal ids from the two loops pin the oracle's control flow -- processor and
warper order, softmax + `torch.multinomial` and its generator consumption, pad-after-EOS, stopping, `max_length` -- against real HF
code.  `oracle/make_golden.py hf_sample_loop` stores HF's ids in tests/golden/; `tests/test_oracle_sampling.py` replays them.
"""
import torch

V, R = 8194, 257
START, STOP = 8192, 8193
PREFIX = 12                     # fake prefix rows (cond + text); trunc_index = PREFIX + 1


def make_table(seed: int, stop_bias: float) -> torch.Tensor:
	g = torch.Generator().manual_seed(seed)
	t = torch.randn(R, V, generator=g) * 2.0
	t[:, STOP] += stop_bias
	return t


def stub_logits(table: torch.Tensor, last: torch.Tensor, k: int) -> torch.Tensor:
	return table[(last * 7 + k * 13) % R]


class StubAR:
	"""the interface `tortoise_oracle.inference_speech` drives (prefix_embeddings / prefill / decode)"""

	def __init__(self, cfg, table):
		self.cfg, self.table = cfg, table

	def prefix_embeddings(self, cond, text):
		return torch.zeros(1, PREFIX, 1)

	def prefill(self, prefix, B):
		lg = torch.zeros(B, PREFIX + 1, V)
		lg[:, -1] = stub_logits(self.table, torch.full((B,), START), 0)
		return lg, None, None

	def decode(self, nxt, k, past):
		return stub_logits(self.table, nxt, k), None, None


# name, table seed, stop-token bias, num_return_sequences, max_generate_length, generate kwargs
CASES = [
	("plain", 1, 0.0, 3, 20, dict(temperature=0.8, top_k=0)),
	("stops_early", 2, 7.0, 4, 40, dict(temperature=1.0, top_k=0)),
	("all_warpers", 3, 4.0, 2, 30, dict(temperature=0.7, top_k=50, top_p=0.9, repetition_penalty=2.0)),
	("suppress", 4, 9.0, 2, 16, dict(temperature=0.9, top_k=0, suppress_tokens=[STOP, 5, 17])),
	("top_p_only", 5, 5.0, 5, 25, dict(top_p=0.8, top_k=0)),
	("hf_defaults", 6, 5.0, 3, 25, dict()),                      # nothing passed: GenerationConfig defaults apply (top_k = 50)
	("sixteen_candidates", 7, 3.0, 16, 48, dict(temperature=0.8, top_k=0, top_p=1.0, repetition_penalty=1.0)),   # TTS.inference defaults
]


def hf_generate(table, B, N, kw):
	"""the installed HuggingFace loop on the stub, called the way `UnifiedVoice.inference_speech` calls it (unified_voice.py:662-665)"""
	from transformers import GenerationMixin, GPT2Config, LogitsProcessorList, PreTrainedModel
	from transformers.modeling_outputs import CausalLMOutput
	trunc = PREFIX + 1

	class StubLM(PreTrainedModel, GenerationMixin):
		config_class = GPT2Config

		def __init__(self, config):
			super().__init__(config)
			self.dummy = torch.nn.Parameter(torch.zeros(1))

		def forward(self, input_ids=None, attention_mask=None, **unused):
			lg = torch.zeros(input_ids.shape[0], input_ids.shape[1], V)
			lg[:, -1] = stub_logits(table, input_ids[:, -1], input_ids.shape[1] - trunc)
			return CausalLMOutput(logits=lg)

	model = StubLM(GPT2Config(vocab_size=V, n_layer=1, n_head=1, n_embd=8)).eval()
	inputs = torch.ones(1, trunc, dtype=torch.long)
	inputs[:, -1] = START
	torch.manual_seed(0)                                          # stream_generator.py:223,296
	with torch.inference_mode():
		out = model.generate(inputs, bos_token_id=START, pad_token_id=STOP, eos_token_id=STOP, max_length=trunc + N,
							 logits_processor=LogitsProcessorList(), num_return_sequences=B, do_sample=True, use_cache=False, **kw)
	return out[:, trunc:]
