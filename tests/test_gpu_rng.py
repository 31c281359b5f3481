"""The multinomial noise drawn inside the mel-head launch (include/ttk.h: ttk_ar_set_noise, csrc/ttk_rng.h) against torch's own
`exponential_` on the device: bit for bit, for every launch geometry ATen picks, and through the token loop (same ids, same generator
position afterwards).  GPU only; calls go through the C ABI."""
import pytest
import torch

from tortoise_tts_amd import _lib
from tortoise_tts_amd import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _geometry(numel):
	props = torch.cuda.get_device_properties(0)
	grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
	return 256 * grid, ((numel - 1) // (256 * grid * 4) + 1) * 4


@pytest.mark.parametrize("shape", [(16, 8194), (1, 8194), (7, 300), (64, 8194), (3, 1 << 20), (1, 5)])
@pytest.mark.parametrize("seed", [0, 1234567891011])
def test_exponential_like_torch_is_torchs_draw(shape, seed):
	lib = _lib.load()
	torch.cuda.init()                                            # default_generators is empty until the lazy init has run
	gen = torch.cuda.default_generators[0]
	torch.cuda.manual_seed(seed)
	torch.rand(17, device=DEV)                                   # some non-zero starting offset
	off = gen.get_offset()
	numel = shape[0] * shape[1]
	threads, step = _geometry(numel)
	want = [torch.empty(shape, device=DEV).exponential_(1) for _ in range(3)]
	assert gen.get_offset() - off == 3 * step                   # the offset arithmetic the product relies on
	got = torch.empty(shape, device=DEV)
	for draw in range(3):
		_lib.check(lib.ttk_exponential_like_torch(got.data_ptr(), numel, seed, off, threads, step, draw, _lib.stream_ptr()), "ttk_exponential_like_torch")
		assert torch.equal(got.view(torch.int32), want[draw].view(torch.int32)), (shape, draw)
	assert float(got.min()) > 0.0


@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_token_loop_with_head_drawn_noise_equals_the_torch_drawn_one(dtype, monkeypatch):
	from tortoise_tts_amd.autoregressive import UnifiedVoice
	cfg = W.AR_SMALL
	sd = W.synth_state_dict(W.ar_shapes(cfg), 31)
	sd["mel_head.bias"] = sd["mel_head.bias"].clone()
	sd["mel_head.bias"][cfg.stop_mel_token] += 4.0
	g = torch.Generator().manual_seed(5)
	text, al = torch.randint(1, 255, (1, 9), generator=g).to(DEV), torch.randn(1, cfg.model_dim, generator=g).to(DEV)
	kw = dict(do_sample=True, temperature=0.8, top_k=16, top_p=0.9, repetition_penalty=2.0, max_generate_length=40, num_return_sequences=5)
	res = {}
	for own in ("0", "1"):
		monkeypatch.setenv("TTK_AR_OWN_RNG", own)
		ar = UnifiedVoice(sd, cfg, dtype=dtype, device=DEV, max_batch=8, max_ctx=128)
		with torch.inference_mode():
			ids = [ar.inference_speech(al, text, **kw) for _ in range(2)]      # second call: the captured step
			after = torch.rand(4, device=DEV)
		st = next(iter(ar._states.values()))
		assert st.own_rng == (own == "1")                        # the self-check against torch passed on this device
		res[own] = (ids, after, dict(ar.last_generate))
	assert all(torch.equal(a, b) for a, b in zip(res["0"][0], res["1"][0]))
	assert torch.equal(res["0"][1], res["1"][1])
	assert res["0"][2] == res["1"][2]
	# and shards of it (rows lo.. of the [C, V] draw)
	monkeypatch.setenv("TTK_AR_OWN_RNG", "1")
	ar = UnifiedVoice(sd, cfg, dtype=dtype, device=DEV, max_batch=8, max_ctx=128)
	with torch.inference_mode():
		part = ar.inference_speech(al, text, candidate_shard=(2, 5), **kw)
	full = res["0"][0][0]
	assert torch.equal(part, full[2:5, :part.shape[1]])
