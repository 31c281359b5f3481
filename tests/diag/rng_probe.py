"""Diagnostic: where does ttk_exponential_like_torch differ from torch.exponential_?"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tortoise_tts_amd import _lib
lib = _lib.load()
dev = "cuda:0"
props = torch.cuda.get_device_properties(0)
print("SMs", props.multi_processor_count, "threads/SM", props.max_threads_per_multi_processor)
gen = torch.cuda.default_generators[0]
for shape in [(1, 5), (1, 1024), (16, 8194)]:
	torch.cuda.manual_seed(0)
	off = gen.get_offset()
	numel = shape[0] * shape[1]
	grid = min(props.multi_processor_count * (props.max_threads_per_multi_processor // 256), (numel + 255) // 256)
	threads = 256 * grid
	want = torch.empty(shape, device=dev).exponential_(1).flatten()
	step = gen.get_offset() - off
	got = torch.empty(numel, device=dev)
	_lib.check(lib.ttk_exponential_like_torch(got.data_ptr(), numel, 0, off, threads, step, 0, _lib.stream_ptr()), "x")
	eq = (got.view(torch.int32) == want.view(torch.int32))
	ulp = (got.view(torch.int32) - want.view(torch.int32)).abs()
	print(shape, "threads", threads, "step", step, "equal", int(eq.sum()), "/", numel, "max ulp", int(ulp.max()), "within 2 ulp", int((ulp <= 2).sum()))
	print(" want", want[:6].tolist()); print(" got ", got[:6].tolist())
	# table of (ii, idx) for it = 0 with a wide thread count: is want[li] anywhere?
	N = 1 << 16
	tab = torch.empty(4 * N, device=dev)
	_lib.check(lib.ttk_exponential_like_torch(tab.data_ptr(), 4 * N, 0, off, N, 4, 0, _lib.stream_ptr()), "x")
	for li in (0, 1, 2, 3, min(numel - 1, 300), numel - 1):
		d = (tab.view(torch.int32) - want[li].view(torch.int32)).abs()
		j = int(d.argmin())
		print(f"  li {li}: closest table entry ii={j // N} idx={j % N} ulp {int(d[j])}")
	# uniform of torch for the same state: u = exp(-x) approx
	u = torch.empty(shape, device=dev)
	gen.set_offset(off)
	u.uniform_(0, 1)
	print(" torch.uniform_ (1-u relation?)", u.flatten()[:4].tolist(), " exp(-want)", torch.exp(-want[:4]).tolist(), " exp(-got)", torch.exp(-got[:4]).tolist())
