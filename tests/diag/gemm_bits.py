"""Diagnostic: the dense GEMM through ttk_gemm_nt under the library TTK_LIB selects, per forced tile (TTK_GEMM_TILE) and shape: digest of C and the largest error
against an f64 matmul relative to sum|a||w| (bf16 operands: f32 accumulation only, < 1e-5).  Two builds agree bit for bit iff their lines are equal."""
import ctypes as C, hashlib, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
	import torch
	from tortoise_tts_amd import _lib
	lib = _lib.load()
	dev = "cuda:0"
	for dt, name in ((1, "bf16"), (4, "f16")):
		for (M, N, K) in [(2176, 1024, 1024), (2000, 1024, 1024), (640, 3072, 1024), (333, 256, 192), (64, 128, 64), (4352, 3072, 1024)]:
			g = torch.Generator().manual_seed(M + N + K)
			A, Wt = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
			A, Wt = (A.bfloat16(), Wt.bfloat16()) if dt == 1 else (A.half(), Wt.half())
			out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float32)
			Ad, Wd = A.to(dev).contiguous(), Wt.to(dev).contiguous()
			_lib.check(lib.ttk_gemm_nt(dt, Ad.data_ptr(), Wd.data_ptr(), M, N, K, C.c_float(0.0), None, out.data_ptr(), _lib.stream_ptr()), "ttk_gemm_nt")
			torch.cuda.synchronize()
			got = out.cpu()
			ref = A.double() @ Wt.double().t()
			mag = A.double().abs() @ Wt.double().abs().t()
			err = ((got.double() - ref).abs() / mag).max().item()
			print(f"tile {os.environ.get('TTK_GEMM_TILE', 'auto'):>4} {name} {M}x{N}x{K}: {hashlib.sha256(got.numpy().tobytes()).hexdigest()[:16]} err {err:.2e}", flush=True)
	sys.exit(0)
for tile in ("", "0", "1", "2", "5", "6", "7", "8"):
	env = dict(os.environ)
	if tile: env["TTK_GEMM_TILE"] = tile
	subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
