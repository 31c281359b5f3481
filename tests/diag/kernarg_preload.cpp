// Diagnostic (not a test, not product code): does SGPR preloading of leading kernel arguments (-mllvm -amdgpu-kernarg-preload-count=16) shorten a
// chain of dependent small kernels?  Two builds of the same chain: args in one by-value struct (what libttk's kernels take; never preloaded)
// vs leading scalar args.  Each kernel: 256 workgroups read 64 KB written by the previous kernel and write 64 KB (a decode-step stand-in).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tests/diag/kernarg_preload.cpp -o tests/diag/kernarg_preload.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct P { const float* in; float* out; const float* w; int n; int pad[50]; };
__global__ __launch_bounds__(256) void k_struct(P p) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	float acc = p.w[i];
	for (int j = 0; j < 4; ++j) acc += p.in[(i + j * 4096) & (p.n - 1)];
	p.out[i & (p.n - 1)] = acc * (1.0f + p.pad[7]);
}
__global__ __launch_bounds__(256) void k_scalar(const float* in, float* out, const float* w, int n, P rest) {
	const int i = blockIdx.x * 256 + threadIdx.x;
	float acc = w[i];
	for (int j = 0; j < 4; ++j) acc += in[(i + j * 4096) & (n - 1)];
	out[i & (n - 1)] = acc * (1.0f + rest.pad[7]);
}
int main() {
	const int n = 16384, chain = 600;
	float *a, *b, *w;
	CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&w, 65536 * 4));
	CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(w, 0, 65536 * 4));
	hipStream_t s; CK(hipStreamCreate(&s));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	for (int variant = 0; variant < 2; ++variant) {
		hipGraph_t g; hipGraphExec_t ge;
		CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
		for (int i = 0; i < chain; ++i) {
			P p = {}; p.in = (i & 1) ? b : a; p.out = (i & 1) ? a : b; p.w = w; p.n = n;
			if (variant == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, s, p);
			else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(256), 0, s, p.in, p.out, p.w, n, p);
		}
		CK(hipStreamEndCapture(s, &g));
		CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		float best = 1e9f;
		for (int rep = 0; rep < 8; ++rep) {
			CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			if (rep >= 2 && ms < best) best = ms;
		}
		printf("%s: %.3f us per dependent launch (graph of %d)\n", variant == 0 ? "args in a by-value struct" : "leading scalar args (preloaded)", best * 1e3 / chain, chain);
		CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
	}
	return 0;
}
