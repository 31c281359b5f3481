// Diagnostic (not a test, not product code): times the dense GEMM on the shapes of the benchmark step.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tortoise_tts_amd/csrc tests/diag/gemm_bench.cpp -o /tmp/gemm_bench && /tmp/gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <stdlib.h>
#include "../../tortoise_tts_amd/csrc/gemm.hip"
bool ttk::g_prof_on = false;
void ttk::prof_start(int, double, hipStream_t) {}
void ttk::prof_stop(hipStream_t) {}
void ttk::prof_pair(int, double, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
using namespace ttk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void fill(unsigned short* p, size_t n, unsigned seed) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) { unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 13; h *= 0x5bd1e995; p[i] = (unsigned short)(0x3c00 + (h & 0x3ff) - ((h >> 10) & 1) * 0x8000); }   // ~[-2, 2) bf16 bit patterns
}
// producer stand-in for the kernel that writes a GEMM's A operand right before it (GroupNorm-apply): rows of A re-written by workgroups whose
// XCD (blockIdx % 8) either owns a contiguous eighth of the rows (aligned = 1: the eighth the m-major GEMM order reads on that XCD) or
// is interleaved strip by strip (aligned = 0, what a plain blockIdx -> strip mapping gives)
__global__ void produce_rows(unsigned short* A, int M, int K, int aligned, unsigned salt) {
	const int strips = M / 2;                           // 2 rows per workgroup, like k_gn_apply at C = 1024
	int strip = blockIdx.x;
	if (aligned) { const int per = strips / 8; strip = (blockIdx.x % 8) * per + blockIdx.x / 8; if (blockIdx.x / 8 >= per) return; }
	for (int i = threadIdx.x; i < 2 * K / 8; i += blockDim.x) {
		uint4 v = make_uint4(0x3c003c00u + salt, 0x3c003c00u, 0xbc003c00u, 0x3c00bc00u);
		*(uint4*)(A + (size_t)strip * 2 * K + (size_t)i * 8) = v;
	}
}
int main(int argc, char** argv) {
	if (argc > 3 && atoi(argv[3]) == 1) {
		// affinity experiment: 1x1 conv shape, ONE A / W set (L2 / MALL resident as in the diffusion loop), A re-written before every GEMM
		const int M = 2176, N = 1024, K = 1024;
		char *A, *W; void* Cb; float* bias;
		hipStream_t s; CK(hipStreamCreate(&s));
		hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
		CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&Cb, (size_t)M * N * 2)); CK(hipMalloc(&bias, N * 4));
		fill<<<(unsigned)(((size_t)N * K + 255) / 256), 256, 0, s>>>((unsigned short*)W, (size_t)N * K, 2);
		CK(hipMemsetAsync(bias, 0, N * 4, s));
		for (int aligned = 0; aligned < 2; ++aligned)
			for (int order = 0; order < 2; ++order) {
				GemmParams g = {};
				g.nseg = 1; g.seg[0] = {A, K, 0, 0}; g.W = W; g.ldw = K; g.M = M; g.N = N; g.K = K; g.rows_per_batch = 1088; g.bias = bias; g.C = Cb; g.ldc = N; g.m_major = order;
				g_force_tile = 101;
				float tot = 0.f;
				const int reps = 200;
				for (int i = 0; i < reps + 10; ++i) {
					produce_rows<<<M / 2, 256, 0, s>>>((unsigned short*)A, M, K, aligned, (unsigned)i & 1);
					if (i >= 10) CK(hipEventRecord(e0, s));
					launch_gemm(DT_BF16, g, s);
					if (i >= 10) { CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; }
				}
				printf("1x1 conv 2176x1024x1024, A just written by %s producer, GEMM tile order %s: %.2f us\n", aligned ? "an XCD-aligned" : "an interleaved", order ? "m-major" : "n-major", tot * 1e3 / reps);
			}
		return 0;
	}
	struct Shape { const char* name; int M, N, K, nseg, T, resid; } shapes[] = {
		{"fixed-cost probe K=128 bf16 out", 2176, 1024, 128, 1, 1088, 0},
		{"fixed-cost probe K=128 f32+res", 2176, 1024, 128, 1, 1088, 1},
		{"tail probe 2048x1024x1024 (256 tiles)", 2048, 1024, 1024, 1, 1024, 0},
		{"tail probe 4096x1024x1024 (512 tiles)", 4096, 1024, 1024, 1, 1024, 0},
		{"1x1 conv  2176x1024x1024", 2176, 1024, 1024, 1, 1088, 0},
		{"proj+res  2176x1024x1024", 2176, 1024, 1024, 1, 1088, 1},
		{"conv3+res 2176x1024x3x1024", 2176, 1024, 1024, 3, 1088, 1},
		{"qkv       2176x3072x1024", 2176, 3072, 1024, 1, 1088, 0},
		{"integ     2176x1024x2x1024", 2176, 1024, 1024, 2, 1088, 0},
		{"ar c_attn 5104x3072x1024", 5104, 3072, 1024, 1, 0, 0},
		{"ar c_fc   5104x4096x1024", 5104, 4096, 1024, 1, 0, 0},
		{"ar proj2  5104x1024x4096", 5104, 1024, 4096, 1, 0, 1},
	};
	hipStream_t s; CK(hipStreamCreate(&s));
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	// cold = 1 (default): every launch reads a different copy of A and W out of pools larger than L2 + MALL, which is what the diffusion
	// network does (320 MB of weights per evaluation, activations just written by another kernel); cold = 0 re-reads one copy (L2-hot).
	const int cold = argc > 1 ? atoi(argv[1]) : 1;
	const int only_tile = argc > 2 ? atoi(argv[2]) : -1;
	for (auto& sh : shapes) {
		const int Npad = (sh.N + 127) / 128 * 128;
		const size_t an = (size_t)sh.M * sh.K * (sh.nseg == 2 ? 2 : 1), wn = (size_t)sh.nseg * Npad * sh.K;
		int nset = cold ? (int)((size_t)640e6 / ((an + wn) * 2)) + 1 : 1;
		nset = nset < 1 ? 1 : (nset > 256 ? 256 : nset);
		char *A, *W; void* Cb; float *bias, *Cf;
		CK(hipMalloc(&A, an * 2 * nset)); CK(hipMalloc(&W, wn * 2 * nset)); CK(hipMalloc(&Cb, (size_t)sh.M * sh.N * 2)); CK(hipMalloc(&Cf, (size_t)sh.M * sh.N * 4)); CK(hipMalloc(&bias, sh.N * 4));
		fill<<<(unsigned)((an * nset + 255) / 256), 256, 0, s>>>((unsigned short*)A, an * nset, 1); fill<<<(unsigned)((wn * nset + 255) / 256), 256, 0, s>>>((unsigned short*)W, wn * nset, 2);
		CK(hipMemsetAsync(bias, 0, sh.N * 4, s)); CK(hipMemsetAsync(Cf, 0, (size_t)sh.M * sh.N * 4, s));
		auto params = [&](int set) {
			GemmParams g = {};
			char* a = A + (size_t)set * an * 2;
			g.nseg = sh.nseg;
			for (int j = 0; j < sh.nseg; ++j) {
				if (sh.nseg == 3) g.seg[j] = {a, sh.K, j - 1, (int64_t)j * Npad * sh.K};
				else if (sh.nseg == 2) g.seg[j] = {a + (size_t)j * sh.M * sh.K * 2, sh.K, 0, (int64_t)j * sh.K};
				else g.seg[j] = {a, sh.K, 0, 0};
			}
			g.W = W + (size_t)set * wn * 2; g.ldw = sh.nseg == 2 ? 2 * sh.K : sh.K; g.M = sh.M; g.N = sh.N; g.K = sh.K; g.rows_per_batch = sh.T; g.bias = bias;
			if (sh.resid) { g.residual = Cf; g.ldr = sh.N; g.C = Cf; g.ldc = sh.N; g.out_f32 = 1; } else { g.C = Cb; g.ldc = sh.N; }
			return g;
		};
		for (int tile = 0; tile < 8; ++tile) {
			if (only_tile >= 0 && tile != only_tile) continue;
			g_force_tile = 100 + tile;
			for (int i = 0; i < 5; ++i) launch_gemm(DT_BF16, params(i % nset), s);
			CK(hipStreamSynchronize(s));
			const int reps = 100;
			CK(hipEventRecord(e0, s));
			for (int i = 0; i < reps; ++i) launch_gemm(DT_BF16, params((i + 5) % nset), s);
			CK(hipEventRecord(e1, s));
			CK(hipEventSynchronize(e1));
			float ms; CK(hipEventElapsedTime(&ms, e0, e1));
			const double us = ms * 1e3 / reps, tf = 2.0 * sh.M * sh.N * (double)sh.K * sh.nseg / (us * 1e-6) / 1e12;
			printf("%-38s %s tile %s  %8.2f us  %7.1f TF/s\n", sh.name, cold ? "cold" : "hot ", tile == 0 ? "128x128 8w" : tile == 1 ? "128x64  4w" : tile == 3 ? "128x128 8w 4st" : tile == 4 ? "128x64 4w 5st" : tile == 5 ? "128x64 8w(4x2)" : tile == 6 ? "128x64 8w(2x4)" : tile == 7 ? "256x64 8w" : "64x64   4w", us, tf);
		}
		CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(Cb)); CK(hipFree(Cf)); CK(hipFree(bias));
	}
	return 0;
}
