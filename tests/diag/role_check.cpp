// Diagnostic (not a test, not product code): each GemmRole instantiation against the generic k_gemm on random operands, bit for bit -- C and the
// fused GroupNorm statistics.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tortoise_tts_amd/csrc tests/diag/role_check.cpp -o tests/diag/role_check.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../tortoise_tts_amd/csrc/gemm.hip"
bool ttk::g_prof_on = false;
void ttk::prof_start(int, double, hipStream_t) {}
void ttk::prof_stop(hipStream_t) {}
void ttk::prof_pair(int, double, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
using namespace ttk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float frand() { return (float)(rand() & 0xFFFFFF) / 16777216.f * 2.f - 1.f; }

int main(int argc, char** argv) {
	const int T = argc > 1 ? atoi(argv[1]) : 1088, nb = argc > 2 ? atoi(argv[2]) : 2, M = nb * T, C = 1024;
	const bool f8 = argc > 3 && !strcmp(argv[3], "f8");      // fp8-e4m3 operands (one byte per element, tensor scale 2^-3 in the epilogue) instead of bf16
	const int dt = f8 ? DT_FP8 : DT_BF16;
	srand(1);
	std::vector<unsigned short> hA((size_t)M * C), hW((size_t)3 * 3 * C * C);
	for (auto& v : hA) v = f2bf(frand());
	for (auto& v : hW) v = f2bf(frand() * 0.05f);
	if (f8) {      // the same buffers read as bytes: random e4m3 codes below the NaN encodings (exponent field < 15), either sign
		unsigned char* a8 = (unsigned char*)hA.data(); unsigned char* w8 = (unsigned char*)hW.data();
		for (size_t i = 0; i < hA.size() * 2; ++i) a8[i] = (unsigned char)(((rand() & 1) << 7) | (rand() % 0x60));
		for (size_t i = 0; i < hW.size() * 2; ++i) w8[i] = (unsigned char)(((rand() & 1) << 7) | (rand() % 0x50));
	}
	std::vector<float> hb(3 * C), hr((size_t)M * C);
	for (auto& v : hb) v = frand();
	for (auto& v : hr) v = frand();
	void *A, *Wt; float *bias, *res0, *Cout[2], *part[2];
	const int nch = T / 64;
	CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&Wt, hW.size() * 2)); CK(hipMalloc(&bias, hb.size() * 4)); CK(hipMalloc(&res0, hr.size() * 4));
	for (int i = 0; i < 2; ++i) { CK(hipMalloc(&Cout[i], (size_t)M * 3 * C * 4)); CK(hipMalloc(&part[i], (size_t)nb * 32 * nch * 3 * 4)); }
	CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(Wt, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
	CK(hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(res0, hr.data(), hr.size() * 4, hipMemcpyHostToDevice));
	hipStream_t s; CK(hipStreamCreate(&s));
	const char* names[5] = {"", "in 1x1", "conv3 + res", "qkv", "proj + res"};
	for (int role = 1; role <= 4; ++role) {
		size_t cbytes = 0;
		bool skipped = false;
		for (int pass = 0; pass < 2; ++pass) {
			g_gemm_roles = pass == 0 ? 0 : 0x1E;
			GemmParams g = {};
			g.W = Wt; g.ldw = C; g.M = M; g.K = C; g.bias = bias; g.C = Cout[pass];
			if (f8) g.out_scale = 0.125f;
			CK(hipMemset(Cout[pass], 0xFF, (size_t)M * 3 * C * 4)); CK(hipMemset(part[pass], 0xFF, (size_t)nb * 32 * nch * 3 * 4));
			if (role == GR_QKV) { g.nseg = 1; g.seg[0] = {A, C, 0, 0}; g.N = 3 * C; g.ldc = 3 * C; g.out_f32 = 0; cbytes = (size_t)M * 3 * C * 2; }
			else {
				if (T % 64) { printf("role %d: T %% 64 != 0 -- the product never fuses the statistics there (gemm_fuses_gn_stats), skipped\n", role); skipped = true; break; }
				g.N = C; g.ldc = C; g.out_f32 = 1; g.gn_part = part[pass]; g.gn_T = T; cbytes = (size_t)M * C * 4;
				if (role == GR_CONV3_RES) { g.nseg = 3; g.rows_per_batch = T; for (int j = 0; j < 3; ++j) g.seg[j] = {A, C, j - 1, (int64_t)j * C * C}; }
				else { g.nseg = 1; g.seg[0] = {A, C, 0, 0}; }
				if (role != GR_IN1x1) {      // the residual aliases C
					CK(hipMemcpy(Cout[pass], res0, (size_t)M * C * 4, hipMemcpyDeviceToDevice));
					g.residual = Cout[pass]; g.ldr = C;
				}
			}
			launch_gemm(dt, g, s);
			CK(hipStreamSynchronize(s));
		}
		if (skipped) continue;
		std::vector<unsigned char> c0(cbytes), c1(cbytes);
		CK(hipMemcpy(c0.data(), Cout[0], cbytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), Cout[1], cbytes, hipMemcpyDeviceToHost));
		size_t bad = 0, first = 0;
		for (size_t i = 0; i < cbytes; ++i) if (c0[i] != c1[i]) { if (!bad) first = i; ++bad; }
		std::vector<float> p0((size_t)nb * 32 * nch * 3), p1(p0.size());
		CK(hipMemcpy(p0.data(), part[0], p0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(p1.data(), part[1], p1.size() * 4, hipMemcpyDeviceToHost));
		size_t pbad = 0, pfirst = 0;
		if (role != GR_QKV) for (size_t i = 0; i < p0.size(); ++i) if (memcmp(&p0[i], &p1[i], 4)) { if (!pbad) pfirst = i; ++pbad; }
		printf("role %d (%s): C bytes differing %zu of %zu (first at %zu), stats words differing %zu of %zu (first %zu: %g vs %g)\n", role, names[role], bad, cbytes, first, pbad,
			   p0.size(), pfirst, pbad ? p0[pfirst] : 0.f, pbad ? p1[pfirst] : 0.f);
	}
	return 0;
}
