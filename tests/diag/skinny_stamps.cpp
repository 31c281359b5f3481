// Diagnostic (not a test, not product code): in-kernel phase timestamps of the decode skinny GEMM.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTTK_STAMPS -I tortoise_tts_amd/csrc tests/diag/skinny_stamps.cpp -o /tmp/skinny_stamps && /tmp/skinny_stamps
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#include <stdlib.h>
#include "../../tortoise_tts_amd/csrc/skinny.hip"
bool ttk::g_prof_on = false;
void ttk::prof_start(int, double, hipStream_t) {}
void ttk::prof_stop(hipStream_t) {}
void ttk::prof_pair(int, double, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
using namespace ttk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
	const int narrow = argc > 1 ? atoi(argv[1]) : 1;
	const int fold = argc > 2 ? atoi(argv[2]) : 0;      // 1: ln1+qkv / ln2+fc with the LayerNorm folded (plain path over fragment-order rows)
	const int d = 1024, B = 16, H = 16, max_ctx = 512;
	// two alternating "layers" so weights are not L2-resident between launches: 40 distinct weight sets (> 256 MiB total)
	const int NSET = 48;
	struct Set { void *wqkv, *wproj, *wfc, *wproj2; } sets[NSET];
	for (int i = 0; i < NSET; ++i) {
		CK(hipMalloc(&sets[i].wqkv, (size_t)3 * d * d * 2)); CK(hipMalloc(&sets[i].wproj, (size_t)d * d * 2));
		CK(hipMalloc(&sets[i].wfc, (size_t)4 * d * d * 2)); CK(hipMalloc(&sets[i].wproj2, (size_t)4 * d * d * 2));
		CK(hipMemset(sets[i].wqkv, 0, (size_t)3 * d * d * 2)); CK(hipMemset(sets[i].wproj, 0, (size_t)d * d * 2));
		CK(hipMemset(sets[i].wfc, 0, (size_t)4 * d * d * 2)); CK(hipMemset(sets[i].wproj2, 0, (size_t)4 * d * d * 2));
	}
	float *x, *qbuf, *bias, *g, *b, *slab; void *kc, *vc, *ao, *hb; int *dpos, *tickets; unsigned long long* stamps;
	CK(hipMalloc(&x, B * d * 4)); CK(hipMalloc(&qbuf, B * d * 4)); CK(hipMalloc(&bias, 4 * d * 4)); CK(hipMalloc(&g, d * 4)); CK(hipMalloc(&b, d * 4));
	CK(hipMalloc(&kc, (size_t)B * H * max_ctx * 64 * 2)); CK(hipMalloc(&vc, (size_t)B * H * max_ctx * 64 * 2));
	CK(hipMalloc(&ao, B * d * 2)); CK(hipMalloc(&hb, B * 4 * d * 2)); CK(hipMalloc(&dpos, 16)); CK(hipMalloc(&stamps, 512 * 8 * 8));
	CK(hipMalloc(&slab, 64 * 4 * 4 * 256 * 4)); CK(hipMalloc(&tickets, 64 * 4)); CK(hipMemset(tickets, 0, 64 * 4));
	CK(hipMemset(x, 0, B * d * 4)); CK(hipMemset(bias, 0, 4 * d * 4)); CK(hipMemset(g, 0, d * 4)); CK(hipMemset(b, 0, d * 4)); CK(hipMemset(dpos, 0, 16));
	CK(hipMemset(ao, 0, B * d * 2)); CK(hipMemset(hb, 0, B * 4 * d * 2));
	hipStream_t s; CK(hipStreamCreate(&s));
	auto run_layer = [&](int i, unsigned long long* st, int which) {
		SkinnyParams p = {};
		p.Wp = sets[i].wqkv; p.N = 3 * d; p.K = d; p.M = B; p.bias = bias;
		if (fold) { p.g1 = bias; p.a = ao; p.lda = d; p.a_frag = 1; } else { p.ln_count = 1; p.x = x; p.ldx = d; p.g1 = g; p.b1 = b; }
		p.mode = SK_QKV; p.qbuf = qbuf; p.kcache = kc; p.vcache = vc; p.d_pos = dpos; p.max_ctx = max_ctx; p.H = H; p.q_scale = 0.125f;
		p.stamps = which == 0 ? st : nullptr;
		launch_skinny(DT_BF16, p, 8, s);
		p = {}; p.Wp = sets[i].wproj; p.N = d; p.K = d; p.M = B; p.bias = bias; p.a = ao; p.lda = d; p.mode = SK_RESIDUAL; p.out_f32 = x; p.ldc = d; p.narrow = narrow; p.a_frag = 1; p.out_T = fold ? ao : nullptr;
		p.stamps = which == 1 ? st : nullptr;
		launch_skinny(DT_BF16, p, 8, s);
		p = {}; p.Wp = sets[i].wfc; p.N = 4 * d; p.K = d; p.M = B; p.bias = bias;
		if (fold) { p.g1 = bias; p.a = ao; p.lda = d; p.a_frag = 1; } else { p.ln_count = 1; p.x = x; p.ldx = d; p.g1 = g; p.b1 = b; }
		p.mode = SK_ACT_T; p.act = ACT_GELU_NEW; p.out_T = hb; p.out_frag = 1; p.stamps = which == 2 ? st : nullptr;
		launch_skinny(DT_BF16, p, 8, s);
		p = {}; p.Wp = sets[i].wproj2; p.N = d; p.K = 4 * d; p.M = B; p.bias = bias; p.a = hb; p.lda = 4 * d; p.mode = SK_RESIDUAL; p.out_f32 = x; p.ldc = d; p.a_frag = 1;
		p.stamps = which == 3 ? st : nullptr; p.narrow = narrow;
		if (!narrow) { p.ksplit = 4; p.slab = slab; p.tickets = tickets; }
		launch_skinny(DT_BF16, p, narrow ? 16 : 8, s);
	};
	const char* names[4] = {"ln1+qkv (192 WG x 8 waves)", "c_proj (64 WG x 8)", "ln2+fc+gelu (256 WG x 8)", "mlp.c_proj (64x4 WG x 8)"};
	const int grids[4] = {192, narrow ? 256 : 64, 256, 256};
	for (int which = 0; which < 4; ++which) {
		for (int i = 0; i < NSET; ++i) run_layer(i, nullptr, -1);   // warm
		CK(hipStreamSynchronize(s));
		std::vector<double> ph[6];
		for (int rep = 0; rep < 3; ++rep)
			for (int i = 0; i < NSET; ++i) {
				CK(hipMemsetAsync(stamps, 0, 512 * 8 * 8, s));
				run_layer(i, stamps, which);
				std::vector<unsigned long long> hs(512 * 8);
				CK(hipMemcpyAsync(hs.data(), stamps, 512 * 8 * 8, hipMemcpyDeviceToHost, s));
				CK(hipStreamSynchronize(s));
				unsigned long long t0 = ~0ull, t5 = 0;
				for (int w = 0; w < grids[which]; ++w) { t0 = std::min(t0, hs[w * 8 + 0]); t5 = std::max(t5, hs[w * 8 + 5]); }
				// per-WG phase durations (median over WGs), in 10 ns ticks
				for (int k = 1; k <= 5; ++k) {
					std::vector<double> v;
					for (int w = 0; w < grids[which]; ++w) if (hs[w * 8 + k] && hs[w * 8 + k - 1]) v.push_back((double)hs[w * 8 + k] - (double)hs[w * 8 + k - 1]);
					if (!v.empty()) { std::sort(v.begin(), v.end()); ph[k].push_back(v[v.size() / 2]); }
				}
				ph[0].push_back((double)(t5 - t0));
			}
		auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2] * 0.01; };
		printf("%-28s first-start..last-end %.2f us | per-WG median: ln %.2f  ln-rest+barrier %.2f  weights+mfma %.2f  reduce-barrier %.2f  epilogue %.2f us\n",
			   names[which], med(ph[0]), med(ph[1]), med(ph[2]), med(ph[3]), med(ph[4]), med(ph[5]));
	}
	return 0;
}
