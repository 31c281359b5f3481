// Diagnostic (not a test, not product code): the five launches of one GPT-2 layer of the KV-cached decode step, as libttk issues them
// (bf16, 16 candidates, LayerNorm folded, fragment-order activations), chained over NL layers in ONE captured graph and replayed like a
// token loop -- (a) microseconds per layer, for A/B runs of kernel variants without the Python loop around them, and (b) in-kernel
// timestamps of EVERY wave of the five kernels of one layer in the middle of the chain: the boundary between two kernels (last wave end ->
// first wave start of the next), the start spread of a launch, and the phases inside the waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTTK_STAMPS=2 -I tortoise_tts_amd/csrc tests/diag/ar_chain.cpp -o tests/diag/ar_chain.bin
//   tests/diag/ar_chain.bin [ctx=190] [replays=40]          env: CH_WV_QKV / CH_WV_PROJ / CH_WV_FC / CH_WV_PROJ2 (waves), CH_NARROW, CH_LAYERS
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../tortoise_tts_amd/csrc/skinny.hip"
#include "../../tortoise_tts_amd/csrc/attn.hip"
#include "../../tortoise_tts_amd/csrc/gemv.hip"
bool ttk::g_prof_on = false;
void ttk::prof_start(int, double, hipStream_t) {}
void ttk::prof_stop(hipStream_t) {}
void ttk::prof_pair(int, double, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
using namespace ttk;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)

static int envi(const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; }

constexpr int SLOTS = 520 * 16 * 8;      // stamps of one kernel: [workgroup <= 520][wave <= 16][8]

struct Dist { double p0, p10, p50, p90, p100; };
static Dist dist(std::vector<double> v) {
	Dist d = {0, 0, 0, 0, 0};
	if (v.empty()) return d;
	std::sort(v.begin(), v.end());
	auto q = [&](double f) { return v[std::min(v.size() - 1, (size_t)(f * (v.size() - 1) + 0.5))]; };
	d.p0 = v.front(); d.p10 = q(0.1); d.p50 = q(0.5); d.p90 = q(0.9); d.p100 = v.back();
	return d;
}

int main(int argc, char** argv) {
	const int ctx = argc > 1 ? atoi(argv[1]) : 190;
	const int replays = argc > 2 ? atoi(argv[2]) : 40;
	const int d = 1024, B = 16, H = 16, max_ctx = 336;
	const int NL = envi("CH_LAYERS", 30), SL = NL / 2;
	const int wv_qkv = envi("CH_WV_QKV", 4), wv_proj = envi("CH_WV_PROJ", 4), wv_fc = envi("CH_WV_FC", 4), wv_proj2 = envi("CH_WV_PROJ2", 8);
	const int narrow = envi("CH_NARROW", 4);
	const int lean = envi("CH_LEAN", 1);      // 1: the specialised launches of gemv.hip; 0: k_skinny
	struct Layer { void *wqkv, *wproj, *wfc, *wproj2, *kc, *vc; };
	std::vector<Layer> L(NL);
	const size_t kvb = (size_t)B * H * max_ctx * 64 * 2;
	for (int i = 0; i < NL; ++i) {
		CK(hipMalloc(&L[i].wqkv, (size_t)3 * d * d * 2)); CK(hipMalloc(&L[i].wproj, (size_t)d * d * 2));
		CK(hipMalloc(&L[i].wfc, (size_t)4 * d * d * 2)); CK(hipMalloc(&L[i].wproj2, (size_t)4 * d * d * 2));
		CK(hipMalloc(&L[i].kc, kvb)); CK(hipMalloc(&L[i].vc, kvb));
		CK(hipMemset(L[i].wqkv, 0, (size_t)3 * d * d * 2)); CK(hipMemset(L[i].wproj, 0, (size_t)d * d * 2));
		CK(hipMemset(L[i].wfc, 0, (size_t)4 * d * d * 2)); CK(hipMemset(L[i].wproj2, 0, (size_t)4 * d * d * 2));
		CK(hipMemset(L[i].kc, 0, kvb)); CK(hipMemset(L[i].vc, 0, kvb));
	}
	float *x, *qbuf, *bias; void *xfrag, *ao, *hb; int* dpos; unsigned long long* stamps;
	CK(hipMalloc(&x, B * d * 4)); CK(hipMalloc(&qbuf, B * d * 4)); CK(hipMalloc(&bias, 4 * d * 4));
	CK(hipMalloc(&xfrag, B * d * 2)); CK(hipMalloc(&ao, B * d * 2)); CK(hipMalloc(&hb, B * 4 * d * 2)); CK(hipMalloc(&dpos, 64));
	CK(hipMalloc(&stamps, (size_t)5 * SLOTS * 8));
	CK(hipMemset(x, 0, B * d * 4)); CK(hipMemset(qbuf, 0, B * d * 4)); CK(hipMemset(bias, 0, 4 * d * 4));
	CK(hipMemset(xfrag, 0, B * d * 2)); CK(hipMemset(ao, 0, B * d * 2)); CK(hipMemset(hb, 0, B * 4 * d * 2));
	const int hp[2] = {ctx, 68};
	int pos_slot = -1;
	if (getenv("AC_POS_LINE") && atoi(getenv("AC_POS_LINE"))) { int* w = nullptr; pos_slot = attn_pos_slot_acquire(&w); if (pos_slot >= 0) dpos = w; }      // round 5: the position words in the attention's link-time line
	printf("position words: %s\n", pos_slot >= 0 ? "in the position line (AC_POS_LINE=1)" : "behind the d_pos pointer");
	CK(hipMemcpy(dpos, hp, 8, hipMemcpyHostToDevice));
	hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

	auto layer = [&](int i, unsigned long long* st) {
		SkinnyParams p = {};
		p.Wp = L[i].wqkv; p.N = 3 * d; p.K = d; p.M = B; p.bias = bias; p.g1 = bias; p.a = xfrag; p.lda = d; p.a_frag = 1;
		p.mode = SK_QKV; p.qbuf = qbuf; p.kcache = L[i].kc; p.vcache = L[i].vc; p.d_pos = dpos; p.max_ctx = max_ctx; p.H = H; p.q_scale = 0.125f;
		p.stamps = st ? st + 0 * SLOTS : nullptr;
		GemvParams gq = {};
		gq.Wp = p.Wp; gq.a = xfrag; gq.bias = bias; gq.csum = bias; gq.qbuf = qbuf; gq.kcache = L[i].kc; gq.vcache = L[i].vc; gq.d_pos = dpos;
		gq.M = B; gq.N = 3 * d; gq.K = d; gq.max_ctx = max_ctx; gq.H = H; gq.q_scale = 0.125f; gq.stamps = p.stamps;
		if (!(lean && launch_gemv(DT_BF16, GV_QKV, gq, s))) launch_skinny(DT_BF16, p, wv_qkv, s);
		AttnDecodeParams a = {};
		a.qbuf = qbuf; a.kcache = L[i].kc; a.vcache = L[i].vc; a.d_pos = dpos; a.pos_slot_p1 = pos_slot + 1; a.B = B; a.H = H; a.max_ctx = max_ctx; a.ctx_hint = ctx; a.out = ao; a.out_frag = 1; a.shared_rows = 1;
		a.stamps = st ? st + 1 * SLOTS : nullptr;
		launch_attn_decode(DT_BF16, a, s);
		p = {}; p.Wp = L[i].wproj; p.N = d; p.K = d; p.M = B; p.bias = bias; p.a = ao; p.lda = d; p.a_frag = 1;
		p.mode = SK_RESIDUAL; p.out_f32 = x; p.ldc = d; p.narrow = narrow; p.out_T = xfrag; p.stamps = st ? st + 2 * SLOTS : nullptr;
		GemvParams gp = {};
		gp.Wp = p.Wp; gp.a = ao; gp.bias = bias; gp.out_f32 = x; gp.out_T = xfrag; gp.M = B; gp.N = d; gp.K = d; gp.stamps = p.stamps;
		if (!(lean && narrow == 4 && launch_gemv(DT_BF16, GV_PROJ, gp, s))) launch_skinny(DT_BF16, p, wv_proj, s);
		p = {}; p.Wp = L[i].wfc; p.N = 4 * d; p.K = d; p.M = B; p.bias = bias; p.g1 = bias; p.a = xfrag; p.lda = d; p.a_frag = 1;
		p.mode = SK_ACT_T; p.act = ACT_GELU_NEW; p.out_T = hb; p.out_frag = 1; p.stamps = st ? st + 3 * SLOTS : nullptr;
		GemvParams gf = {};
		gf.Wp = p.Wp; gf.a = xfrag; gf.bias = bias; gf.csum = bias; gf.out_T = hb; gf.M = B; gf.N = 4 * d; gf.K = d; gf.stamps = p.stamps;
		if (!(lean && launch_gemv(DT_BF16, GV_FC, gf, s))) launch_skinny(DT_BF16, p, wv_fc, s);
		p = {}; p.Wp = L[i].wproj2; p.N = d; p.K = 4 * d; p.M = B; p.bias = bias; p.a = hb; p.lda = 4 * d; p.a_frag = 1;
		p.mode = SK_RESIDUAL; p.out_f32 = x; p.ldc = d; p.narrow = narrow; p.out_T = xfrag; p.stamps = st ? st + 4 * SLOTS : nullptr;
		GemvParams g2 = {};
		g2.Wp = p.Wp; g2.a = hb; g2.bias = bias; g2.out_f32 = x; g2.out_T = xfrag; g2.M = B; g2.N = d; g2.K = 4 * d; g2.stamps = p.stamps;
		if (!(lean && narrow == 4 && launch_gemv(DT_BF16, GV_PROJ, g2, s))) launch_skinny(DT_BF16, p, wv_proj2, s);
	};
	for (int i = 0; i < NL; ++i) layer(i, nullptr);      // warm (code objects, first-touch)
	CK(hipStreamSynchronize(s));

	hipGraph_t g; hipGraphExec_t ge, ges;
	CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
	for (int i = 0; i < NL; ++i) layer(i, nullptr);
	CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
	CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
	for (int i = 0; i < NL; ++i) layer(i, i == SL ? stamps : nullptr);
	CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ges, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));

	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		CK(hipEventRecord(e0, s));
		for (int r = 0; r < replays; ++r) CK(hipGraphLaunch(ge, s));
		CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
		float ms; CK(hipEventElapsedTime(&ms, e0, e1));
		if (rep >= 1) best = std::min(best, ms);
	}
	printf("chain: lean %d ctx %d, %d layers x %d replays, waves %d/%d/%d/%d narrow %d: %.3f us per layer (%.3f us per launch)\n", lean, ctx, NL, replays,
		   wv_qkv, wv_proj, wv_fc, wv_proj2, narrow, best * 1e3 / (replays * NL), best * 1e3 / (replays * NL * 5));

	// ---- stamps of layer SL inside the replayed chain (median over several replays of every statistic)
	const char* names[5] = {"c_attn(fold)", "attention", "c_proj", "c_fc(fold)", "mlp.c_proj"};
	const int grids[5] = {192, B * H, narrow ? (d / 16) * narrow : d / 16, 256, narrow ? (d / 16) * narrow : d / 16};
	const int waves[5] = {lean ? 4 : wv_qkv, 16, lean ? 4 : wv_proj, lean ? 4 : wv_fc, lean ? 8 : wv_proj2};
	// stamp slots: skinny 0 start, 2 loads issued, 3 own MFMAs done, 4 after the barrier, 5 stores issued, 6 stores acknowledged
	//              attention 0 start, 1 position + q known, 2 own keys reduced, 3 after the barrier, 5 stores issued, 6 acknowledged
	std::vector<unsigned long long> hs((size_t)5 * SLOTS);
	const int NREP = 9;
	std::vector<double> stat[5][12];
	for (int rep = 0; rep < NREP + 2; ++rep) {
		CK(hipMemsetAsync(stamps, 0, (size_t)5 * SLOTS * 8, s));
		CK(hipGraphLaunch(ge, s)); CK(hipGraphLaunch(ges, s)); CK(hipGraphLaunch(ge, s));
		CK(hipMemcpyAsync(hs.data(), stamps, (size_t)5 * SLOTS * 8, hipMemcpyDeviceToHost, s));
		CK(hipStreamSynchronize(s));
		if (rep < 2) continue;
		double prev_end = 0;
		for (int k = 0; k < 5; ++k) {
			const unsigned long long* st = hs.data() + (size_t)k * SLOTS;
			double t0 = 1e30, t_end = 0, t_ack = 0;
			std::vector<double> starts, durs, ph_issue, ph_data, ph_bar, ph_epi, ph_ack, wgdur;
			for (int w = 0; w < grids[k]; ++w) {
				double ws = 1e30, we = 0;
				for (int v = 0; v < waves[k]; ++v) {
					const unsigned long long* q = st + ((size_t)w * 16 + v) * 8;
					if (!q[0]) continue;
					t0 = std::min(t0, (double)q[0]);
					ws = std::min(ws, (double)q[0]);
					const double end = (double)(q[6] ? q[6] : (q[5] ? q[5] : q[3]));
					we = std::max(we, end);
					t_end = std::max(t_end, (double)std::max(q[5], q[3])); t_ack = std::max(t_ack, end);
				}
				if (we > 0) wgdur.push_back(we - ws);
			}
			for (int w = 0; w < grids[k]; ++w)
				for (int v = 0; v < waves[k]; ++v) {
					const unsigned long long* q = st + ((size_t)w * 16 + v) * 8;
					if (!q[0]) continue;
					starts.push_back((double)q[0] - t0);
					const double issued = (double)(k == 1 ? q[1] : q[2]);
					if (issued) ph_issue.push_back(issued - q[0]);
					const double data = (double)(k == 1 ? q[2] : q[3]);
					if (issued && data) ph_data.push_back(data - issued);
					const double bar = (double)(k == 1 ? q[3] : q[4]);
					if (data && bar) ph_bar.push_back(bar - data);
					if (bar && q[5]) ph_epi.push_back((double)q[5] - bar);
					if (q[5] && q[6]) ph_ack.push_back((double)q[6] - (double)q[5]);
				}
			auto push = [&](int i, double v) { stat[k][i].push_back(v * 0.01); };
			push(0, k == 0 ? 0 : t0 - prev_end);                 // boundary: previous kernel's last acknowledged store -> this kernel's first wave
			const Dist ds = dist(starts); push(1, ds.p50); push(2, ds.p90); push(3, ds.p100);
			push(4, dist(ph_issue).p50); push(5, dist(ph_data).p50); push(6, dist(ph_data).p100); push(7, dist(ph_bar).p50); push(8, dist(ph_epi).p50); push(9, dist(ph_ack).p50);
			push(10, dist(wgdur).p50); push(11, t_ack - t0);
			prev_end = t_ack;
		}
	}
	auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
	printf("%-14s %5s | %8s | start spread p50/p90/max | issue  data(p50/max)  barrier  epilogue  ack | WG dur p50 | first start..last ack\n", "kernel", "WGxW", "boundary");
	double sum = 0;
	for (int k = 0; k < 5; ++k) {
		printf("%-14s %3dx%-2d | %8.2f | %6.2f %6.2f %6.2f       | %5.2f  %5.2f /%5.2f   %5.2f    %5.2f   %5.2f | %7.2f    | %6.2f us\n", names[k], grids[k], waves[k],
			   med(stat[k][0]), med(stat[k][1]), med(stat[k][2]), med(stat[k][3]), med(stat[k][4]), med(stat[k][5]), med(stat[k][6]), med(stat[k][7]), med(stat[k][8]), med(stat[k][9]),
			   med(stat[k][10]), med(stat[k][11]));
		sum += med(stat[k][0]) + med(stat[k][11]);
	}
	printf("sum of boundaries + spans of the stamped layer: %.2f us (boundary of c_attn not included)\n", sum);
	return 0;
}
