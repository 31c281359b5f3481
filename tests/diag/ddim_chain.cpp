// Diagnostic (not a test, not product code): the launches of one DiffusionLayer of the DDIM loop's body as libttk issues them at configs[1]'s
// size (bf16, cond + cond-free batch of 2, T = 1088: M = 2176 rows, C = 1024) -- ResBlock {GroupNorm-apply, 1x1 GEMM, GroupNorm-apply, k=3 GEMM +
// residual} + AttentionBlock {GroupNorm-apply, QKV GEMM, attention, proj GEMM + residual} -- chained over NL layers with distinct weights in ONE
// captured graph: (a) microseconds per layer, and (b) in-kernel timestamps of EVERY wave of the eight kernels of one layer in the middle of the
// chain: kernel boundaries, start spread, and the phases inside (GEMM: first requests issued, first tile landed, k-loop, epilogue, stores acknowledged).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTTK_STAMPS=2 -I tortoise_tts_amd/csrc tests/diag/ddim_chain.cpp -o tests/diag/ddim_chain.bin
//   tests/diag/ddim_chain.bin [replays=10]          env: DC_LAYERS (20), DC_T (1088), DC_NB (2), DC_PF (1: GroupNorm-apply / attention touch the next GEMM's weights), DC_RANDOM (0; 1 = hashed operands instead of zeros)
//   -DTTK_CLOCK_STAMPS (with -DTTK_STAMPS=2): s_memtime / s_memrealtime around the GEMMs' k-loop and the attention's key loop -> in-kernel clock per kernel (round 6)
//   DC_SIDE=k   (round 5) the timed loop runs TWO streams: step j+1's conditioning_timestep_integrator (3 of these layers, own weights and
//               activations) beside the 13 layers of step j's body (csrc/diff.hip).  With DC_SIDE=k a second stream replays k layers (own buffers, own weights) with
//               every replay of the main chain, so the stamps and the per-layer time are taken under the contention the bench has (k = 5 beside 20 main layers ~ 3 : 13).
//               DC_STAMP_LAYER picks the stamped layer (default the middle one; 2 lies under the side lane, which runs beside the first layers of every replay).
//   DC_EAGER=1  counter mode: no graph, no stamps -- `replays` x DC_LAYERS layers launched eagerly (~10^3 dispatches: inside what `rocprofv3 --pmc` survives on this
//               image, which faults on a captured graph's dispatches), then exit.  Build WITHOUT -DTTK_STAMPS for it (tests/diag/pmc_kloop.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../../tortoise_tts_amd/csrc/gemm.hip"
#include "../../tortoise_tts_amd/csrc/norm.hip"
#include "../../tortoise_tts_amd/csrc/attn.hip"
bool ttk::g_prof_on = false;
void ttk::prof_start(int, double, hipStream_t) {}
void ttk::prof_stop(hipStream_t) {}
void ttk::prof_pair(int, double, hipEvent_t* a, hipEvent_t* b) { *a = nullptr; *b = nullptr; }
using namespace ttk;
#ifdef TTK_STAMPS
#define SETST(P, V) (P).stamps = (V)
#else
#define SETST(P, V) (void)(V)      // the plain build (counter mode) has no stamp field
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s (line %d)\n", #x, hipGetErrorString(e), __LINE__); return 1; } } while (0)
static int envi(const char* n, int d) { const char* e = getenv(n); return e ? atoi(e) : d; }
// DC_RANDOM=1: operands from a hash instead of zeros (the clock a chip holds under load depends on the data: MI355X_MICROARCH.md, DVFS give-back items 1 and 7)
__global__ void fill_bf16(unsigned short* p, size_t n, unsigned seed, float scale) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
	const float v = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
	p[i] = (unsigned short)(__float_as_uint(v) >> 16);
}
__global__ void fill_f32(float* p, size_t n, unsigned seed, float scale) {
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
	p[i] = ((int)(h & 0xffff) - 32768) * (scale / 32768.f);
}

constexpr int MAXWG = 1200;
constexpr size_t SLOTS = (size_t)MAXWG * 16 * 8;

static double q(std::vector<double> v, double f) {
	if (v.empty()) return 0;
	std::sort(v.begin(), v.end());
	return v[std::min(v.size() - 1, (size_t)(f * (v.size() - 1) + 0.5))];
}

int main(int argc, char** argv) {
	const int replays = argc > 1 ? atoi(argv[1]) : 10;
	const int NL = envi("DC_LAYERS", 20), SL = envi("DC_STAMP_LAYER", NL / 2), T = envi("DC_T", 1088), nb = envi("DC_NB", 2), C = 1024, H = 16, M = nb * T, pf = envi("DC_PF", 1);
	struct Layer { void *w1, *w3, *wqkv, *wproj; };
	std::vector<Layer> L(NL);
	for (int i = 0; i < NL; ++i) {
		CK(hipMalloc(&L[i].w1, (size_t)C * C * 2)); CK(hipMalloc(&L[i].w3, (size_t)3 * C * C * 2)); CK(hipMalloc(&L[i].wqkv, (size_t)3 * C * C * 2)); CK(hipMalloc(&L[i].wproj, (size_t)C * C * 2));
		CK(hipMemset(L[i].w1, 0, (size_t)C * C * 2)); CK(hipMemset(L[i].w3, 0, (size_t)3 * C * C * 2)); CK(hipMemset(L[i].wqkv, 0, (size_t)3 * C * C * 2)); CK(hipMemset(L[i].wproj, 0, (size_t)C * C * 2));
	}
	const int NSIDE = envi("DC_SIDE", 0), EAGER = envi("DC_EAGER", 0);
	std::vector<Layer> LS(NSIDE);
	for (int i = 0; i < NSIDE; ++i) {
		CK(hipMalloc(&LS[i].w1, (size_t)C * C * 2)); CK(hipMalloc(&LS[i].w3, (size_t)3 * C * C * 2)); CK(hipMalloc(&LS[i].wqkv, (size_t)3 * C * C * 2)); CK(hipMalloc(&LS[i].wproj, (size_t)C * C * 2));
		CK(hipMemset(LS[i].w1, 0, (size_t)C * C * 2)); CK(hipMemset(LS[i].w3, 0, (size_t)3 * C * C * 2)); CK(hipMemset(LS[i].wqkv, 0, (size_t)3 * C * C * 2)); CK(hipMemset(LS[i].wproj, 0, (size_t)C * C * 2));
	}
	float *gam, *bet, *bias, *relb; unsigned long long* stamps;
	struct Bufs { float *x, *hf, *ms; void *a, *qkv, *ao; };
	const int nch = gn_num_chunks(T, C);
	CK(hipMalloc(&gam, C * 4)); CK(hipMalloc(&bet, C * 4)); CK(hipMalloc(&bias, 3 * C * 4)); CK(hipMalloc(&relb, H * 129 * 4));
	CK(hipMemset(gam, 0, C * 4)); CK(hipMemset(bet, 0, C * 4)); CK(hipMemset(bias, 0, 3 * C * 4)); CK(hipMemset(relb, 0, H * 129 * 4));
	CK(hipMalloc(&stamps, 8 * SLOTS * 8));
	Bufs BF[2];
	for (int b = 0; b < (NSIDE ? 2 : 1); ++b) {
		Bufs& B = BF[b];
		CK(hipMalloc(&B.x, (size_t)M * C * 4)); CK(hipMalloc(&B.hf, (size_t)M * C * 4)); CK(hipMalloc(&B.ms, (size_t)nb * 32 * nch * 3 * 4));
		CK(hipMalloc(&B.a, (size_t)M * C * 2)); CK(hipMalloc(&B.qkv, (size_t)M * 3 * C * 2)); CK(hipMalloc(&B.ao, (size_t)M * C * 2));
		CK(hipMemset(B.x, 0, (size_t)M * C * 4)); CK(hipMemset(B.hf, 0, (size_t)M * C * 4));
		CK(hipMemset(B.a, 0, (size_t)M * C * 2)); CK(hipMemset(B.qkv, 0, (size_t)M * 3 * C * 2)); CK(hipMemset(B.ao, 0, (size_t)M * C * 2));
		// valid statistics: every chunk (64 rows x 32 channels) count 2048, mean 0, M2 2048
		std::vector<float> h((size_t)nb * 32 * nch * 3);
		for (size_t i = 0; i < h.size(); i += 3) { h[i] = 2048.f; h[i + 1] = 0.f; h[i + 2] = 2048.f; }
		CK(hipMemcpy(B.ms, h.data(), h.size() * 4, hipMemcpyHostToDevice));
	}
	const int RANDOM = envi("DC_RANDOM", 0);
	if (RANDOM) {      // weights ~ U(-0.03, 0.03) (1 / sqrt(1024)), residual stream ~ U(-1, 1), GroupNorm gamma = 1: activations of order one through every layer
		auto fb = [&](void* q, size_t n, unsigned seed, float sc) { fill_bf16<<<(unsigned)((n + 255) / 256), 256>>>((unsigned short*)q, n, seed, sc); };
		auto ff = [&](float* q, size_t n, unsigned seed, float sc) { fill_f32<<<(unsigned)((n + 255) / 256), 256>>>(q, n, seed, sc); };
		for (int i = 0; i < NL; ++i) { fb(L[i].w1, (size_t)C * C, 11 + i, 0.03f); fb(L[i].w3, (size_t)3 * C * C, 211 + i, 0.03f); fb(L[i].wqkv, (size_t)3 * C * C, 411 + i, 0.03f); fb(L[i].wproj, (size_t)C * C, 611 + i, 0.03f); }
		for (int i = 0; i < NSIDE; ++i) { fb(LS[i].w1, (size_t)C * C, 1011 + i, 0.03f); fb(LS[i].w3, (size_t)3 * C * C, 1211 + i, 0.03f); fb(LS[i].wqkv, (size_t)3 * C * C, 1411 + i, 0.03f); fb(LS[i].wproj, (size_t)C * C, 1611 + i, 0.03f); }
		for (int b = 0; b < (NSIDE ? 2 : 1); ++b) { ff(BF[b].x, (size_t)M * C, 5 + b, 1.f); ff(BF[b].hf, (size_t)M * C, 7 + b, 1.f); fb(BF[b].a, (size_t)M * C, 9 + b, 1.f); fb(BF[b].qkv, (size_t)M * 3 * C, 13 + b, 1.f); fb(BF[b].ao, (size_t)M * C, 15 + b, 1.f); }
		std::vector<float> one(C, 1.f); CK(hipMemcpy(gam, one.data(), C * 4, hipMemcpyHostToDevice));
		ff(relb, (size_t)H * 129, 3, 0.5f);
		CK(hipDeviceSynchronize());
	}
	hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	hipStream_t cur = s; Bufs* CB = &BF[0];      // the lane the helpers below enqueue into

	auto gn = [&](const float* src, const void* next_w, int64_t next_bytes, int taps, unsigned long long* st) {
		GnApplyParams p = {};
		p.x = src; p.ms = CB->ms; p.gamma = gam; p.beta = bet; p.nb = nb; p.T = T; p.Tout = T; p.C = C; p.nchunks = nch; p.act = ACT_SILU; p.out = CB->a;
		if (pf) { p.pf = next_w; p.pf_bytes = next_bytes; p.pf_taps = taps; }
		SETST(p, st);
		launch_gn_apply(DT_BF16, p, cur);
	};
	auto layer_of = [&](const Layer& W, unsigned long long* st) {
		float *x = CB->x, *hf = CB->hf, *ms = CB->ms; void *a = CB->a, *qkv = CB->qkv, *ao = CB->ao;
		auto S = [&](int k) { return st ? st + k * SLOTS : nullptr; };
		// ResBlock
		gn(x, W.w1, (int64_t)C * C * 2, 1, S(0));
		GemmParams g = {};
		g.nseg = 1; g.seg[0] = {a, C, 0, 0}; g.W = W.w1; g.ldw = C; g.M = M; g.N = C; g.K = C; g.bias = bias; g.C = hf; g.ldc = C; g.out_f32 = 1; g.gn_part = ms; g.gn_T = T; SETST(g, S(1));
		launch_gemm(DT_BF16, g, cur);
		gn(hf, W.w3, (int64_t)C * C * 2, 3, S(2));
		g = {};
		g.nseg = 3; for (int j = 0; j < 3; ++j) g.seg[j] = {a, C, j - 1, (int64_t)j * C * C};
		g.W = W.w3; g.ldw = C; g.M = M; g.N = C; g.K = C; g.rows_per_batch = T; g.bias = bias; g.residual = x; g.ldr = C; g.C = x; g.ldc = C; g.out_f32 = 1; g.gn_part = ms; g.gn_T = T; SETST(g, S(3));
		launch_gemm(DT_BF16, g, cur);
		// AttentionBlock
		gn(x, W.wqkv, (int64_t)3 * C * C * 2, 1, S(4));
		g = {};
		g.nseg = 1; g.seg[0] = {a, C, 0, 0}; g.W = W.wqkv; g.ldw = C; g.M = M; g.N = 3 * C; g.K = C; g.bias = bias; g.C = qkv; g.ldc = 3 * C; SETST(g, S(5));
		launch_gemm(DT_BF16, g, cur);
		AttnParams at = {};
		at.qkv = qkv; at.ld = 3 * C; at.q_off = 0; at.k_off = 64; at.v_off = 128; at.head_stride = 192; at.out = ao; at.ldo = C; at.nb = nb; at.T = T; at.H = H; at.bias = relb; at.scale = 0.125f;
		if (pf) { at.pf = W.wproj; at.pf_bytes = (int64_t)C * C * 2; at.pf_taps = 1; }
		SETST(at, S(6));
		launch_attn_fwd(DT_BF16, at, cur);
		g = {};
		g.nseg = 1; g.seg[0] = {ao, C, 0, 0}; g.W = W.wproj; g.ldw = C; g.M = M; g.N = C; g.K = C; g.bias = bias; g.residual = x; g.ldr = C; g.C = x; g.ldc = C; g.out_f32 = 1; g.gn_part = ms; g.gn_T = T; SETST(g, S(7));
		launch_gemm(DT_BF16, g, cur);
	};
	auto layer = [&](int i, unsigned long long* st) { layer_of(L[i], st); };
	auto side_layers = [&]() { cur = s2; CB = &BF[1]; for (int i = 0; i < NSIDE; ++i) layer_of(LS[i], nullptr); cur = s; CB = &BF[0]; };
	if (EAGER) {      // counter mode: bounded eager loop, nothing else
		for (int r = 0; r < replays; ++r) { for (int i = 0; i < NL; ++i) layer(i, nullptr); if (NSIDE) side_layers(); }
		CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
		printf("ddim chain (eager counter mode): T %d nb %d, %d layers x %d replays = %d dispatches%s\n", T, nb, NL, replays, NL * replays * 8, NSIDE ? " + side lane" : "");
		return 0;
	}
	for (int i = 0; i < NL; ++i) layer(i, nullptr);
	CK(hipStreamSynchronize(s));
	hipGraph_t gr; hipGraphExec_t ge, ges;
	CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
	for (int i = 0; i < NL; ++i) layer(i, nullptr);
	CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0)); CK(hipGraphDestroy(gr));
	CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
	for (int i = 0; i < NL; ++i) layer(i, i == SL ? stamps : nullptr);
	CK(hipStreamEndCapture(s, &gr)); CK(hipGraphInstantiate(&ges, gr, nullptr, nullptr, 0)); CK(hipGraphDestroy(gr));
	hipGraphExec_t gside = nullptr;
	if (NSIDE) {
		side_layers(); CK(hipStreamSynchronize(s2));
		CK(hipStreamBeginCapture(s2, hipStreamCaptureModeThreadLocal));
		side_layers();
		CK(hipStreamEndCapture(s2, &gr)); CK(hipGraphInstantiate(&gside, gr, nullptr, nullptr, 0)); CK(hipGraphDestroy(gr));
	}
	hipEvent_t efork; CK(hipEventCreateWithFlags(&efork, hipEventDisableTiming));
	// the side lane of a replay starts WITH that replay's main chain (an event on the main stream releases it), as step j+1's integrator starts with step j's body
	auto launch_main = [&](hipGraphExec_t g) { if (gside) { (void)hipEventRecord(efork, s); (void)hipStreamWaitEvent(s2, efork, 0); (void)hipGraphLaunch(gside, s2); } return hipGraphLaunch(g, s); };
	hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
	float best = 1e9f, best_eager = 1e9f;
	for (int rep = 0; rep < 4; ++rep) {
		CK(hipEventRecord(e0, s));
		for (int r = 0; r < replays; ++r) CK(launch_main(ge));
		CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipStreamSynchronize(s2));
		float msv; CK(hipEventElapsedTime(&msv, e0, e1));
		if (rep >= 1) best = std::min(best, msv);
	}
	for (int rep = 0; rep < 3; ++rep) {
		CK(hipEventRecord(e0, s));
		for (int r = 0; r < replays; ++r) for (int i = 0; i < NL; ++i) layer(i, nullptr);
		CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
		float msv; CK(hipEventElapsedTime(&msv, e0, e1));
		if (rep >= 1) best_eager = std::min(best_eager, msv);
	}
	if (NSIDE) printf("side lane: %d layers on a second stream with every replay of the main chain\n", NSIDE);
	printf("ddim chain: T %d nb %d (M = %d), %d layers x %d replays, prefetch %d: graph %.2f us per layer (%.2f per launch), eager %.2f us per layer\n", T, nb, M, NL, replays, pf,
		   best * 1e3 / (replays * NL), best * 1e3 / (replays * NL * 8), best_eager * 1e3 / (replays * NL));

	const char* names[8] = {"gn_apply (res in)", "gemm 1x1", "gn_apply (res out)", "gemm k=3 + res", "gn_apply (attn)", "gemm qkv", "attention", "gemm proj + res"};
	const bool is_gemm[8] = {false, true, false, true, false, true, false, true};
	std::vector<unsigned long long> hs(8 * SLOTS);
	std::vector<double> stat[8][10];
	std::vector<double> clk[8], kcyc[8];      // -DTTK_CLOCK_STAMPS: per wave, shader cycles / 100 MHz ticks of the k-loop (slot 6)
	std::vector<double> aph[2][4];            // -DTTK_ATTN_PHASES: attention key-loop phases (slots 1, 2), [SIMD 0 | SIMDs 1..3][phase]
	for (int rep = 0; rep < 9; ++rep) {
		CK(hipMemsetAsync(stamps, 0, 8 * SLOTS * 8, s));
		CK(launch_main(ge)); CK(launch_main(ges)); CK(launch_main(ge));
		CK(hipMemcpyAsync(hs.data(), stamps, 8 * SLOTS * 8, hipMemcpyDeviceToHost, s));
		CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
		if (rep < 2) continue;
		double prev_end = 0;
		for (int k = 0; k < 8; ++k) {
			const unsigned long long* st = hs.data() + k * SLOTS;
			double t0 = 1e30, t_ack = 0;
			std::vector<double> starts, ph[5], wgd;
			int nw = 0;
			for (int w = 0; w < MAXWG; ++w) {
				double ws = 1e30, we = 0;
				for (int v = 0; v < 16; ++v) {
					const unsigned long long* p = st + ((size_t)w * 16 + v) * 8;
					if (!p[0]) continue;
					++nw;
					t0 = std::min(t0, (double)p[0]); ws = std::min(ws, (double)p[0]);
					const double end = (double)(p[5] ? p[5] : p[4]);
					we = std::max(we, end); t_ack = std::max(t_ack, end);
				}
				if (we > 0) wgd.push_back(we - ws);
			}
			for (int w = 0; w < MAXWG; ++w)
				for (int v = 0; v < 16; ++v) {
					const unsigned long long* p = st + ((size_t)w * 16 + v) * 8;
					if (!p[0]) continue;
					starts.push_back((double)p[0] - t0);
					if (p[6] && (p[6] & 0xffffffffull)) { clk[k].push_back((double)(p[6] >> 32) / (double)(p[6] & 0xffffffffull) * 0.1); kcyc[k].push_back((double)(p[6] >> 32)); }
					if (is_gemm[k]) {
						if (p[1]) ph[0].push_back((double)p[1] - p[0]);
						if (p[1] && p[2]) ph[1].push_back((double)p[2] - p[1]);
						if (p[2] && p[3]) ph[2].push_back((double)p[3] - p[2]);
						if (p[3] && p[4]) ph[3].push_back((double)p[4] - p[3]);
						if (p[4] && p[5]) ph[4].push_back((double)p[5] - p[4]);
					} else if (k == 6) {
						if (p[1] && p[2]) {      // -DTTK_ATTN_PHASES: per-wave cycle sums of the key loop's phases, waves of SIMD 0 (three per SIMD in a 9-tile workgroup) kept apart
							const int grp = (v & 3) == 0 ? 0 : 1;
							aph[grp][0].push_back((double)(p[1] >> 32)); aph[grp][1].push_back((double)(p[1] & 0xffffffffull));
							aph[grp][2].push_back((double)(p[2] >> 32)); aph[grp][3].push_back((double)(p[2] & 0xffffffffull));
						}
						if (p[4]) ph[2].push_back((double)p[4] - p[0]);
						if (p[4] && p[5]) ph[4].push_back((double)p[5] - p[4]);
					} else {
						if (p[1]) ph[0].push_back((double)p[1] - p[0]);
						if (p[1] && p[2]) ph[1].push_back((double)p[2] - p[1]);
						if (p[2] && p[4]) ph[3].push_back((double)p[4] - p[2]);
						if (p[4] && p[5]) ph[4].push_back((double)p[5] - p[4]);
					}
				}
			auto push = [&](int i, double v) { stat[k][i].push_back(v * 0.01); };
			push(0, k == 0 ? 0 : t0 - prev_end);
			push(1, q(starts, 0.5)); push(2, q(starts, 1.0));
			for (int i = 0; i < 5; ++i) push(3 + i, q(ph[i], 0.5));
			push(8, q(wgd, 0.5)); push(9, t_ack - t0);
			prev_end = t_ack;
			if (rep == 2) stat[k][0].push_back(0), stat[k][0].pop_back();
			(void)nw;
		}
	}
	auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
	for (int g = 0; g < 2; ++g)
		if (!aph[g][0].empty())
			printf("attention phases, waves on %s (%zu stamps), shader cycles per wave summed over the key loop, median: barriers + staging %.0f | K reads + QK^T %.0f | softmax %.0f | V reads + PV %.0f\n",
				   g == 0 ? "SIMD 0    " : "SIMDs 1..3", aph[g][0].size(), med(aph[g][0]), med(aph[g][1]), med(aph[g][2]), med(aph[g][3]));
	printf("%-20s | %8s | start p50 / max | issue   1st-tile  k-loop   epilogue  ack  (gn: loads, merge+barrier, -, apply+stores, ack) | WG dur p50 | first start..last ack\n", "kernel", "boundary");
	double sum = 0;
	for (int k = 0; k < 8; ++k) {
		printf("%-20s | %8.2f | %6.2f %6.2f   | %5.2f   %5.2f    %6.2f   %5.2f    %5.2f | %8.2f   | %7.2f us\n", names[k], med(stat[k][0]), med(stat[k][1]), med(stat[k][2]),
			   med(stat[k][3]), med(stat[k][4]), med(stat[k][5]), med(stat[k][6]), med(stat[k][7]), med(stat[k][8]), med(stat[k][9]));
		sum += med(stat[k][0]) + med(stat[k][9]);
	}
	printf("sum of boundaries + spans of the stamped layer: %.2f us\n", sum);
	for (int k = 0; k < 8; ++k)
		if (!clk[k].empty()) printf("in-kernel clock, %-18s: %.3f GHz median over %zu wave stamps (p10 %.3f, p90 %.3f); main loop %.0f shader cycles median\n", names[k], q(clk[k], 0.5), clk[k].size(), q(clk[k], 0.1), q(clk[k], 0.9), q(kcyc[k], 0.5));
	return 0;
}
