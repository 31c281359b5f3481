// One sampled token per candidate row, fused: temperature -> softmax -> multinomial(1) -> finished-row padding -> bookkeeping.
//
// Reference: HF `_sample` as driven by stream_generator.py (warpers :56-101, then HF:generation/utils.py:2894-2937):
//     probs = softmax(scores / T);  next = multinomial(probs, 1);  next = next * unfinished + pad * (1 - unfinished);
//     input_ids = cat(input_ids, next);  unfinished &= next != eos
// ATen's multinomial for one sample is `argmax(probs / q)` with q ~ Exp(1) drawn by `exponential_` on the caller's generator
// (aten/src/ATen/native/Distributions.cpp).  The noise q stays a torch op in the caller so the Philox stream is the reference's; this
// kernel is everything around it, i.e. ~15 tiny elementwise/reduction launches per token collapsed into one.
#include "ttk_common.h"
#include "ttk_kernels.h"
#include "ttk_host.h"

namespace ttk {

constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_NPT = 9;          // elements per thread held in registers on the fast path

__device__ __forceinline__ float block_max(float v, float* red, int tid) {
	v = wave_max(v);
	__syncthreads();
	if ((tid & 63) == 0) red[tid >> 6] = v;
	__syncthreads();
	float m = red[0];
#pragma unroll
	for (int w = 1; w < SAMPLE_THREADS / 64; ++w) m = fmaxf(m, red[w]);
	return m;
}

__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
	v = wave_sum(v);
	__syncthreads();
	if ((tid & 63) == 0) red[tid >> 6] = v;
	__syncthreads();
	float s = 0.f;
#pragma unroll
	for (int w = 0; w < SAMPLE_THREADS / 64; ++w) s += red[w];     // fixed order: the same sum in every thread and every run
	return s;
}

// grid = B rows, 1024 threads.  max, sum of exp, argmax of p / q over the row; rows up to 9216 wide stay in registers between the
// three reductions, wider ones are re-read (32 KB, L2-resident after the first pass).
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_step(const float* scores, int64_t ld, int V, const float* q, int64_t ldq, const unsigned char* suppress, float inv_t,
		int64_t stop_token, int64_t* unfinished, int64_t* tok, int64_t* ids, int64_t ids_ld, int64_t ids_cols, int64_t* col, int64_t* history,
		int64_t hist_ld, int64_t hist_off, int* live_rows, int* all_done) {
	__shared__ float red[SAMPLE_THREADS / 64];
	__shared__ int redi[SAMPLE_THREADS / 64];
	const int b = blockIdx.x, tid = threadIdx.x;
	const float* s = scores + (int64_t)b * ld;
	const float* qq = q + (int64_t)b * ldq;
	// ATen divides a tensor by a host scalar as `x * (1 / t)` with the reciprocal rounded to f32 (BinaryDivTrueKernel.cu), and that
	// is what TemperatureLogitsWarper's `scores / temperature` runs on the GPU; inv_t is that reciprocal.
	const bool scale = inv_t != 1.0f;
	float best = -INFINITY;
	int besti = 0x7fffffff;
	if (V <= SAMPLE_NPT * SAMPLE_THREADS) {
		// the row fits in registers (8194 mel codes = 9 per thread): one trip to memory for scores and noise, everything else on chip
		float v[SAMPLE_NPT], qv[SAMPLE_NPT];
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) {       // unconditional clamped loads: all 18 requests leave before the first use
			const int i = tid + j * SAMPLE_THREADS, ic = i < V ? i : V - 1;
			const float x = s[ic];
			qv[j] = qq[ic];
			const bool sup = suppress && suppress[ic];
			float t = sup ? -INFINITY : x;
			t = scale ? t * inv_t : t;
			v[j] = i < V ? t : -INFINITY;
		}
		float m = -INFINITY;
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) m = fmaxf(m, v[j]);
		m = block_max(m, red, tid);
		float sum = 0.f;
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) { v[j] = expf(v[j] - m); sum += v[j]; }      // exp(-inf) = 0 for the padding lanes
		sum = block_sum(sum, red, tid);
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) {
			const int i = tid + j * SAMPLE_THREADS;
			const float r = (v[j] / sum) / qv[j];
			if (i < V && r > best) { best = r; besti = i; }          // strict: the lowest index wins a tie, as in ATen's argmax
		}
	} else {
		float m = -INFINITY;
		for (int i = tid; i < V; i += SAMPLE_THREADS) { float v = (suppress && suppress[i]) ? -INFINITY : s[i]; v = scale ? v * inv_t : v; m = fmaxf(m, v); }
		m = block_max(m, red, tid);
		float sum = 0.f;
		for (int i = tid; i < V; i += SAMPLE_THREADS) { float v = (suppress && suppress[i]) ? -INFINITY : s[i]; v = scale ? v * inv_t : v; sum += expf(v - m); }
		sum = block_sum(sum, red, tid);
		for (int i = tid; i < V; i += SAMPLE_THREADS) {
			float v = (suppress && suppress[i]) ? -INFINITY : s[i]; v = scale ? v * inv_t : v;
			const float r = (expf(v - m) / sum) / qq[i];
			if (r > best) { best = r; besti = i; }
		}
	}
	// wave then block reduction of (value, index) with the same tie rule
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		const float ov = __shfl_xor(best, off);
		const int oi = __shfl_xor(besti, off);
		if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; }
	}
	__syncthreads();
	if ((tid & 63) == 0) { red[tid >> 6] = best; redi[tid >> 6] = besti; }
	__syncthreads();
	if (tid == 0) {
		for (int w = 1; w < SAMPLE_THREADS / 64; ++w)
			if (red[w] > best || (red[w] == best && redi[w] < besti)) { best = red[w]; besti = redi[w]; }
		if (besti >= V) besti = 0;                       // all-NaN / empty row: ATen returns index 0 as well
		const int64_t live = unfinished[b];
		const int64_t nxt = (int64_t)besti * live + stop_token * (1 - live);
		tok[b] = nxt;
		const int64_t c = col[b];
		if (c < ids_cols) ids[(int64_t)b * ids_ld + c] = nxt;
		if (history) history[(int64_t)b * hist_ld + hist_off + c] = nxt;
		col[b] = c + 1;
		const int64_t still = live * (nxt != stop_token ? 1 : 0);
		unfinished[b] = still;
		// stopping criterion without a host round trip per token: the row that finishes last raises a flag the host can poll (the
		// flag may live in pinned host memory; it only ever goes 0 -> 1 within a generation, so a late read is merely late)
		if (live_rows && live != 0 && still == 0) {
			if (__hip_atomic_fetch_add(live_rows, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1 && all_done)
				__hip_atomic_store(all_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

}  // namespace ttk

extern "C" int ttk_sample_step(const float* scores, int64_t ld, int B, int V, const float* q, int64_t ldq, const unsigned char* suppress, float temperature,
		int64_t stop_token, int64_t* unfinished, int64_t* tok, int64_t* ids, int64_t ids_ld, int64_t ids_cols, int64_t* col, int64_t* history, int64_t hist_ld,
		int64_t hist_off, int* live_rows, int* all_done, void* stream) {
	using namespace ttk;
	TTK_REQUIRE(scores && q && unfinished && tok && ids && col, TTK_E_ARG, "ttk_sample_step: null argument");
	TTK_REQUIRE(B >= 1 && V >= 1 && ld >= V && ldq >= V, TTK_E_ARG, "ttk_sample_step: bad shape (B %d, V %d)", B, V);
	TTK_REQUIRE(temperature > 0.f, TTK_E_ARG, "ttk_sample_step: temperature must be positive");
	hipLaunchKernelGGL(k_sample_step, dim3(B), dim3(SAMPLE_THREADS), 0, (hipStream_t)stream, scores, ld, V, q, ldq, suppress, 1.0f / temperature, stop_token, unfinished, tok,
					   ids, ids_ld, ids_cols, col, history, hist_ld, hist_off, live_rows, all_done);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}
