// One sampled token per candidate row, fused: the logits processors and warpers of the reference's sample branch, softmax,
// multinomial(1), finished-row padding, bookkeeping and (AR-aware entry) the next decode step's input embedding.
//
// Reference: HF `_sample` as driven by stream_generator.py (processors / warpers :56-101, then HF:generation/utils.py:2894-2937):
//     scores = top_p(top_k(temperature(suppress(repetition_penalty(input_ids, logits)))))
//     probs = softmax(scores);  next = multinomial(probs, 1);  next = next * unfinished + pad * (1 - unfinished);
//     input_ids = cat(input_ids, next);  unfinished &= next != eos
// and the first lines of the following forward (unified_voice.py:212-214): emb = mel_embedding(next) + mel_pos_embedding[k + 1].
// ATen's multinomial for one sample is `argmax(probs / q)` with q ~ Exp(1) drawn by `exponential_` on the caller's generator
// (aten/src/ATen/native/Distributions.cpp).  The noise q stays a torch op in the caller so the Philox stream is the reference's; this
// kernel is everything around it, i.e. the ~15-40 tiny elementwise / sort / reduction launches per token collapsed into one, with no
// host round trip, so the whole token step stays inside one captured HIP graph for every warper combination of the CLI
// (__main__.py:17-21: temperature 0.8, top-k 16, top-p 1, repetition penalty 1).
//
// Selection without sorting.  top-k needs the k-th largest score, top-p the smallest score whose ascending cumulative probability
// exceeds 1 - top_p: both are a 4-pass radix descent over the order-preserving 32-bit key of the float (256-bin histogram in LDS per
// pass) -- counts for top-k, fixed-point probability mass (p * 2^40 as 64-bit integers, so the sums are exact and independent of the
// order in which the atomics land: bitwise reproducible) for top-p.  Everything at or above the threshold VALUE is kept, which is what
// `scores < kth` / the sorted cumulative sum do except inside a group of exactly equal scores that straddles the top-p boundary.
// Known deviation of top-p: HF's TopPLogitsWarper takes the cut from an f32 `cumsum` (a parallel scan) of the sorted f32
// probabilities compared with 1 - top_p, i.e. from sums that carry f32 rounding; here the masses are exact 2^-40 fixed-point integers.  A row
// whose ascending cumulative mass at some token ties with 1 - top_p within f32 rounding (~1e-6) can therefore keep one token more or less than
// the torch chain.  Measured (tests/test_gpu_parity.py::test_top_p_boundary_stress, 3072 rows x 8 values of top_p): a fraction of a percent of
// the rows, every one of them such a tie.  top_p is 1 (off) in TTS.inference's and the CLI's defaults; a caller that needs HF's rounding bit for
// bit builds the model with hf_exact_top_p=True (tortoise_tts_amd/autoregressive.py): the torch warper then runs in front of this kernel.
#include "ttk_common.h"
#include "ttk_kernels.h"
#include "ttk_host.h"
#include "ttk_rng.h"

namespace ttk {

constexpr int SAMPLE_THREADS = 1024;
constexpr int SAMPLE_NPT = 9;          // elements per thread held in registers on the fast path
constexpr int SAMPLE_MAXV = SAMPLE_NPT * SAMPLE_THREADS;

static_assert(sizeof(ttk_sample_args) == 168, "ttk_sample_args layout (tortoise_tts_amd/_lib.py: SampleArgs mirrors it)");

struct SampleParams {
	const float* scores; int64_t ld; int V;
	const float* q; int64_t ldq;
	const unsigned char* suppress; float inv_t;
	int top_k; float top_p; float penalty, inv_penalty;          // 0 / >= 1 / 1 = off
	float typical_mass;                                           // TypicalLogitsWarper (unified_voice.py:47-75); 0 / >= 1 = off
	int64_t stop_token;
	int64_t *unfinished, *tok, *ids; int64_t ids_ld, ids_cols; int64_t* col;
	int64_t* history; int64_t hist_ld, hist_off;
	int *live_rows, *all_done;
	// next decode step's input row (AR-aware entry only): x[b] = emb[next] + pos[col + 2]
	const float *emb, *pos; float* x_out; int d, pos_rows;
	void* x_frag; int x_frag_f32;      // optional copy of the same rows in A-fragment order (T-typed; x_frag_f32 = its ttk::ElemKind) for a folded-LayerNorm first launch
};

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

// order-preserving key: a < b  <=>  key(a) < key(b)   (-inf lowest; NaNs sort to the ends and are not expected here).  -0.0 takes +0.0's key:
// the torch chain compares VALUES (`scores < kth`), for which the two zeros are equal; as bit patterns -0.0 would sort below +0.0 and a k-th
// largest score of +0.0 would drop the -0.0 entries torch keeps.
__device__ __forceinline__ unsigned fkey(float f) {
	const unsigned u = __float_as_uint(f + 0.0f);      // -0.0 + 0.0 = +0.0 (round to nearest); every other value unchanged
	return (u >> 31) ? ~u : (u | 0x80000000u);
}

// One level of the radix descent, run by wave 0 over the 256 bins `h` (counts or masses): walking the bins in DESCENDING (top-k) or
// ASCENDING (top-p) order, find the first bin at which the running total reaches past `target` -- descending: total >= target (the
// k-th largest lies in it), ascending: total > target (the first kept element lies in it).  Returns the bin and the total BEFORE it.
template <typename C, bool DESC>
__device__ __forceinline__ void pick_bin(const C* h, C target, int lane, int& bin_out, C& before_out) {
	C g[4], s = 0;
#pragma unroll
	for (int i = 0; i < 4; ++i) { const int b = DESC ? 255 - (4 * lane + i) : 4 * lane + i; g[i] = h[b]; s += g[i]; }
	C incl = s;
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const C o = __shfl_up(incl, off);
		if (lane >= off) incl += o;
	}
	const C excl = incl - s;
	const bool hit = DESC ? (incl >= target) : (incl > target);
	const unsigned long long m = __ballot(hit);
	const int first = m ? __ffsll((long long)m) - 1 : 63;        // no lane reaches it (rounding of the total): take the last bin
	int bin = DESC ? 255 - (4 * first + 3) : 4 * first + 3;
	C before = excl + g[0] + g[1] + g[2];
	if (lane == first) {
		C run = excl;
		bool found = false;
#pragma unroll
		for (int i = 0; i < 4; ++i) {
			const bool here = DESC ? (run + g[i] >= target) : (run + g[i] > target);
			if (!found && (here || i == 3)) { bin = DESC ? 255 - (4 * lane + i) : 4 * lane + i; before = run; found = true; }
			run += g[i];
		}
	}
	bin_out = __shfl(bin, first);
	before_out = __shfl(before, first);
}

// grid = B rows, 1024 threads.  Rows up to 9216 wide (8194 mel codes = 9 per thread) stay in registers from the one trip to memory
// to the argmax; wider rows take the plain three-pass path, which supports suppress + temperature only.
__global__ __launch_bounds__(SAMPLE_THREADS) void k_sample_step(SampleParams p) {
	__shared__ float red[SAMPLE_THREADS / 64];
	__shared__ int redi[SAMPLE_THREADS / 64];
	__shared__ unsigned seen[SAMPLE_MAXV / 32];                  // repetition penalty: bit i = token i occurs in input_ids
	__shared__ unsigned hist32[256];
	__shared__ unsigned long long hist64[256];
	__shared__ int s_bin;
	__shared__ unsigned long long s_before;
	__shared__ long long s_next;
	const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
	const int V = p.V;
	const float* s = p.scores + (int64_t)b * p.ld;
	const float* qq = p.q + (int64_t)b * p.ldq;
	// ATen divides a tensor by a host scalar as `x * (1 / t)` with the reciprocal rounded to f32 (BinaryDivTrueKernel.cu), and that
	// is what TemperatureLogitsWarper's `scores / temperature` and the penalty's `score / penalty` run on the GPU.
	const bool scale = p.inv_t != 1.0f;
	const int64_t c0 = p.col[b];
	float best = -INFINITY;
	int besti = 0x7fffffff;
	if (V <= SAMPLE_MAXV) {
		float v[SAMPLE_NPT], qv[SAMPLE_NPT];
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) {       // unconditional clamped loads: all 18 requests leave before the first use
			const int i = tid + j * SAMPLE_THREADS, ic = i < V ? i : V - 1;
			v[j] = s[ic];
			qv[j] = qq[ic];
		}
		// ---- RepetitionPenaltyLogitsProcessor: every id present in input_ids (prefix ids + tokens generated so far), once
		if (p.penalty != 1.0f && p.history) {
			for (int i = tid; i < SAMPLE_MAXV / 32; i += SAMPLE_THREADS) seen[i] = 0;
			__syncthreads();
			const int64_t* hrow = p.history + (int64_t)b * p.hist_ld;
			const int64_t n = p.hist_off + c0;
			for (int64_t i = tid; i < n; i += SAMPLE_THREADS) {
				const int64_t t = hrow[i];
				if (t >= 0 && t < V) atomicOr(&seen[t >> 5], 1u << (t & 31));
			}
			__syncthreads();
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) {
				const int i = tid + j * SAMPLE_THREADS;
				if (i < V && ((seen[i >> 5] >> (i & 31)) & 1)) v[j] = v[j] < 0.f ? v[j] * p.penalty : v[j] * p.inv_penalty;
			}
		}
		// ---- SuppressTokensLogitsProcessor
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) {
			const int i = tid + j * SAMPLE_THREADS, ic = i < V ? i : V - 1;
			const float t = (p.suppress && p.suppress[ic]) ? -INFINITY : v[j];
			v[j] = i < V ? t : -INFINITY;
		}
		// ---- TypicalLogitsWarper (unified_voice.py:47-75; the reference passes it as a custom processor: after the processors above, before the
		// warpers below): with logp = log_softmax(scores), H = -sum p logp, every token gets the distance |-logp - H|; tokens are taken in ascending
		// distance until their probability mass reaches `mass`, the rest becomes -inf.  No sort: the same 4-pass radix descent as top-p, over the
		// order-preserving key of the DISTANCE with exact fixed-point probability masses -- the first key at which the running mass reaches
		// mass * total is the threshold, everything at or below it stays.  (As with top-p, a row whose running mass ties with the threshold within
		// f32 rounding can keep one token more or less than torch's f32 cumsum does.)
		if (p.typical_mass > 0.f && p.typical_mass < 1.0f) {
			float m0 = -INFINITY;
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) m0 = fmaxf(m0, v[j]);
			m0 = block_max(m0, red, tid);
			float e[SAMPLE_NPT], sum = 0.f;
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) { e[j] = expf(v[j] - m0); sum += e[j]; }
			sum = block_sum(sum, red, tid);
			const float lse = m0 + logf(sum);
			float d[SAMPLE_NPT], hpart = 0.f;
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) {
				const float lp = v[j] - lse, pr = e[j] / sum;
				d[j] = lp;                                        // log p (kept; becomes the distance below)
				if (v[j] != -INFINITY) hpart -= pr * lp;          // nansum: removed tokens contribute (-inf * 0) = nan -> skipped
			}
			const float H = block_sum(hpart, red, tid);
			unsigned long long w[SAMPLE_NPT], tot = 0;
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) {
				w[j] = (unsigned long long)((e[j] / sum) * 1099511627776.0f); tot += w[j];
				d[j] = v[j] != -INFINITY ? fabsf(-d[j] - H) : INFINITY;      // |-logp - H|; removed tokens sort last (inf), as abs(inf - H) does
			}
			if (tid < 256) hist64[tid] = 0;
			__syncthreads();
			atomicAdd(&hist64[0], tot);
			__syncthreads();
			const unsigned long long total = hist64[0];
			// first position (ascending distance) whose cumulative mass is NOT < mass  <=>  running total >= ceil(mass * total)  <=>  > that - 1
			unsigned long long target = (unsigned long long)((double)p.typical_mass * (double)total);
			if (target >= total && total > 0) target = total - 1;
			if (target > 0) target -= 1;
			__syncthreads();
			unsigned prefix = 0, mask = 0;
			unsigned long long below = 0;
			for (int pass = 3; pass >= 0; --pass) {
				const int shift = 8 * pass;
				if (tid < 256) hist64[tid] = 0;
				__syncthreads();
#pragma unroll
				for (int j = 0; j < SAMPLE_NPT; ++j) {
					const unsigned k = fkey(d[j]);
					if (w[j] && (k & mask) == prefix) atomicAdd(&hist64[(k >> shift) & 255], w[j]);
				}
				__syncthreads();
				if (tid < 64) {
					int bin; unsigned long long before;
					pick_bin<unsigned long long, false>(hist64, target - below, lane, bin, before);
					if (tid == 0) { s_bin = bin; s_before = before; }
				}
				__syncthreads();
				prefix |= (unsigned)s_bin << shift;
				mask |= 0xffu << shift;
				below += s_before;
			}
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j)
				if (fkey(d[j]) > prefix) v[j] = -INFINITY;        // sorted_scores > sorted_scores[last_ind]
		}
		// ---- TemperatureLogitsWarper
		if (scale) {
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) v[j] = v[j] * p.inv_t;
		}
		// ---- TopKLogitsWarper: scores < (k-th largest) -> -inf
		if (p.top_k > 0 && p.top_k < V) {
			unsigned prefix = 0, mask = 0, remaining = (unsigned)p.top_k;
			for (int pass = 3; pass >= 0; --pass) {
				const int shift = 8 * pass;
				if (tid < 256) hist32[tid] = 0;
				__syncthreads();
#pragma unroll
				for (int j = 0; j < SAMPLE_NPT; ++j) {
					const int i = tid + j * SAMPLE_THREADS;
					const unsigned k = fkey(v[j]);
					if (i < V && (k & mask) == prefix) atomicAdd(&hist32[(k >> shift) & 255], 1u);
				}
				__syncthreads();
				if (tid < 64) {
					int bin; unsigned before;
					pick_bin<unsigned, true>(hist32, remaining, lane, bin, before);
					if (tid == 0) { s_bin = bin; s_before = before; }
				}
				__syncthreads();
				prefix |= (unsigned)s_bin << shift;
				mask |= 0xffu << shift;
				remaining -= (unsigned)s_before;
			}
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j)
				if (fkey(v[j]) < prefix) v[j] = -INFINITY;
		}
		float m = -INFINITY;
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) m = fmaxf(m, v[j]);
		m = block_max(m, red, tid);
		// ---- TopPLogitsWarper: ascending cumulative softmax <= 1 - top_p -> -inf (the largest score always stays)
		if (p.top_p > 0.f && p.top_p < 1.0f) {
			float e[SAMPLE_NPT], sum = 0.f;
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) { e[j] = expf(v[j] - m); sum += e[j]; }
			sum = block_sum(sum, red, tid);
			unsigned long long w[SAMPLE_NPT], tot = 0;
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j) { w[j] = (unsigned long long)((e[j] / sum) * 1099511627776.0f); tot += w[j]; }
			// block total of the integer masses (exact), to keep the target below it: the largest score is never removed
			if (tid < 256) hist64[tid] = 0;
			__syncthreads();
			atomicAdd(&hist64[0], tot);
			__syncthreads();
			const unsigned long long total = hist64[0];
			unsigned long long target = (unsigned long long)((1.0f - p.top_p) * 1099511627776.0f);
			if (total > 0 && target >= total) target = total - 1;
			__syncthreads();
			unsigned prefix = 0, mask = 0;
			unsigned long long below = 0;
			for (int pass = 3; pass >= 0; --pass) {
				const int shift = 8 * pass;
				if (tid < 256) hist64[tid] = 0;
				__syncthreads();
#pragma unroll
				for (int j = 0; j < SAMPLE_NPT; ++j) {
					const unsigned k = fkey(v[j]);
					if (w[j] && (k & mask) == prefix) atomicAdd(&hist64[(k >> shift) & 255], w[j]);
				}
				__syncthreads();
				if (tid < 64) {
					int bin; unsigned long long before;
					pick_bin<unsigned long long, false>(hist64, target - below, lane, bin, before);
					if (tid == 0) { s_bin = bin; s_before = before; }
				}
				__syncthreads();
				prefix |= (unsigned)s_bin << shift;
				mask |= 0xffu << shift;
				below += s_before;
			}
#pragma unroll
			for (int j = 0; j < SAMPLE_NPT; ++j)
				if (fkey(v[j]) < prefix) v[j] = -INFINITY;
		}
		// ---- softmax, multinomial(1) = argmax(p / q)
		float sum = 0.f;
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) { v[j] = expf(v[j] - m); sum += v[j]; }      // exp(-inf) = 0 for removed / padding lanes
		sum = block_sum(sum, red, tid);
#pragma unroll
		for (int j = 0; j < SAMPLE_NPT; ++j) {
			const int i = tid + j * SAMPLE_THREADS;
			const float r = (v[j] / sum) / qv[j];
			if (i < V && r > best) { best = r; besti = i; }          // strict: the lowest index wins a tie, as in ATen's argmax
		}
	} else {
		const unsigned char* suppress = p.suppress;
		const float inv_t = p.inv_t;
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
		const int64_t live = p.unfinished[b];
		const int64_t nxt = (int64_t)besti * live + p.stop_token * (1 - live);
		p.tok[b] = nxt;
		if (c0 < p.ids_cols) p.ids[(int64_t)b * p.ids_ld + c0] = nxt;
		if (p.history) p.history[(int64_t)b * p.hist_ld + p.hist_off + c0] = nxt;
		p.col[b] = c0 + 1;
		const int64_t still = live * (nxt != p.stop_token ? 1 : 0);
		p.unfinished[b] = still;
		s_next = nxt;
		// stopping criterion without a host round trip per token: the row that finishes last raises a flag the host can poll (the
		// flag may live in pinned host memory; it changes once within a generation, 0 -> the number of tokens sampled when the last row
		// finished, so a late read is merely late and a reader that runs ahead can still tell WHERE the generation ended)
		if (p.live_rows && live != 0 && still == 0) {
			if (__hip_atomic_fetch_add(p.live_rows, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1 && p.all_done)
				__hip_atomic_store(p.all_done, (int)(c0 + 1), __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
	if (p.x_out) {
		// the row the following decode step starts from: mel_embedding[next] + mel_pos_embedding[k + 1], k = c0 + 1 tokens generated
		// (unified_voice.py:212-214) -- what a separate gather launch did before the first layer
		__syncthreads();
		const int64_t nxt = s_next;
		int64_t pi = c0 + 2;
		pi = pi < p.pos_rows ? pi : p.pos_rows - 1;      // guard; the host validates lengths up front
		const float4* e = (const float4*)(p.emb + nxt * p.d);
		const float4* w = (const float4*)(p.pos + pi * p.d);
		float4* o = (float4*)(p.x_out + (int64_t)b * p.d);
		for (int i = tid; i < p.d / 4; i += SAMPLE_THREADS) {
			const float4 a = e[i], c = w[i];
			const float4 v = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
			o[i] = v;
			if (p.x_frag) {   // element (m = b, n = 4i + j) of [m_tile][d/32][lane = (n>>3 & 3) * 16 + (m & 15)][n & 7]: four consecutive n are contiguous
				const int n = 4 * i;
				const int64_t fi = ((((int64_t)(b >> 4) * (p.d / 32) + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (b & 15)) * 8 + (n & 7));
				store4_kind(p.x_frag, fi, v, p.x_frag_f32);      // x_frag_f32: ttk::ElemKind of the copy
			}
		}
	}
}

int launch_sample_step(const ttk_sample_args* a, const float* emb, const float* pos, float* x_out, int d, int pos_rows, void* x_frag, int x_frag_f32, hipStream_t stream, const char* who) {
	TTK_REQUIRE(a && a->scores && a->q && a->unfinished && a->tok && a->ids && a->col, TTK_E_ARG, "%s: null argument", who);
	TTK_REQUIRE(a->B >= 1 && a->V >= 1 && a->ld >= a->V && a->ldq >= a->V, TTK_E_ARG, "%s: bad shape (B %d, V %d)", who, a->B, a->V);
	TTK_REQUIRE(a->temperature > 0.f, TTK_E_ARG, "%s: temperature must be positive", who);
	const bool warp = a->top_k > 0 || (a->top_p > 0.f && a->top_p < 1.0f) || (a->repetition_penalty > 0.f && a->repetition_penalty != 1.0f) ||
					  (a->typical_mass > 0.f && a->typical_mass < 1.0f);
	TTK_REQUIRE(!warp || a->V <= SAMPLE_MAXV, TTK_E_ARG, "%s: top-k / top-p / typical sampling / repetition penalty need V <= %d (V %d)", who, SAMPLE_MAXV, a->V);
	TTK_REQUIRE(a->typical_mass >= 0.f, TTK_E_ARG, "%s: negative typical_mass", who);
	TTK_REQUIRE(a->top_p >= 0.f && a->repetition_penalty >= 0.f, TTK_E_ARG, "%s: negative top_p / repetition_penalty", who);
	TTK_REQUIRE(!(a->repetition_penalty > 0.f && a->repetition_penalty != 1.0f) || a->history, TTK_E_ARG, "%s: the repetition penalty needs the history buffer", who);
	SampleParams p = {};
	p.scores = a->scores; p.ld = a->ld; p.V = a->V; p.q = a->q; p.ldq = a->ldq; p.suppress = a->suppress; p.inv_t = 1.0f / a->temperature;
	p.top_k = a->top_k; p.top_p = a->top_p; p.typical_mass = a->typical_mass;
	p.penalty = a->repetition_penalty > 0.f ? a->repetition_penalty : 1.0f; p.inv_penalty = 1.0f / p.penalty;
	p.stop_token = a->stop_token; p.unfinished = a->unfinished; p.tok = a->tok; p.ids = a->ids; p.ids_ld = a->ids_ld; p.ids_cols = a->ids_cols;
	p.col = a->col; p.history = a->history; p.hist_ld = a->hist_ld; p.hist_off = a->hist_off; p.live_rows = a->live_rows; p.all_done = a->all_done;
	p.emb = emb; p.pos = pos; p.x_out = x_out; p.d = d; p.pos_rows = pos_rows; p.x_frag = x_frag; p.x_frag_f32 = x_frag_f32;
	hipLaunchKernelGGL(k_sample_step, dim3(a->B), dim3(SAMPLE_THREADS), 0, stream, p);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

}  // namespace ttk

extern "C" int ttk_sample_step(const float* scores, int64_t ld, int B, int V, const float* q, int64_t ldq, const unsigned char* suppress, float temperature,
		int64_t stop_token, int64_t* unfinished, int64_t* tok, int64_t* ids, int64_t ids_ld, int64_t ids_cols, int64_t* col, int64_t* history, int64_t hist_ld,
		int64_t hist_off, int* live_rows, int* all_done, void* stream) {
	ttk_sample_args a = {};
	a.scores = scores; a.ld = ld; a.B = B; a.V = V; a.q = q; a.ldq = ldq; a.suppress = suppress; a.temperature = temperature; a.top_k = 0; a.top_p = 1.0f;
	a.repetition_penalty = 1.0f; a.stop_token = stop_token; a.unfinished = unfinished; a.tok = tok; a.ids = ids; a.ids_ld = ids_ld; a.ids_cols = ids_cols;
	a.col = col; a.history = history; a.hist_ld = hist_ld; a.hist_off = hist_off; a.live_rows = live_rows; a.all_done = all_done;
	return ttk::launch_sample_step(&a, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, (hipStream_t)stream, "ttk_sample_step");
}

extern "C" int ttk_sample_step_warped(const ttk_sample_args* a, void* stream) {
	return ttk::launch_sample_step(a, nullptr, nullptr, nullptr, 0, 0, nullptr, 0, (hipStream_t)stream, "ttk_sample_step_warped");
}

namespace ttk {
__global__ void k_exponential_like_torch(float* out, int64_t numel, RngArgs a, int64_t draw) {
	const int64_t li = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (li < numel) out[li] = torch_exponential_at(a, draw, li);
}
}  // namespace ttk

extern "C" int ttk_exponential_like_torch(float* out, int64_t numel, int64_t seed, int64_t offset0, int64_t threads, int64_t step, int64_t draw, void* stream) {
	using namespace ttk;
	TTK_REQUIRE(out && numel >= 1 && threads >= 1, TTK_E_ARG, "ttk_exponential_like_torch: bad argument");
	RngArgs a = {seed, offset0, threads, step, 0, 0};
	hipLaunchKernelGGL(k_exponential_like_torch, dim3((unsigned)((numel + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, numel, a, draw);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

extern "C" int ttk_graph_launch(void* graph_exec, void* stream) {
	using namespace ttk;
	TTK_REQUIRE(graph_exec, TTK_E_ARG, "ttk_graph_launch: null graph");
	TTK_HIP(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
	return TTK_OK;
}
