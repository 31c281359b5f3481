// Attention kernels (head_dim 64) for gfx950.
//
// k_attn_fwd: tiled flash-style forward on MFMA, never materialising the [T,T] score matrix that the reference
//   builds (`QKVAttentionLegacy` /root/reference/tortoise_tts/models/arch_utils.py:59-94 einsum + softmax; GPT-2
//   eager/sdpa attention HF:models/gpt2/modeling_gpt2.py:144-226).  Two uses:
//     * DiffusionTTS AttentionBlock: non-causal, T5 relative-position bias added before the softmax
//       (xtransformers.py:148-188; bias depends on clamp(k-q, -64, 64) only, so a 129-entry per-head table).
//     * GPT-2 prefill / latent pass: causal, no bias.
//   Formulation: S^T = K Q^T (keys on MFMA rows, queries on lanes) so each lane owns one query column: the online
//   softmax reduces over registers + 2 cross-lane steps, and P^T in accumulator layout is directly the B operand
//   of O^T = V^T P^T with a permuted key order that the V^T fragment reads match (ds_read_b64_tr_b16 for bf16).
//   4 waves x 32 queries per workgroup, 64-key K/V tiles staged through LDS, f32 softmax/accumulators.
//
// k_attn_decode: single-query KV-cached attention for the AR decode step (HBM-bound: streams the K/V cache once).
#include <stdlib.h>

#include <hip/hip_ext.h>

#include <mutex>
#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

#ifndef TTK_ATTN_XCD_MAP
#define TTK_ATTN_XCD_MAP 1      // 0: strip index = blockIdx.x, as before round 6 (A/B builds)
#endif
constexpr int HD = 64;
constexpr float NEG_BIG = -1e30f;
constexpr float LOG2E = 1.4426950408889634f;

template <typename T> struct VT;   // V^T fragment loader from an LDS tile [64 keys][64 d] with swizzled 16-byte chunks
template <> struct VT<bf16> {
	// lane (i = lane & 15, g = lane >> 4): elements j<4: V[kb + 4g + j][d0 + i], j>=4: V[kb + 16 + 4g + (j-4)][d0 + i]
	static __device__ __forceinline__ bf16x8 load(const char* Vs, int kb, int d0, int lane) {
		const int i = lane & 15, g = lane >> 4;
		const int q = i >> 2, p = i & 3;            // this lane supplies row q, columns 4p..4p+3 of the 4x16 block
		const int col = d0 + 4 * p;
		const int r0 = kb + 4 * g + q, r1 = r0 + 16;
		const int ch = col >> 3, sub = (col & 7) * 2;
		typedef __attribute__((address_space(3))) bf16x4 lds_v4;
		const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(Vs + r0 * 128 + ((ch ^ (r0 & 7)) << 4) + sub));
		const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_v4*)(Vs + r1 * 128 + ((ch ^ (r1 & 7)) << 4) + sub));
		return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
	}
};
template <> struct VT<f16> {      // the bf16 form on the f16 flavour of the transposing read
	static __device__ __forceinline__ f16x8 load(const char* Vs, int kb, int d0, int lane) {
		const int i = lane & 15, g = lane >> 4;
		const int q = i >> 2, p = i & 3;
		const int col = d0 + 4 * p;
		const int r0 = kb + 4 * g + q, r1 = r0 + 16;
		const int ch = col >> 3, sub = (col & 7) * 2;
		typedef short s16x4 __attribute__((ext_vector_type(4)));
		typedef __attribute__((address_space(3))) s16x4 lds_v4;
		union { s16x4 s; f16x4 h; } lo, hi;      // the builtin's f16 flavour is typed on __fp16 vectors; the 16-bit integer one moves the same bits
		lo.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(Vs + r0 * 128 + ((ch ^ (r0 & 7)) << 4) + sub));
		hi.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4*)(Vs + r1 * 128 + ((ch ^ (r1 & 7)) << 4) + sub));
		return f16x8{lo.h[0], lo.h[1], lo.h[2], lo.h[3], hi.h[0], hi.h[1], hi.h[2], hi.h[3]};
	}
};
template <> struct VT<float> {
	static __device__ __forceinline__ f32x8 load(const char* Vs, int kb, int d0, int lane) {
		const int i = lane & 15, g = lane >> 4;
		const int col = d0 + i;
		f32x8 r;
#pragma unroll
		for (int j = 0; j < 8; ++j) {
			const int row = kb + 4 * g + (j & 3) + (j >> 2) * 16;
			r[j] = *(const float*)(Vs + row * 256 + (((col >> 2) ^ (row & 15)) << 4) + (col & 3) * 4);
		}
		return r;
	}
};

// QT = 16-query tiles per wave: 2 (128 queries per workgroup) when that still gives >= 2 waves per SIMD, else 1 (64 per
// workgroup): a lone wave on a SIMD issues a wave64 VALU op every 4 cycles instead of 2 and cannot overlap its MFMAs.
// NWV = waves per workgroup (4 or 8), each owning 16 * QT queries: 8 x 16 = 128-query blocks halve the K / V bytes a CU takes in per query (every
// workgroup streams all keys of its head through LDS; at T = 1088 that is 590 KB per CU and launch with 64-query blocks, against ~60 GB/s a
// CU can pull from L2) and still leave two waves per SIMD to overlap one wave's softmax VALU work with the other's MFMAs.
// NWV = 9 = the BALANCED form (non-causal, QT = 1): gridDim.x workgroups per (batch, head) share its 16-query tiles as evenly as they divide -- a
// workgroup gets `cnt` = 8 or 9 consecutive tiles, one per wave -- chosen by the launcher so that batch x heads x gridDim.x is exactly the number of
// CUs.  At T = 1088 (68 tiles) with the conditioned + conditioning-free pair that is 8 workgroups per head, 256 in all, one per CU, where 64-query
// blocks make 544 workgroups: 2.125 per CU, so 32 CUs run three at once and the launch lasts 27 us for a median workgroup of 21 (stamps of every
// wave: tests/diag/ddim_chain.cpp).  Each tile's arithmetic is the QT = 1 form's, so the output is bit-identical.  K / V tiles are staged by the
// first 512 threads.
template <typename T, bool CAUSAL, bool BIAS, int QT, int NWV = 4>
__global__ __launch_bounds__(64 * NWV, NWV == 4 ? 2 : 1) void k_attn_fwd(AttnParams p) {
	constexpr bool BAL = NWV == 9;
	typedef typename Frag<T>::type FragT;
	constexpr int ES = sizeof(T);
	constexpr int ROWB = HD * ES;             // LDS row bytes (128 bf16 / 256 f32)
	constexpr int NCH = ROWB / 16;            // 16-byte chunks per row (8 / 16)
	constexpr int SWM = NCH - 1;              // swizzle mask on the row index
	constexpr int FCH = 8 * ES / 16;          // chunks per fragment
	constexpr int TILE_CH = 64 * NCH;         // chunks per K (or V) tile
	constexpr int NTH = BAL ? 512 : 64 * NWV; // threads that stage K / V tiles
	constexpr int CPT = TILE_CH / NTH;        // chunks per thread (bf16 / f32: 2 / 4 with 4 waves, 1 / 2 with 8)
	__shared__ __attribute__((aligned(16))) char Ks[64 * ROWB];
	__shared__ __attribute__((aligned(16))) char Vs[64 * ROWB];
	__shared__ float bias_s[132];
	__shared__ float band_s[BIAS ? 288 : 4];   // the bias as a function of key - query over [-144, 144), saturation included: band tiles index it without clamps

	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	// Balanced form: which (query strip, head, batch element) this workgroup serves is remapped so that the G workgroups of ONE (batch, head) run on ONE XCD (round 6).  Workgroups are
	// dealt to the 8 XCDs round-robin by linear id, x fastest: with G = 8 strips the strip index WAS the XCD, i.e. every XCD fetched every head's K and V (278 KB at T = 1088) for one
	// strip each -- 84 % of the key loop's reads missed the L2 (TCC hit 16 %, 618 cycles per request, profiles/r06_pmc_kloop.txt).  With a head's strips on one XCD its K / V tiles
	// come from the Infinity Cache once per XCD and from that XCD's L2 seven times.  Same work per workgroup, same arithmetic: bit-identical output.
	int bx = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
	if (BAL && TTK_ATTN_XCD_MAP && ((gridDim.y * gridDim.z) & 7) == 0) {
		const int L = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, G = gridDim.x;
		const int xcd = L & 7, sidx = L >> 3, bh = (sidx / G) * 8 + xcd;
		bx = sidx - (sidx / G) * G; h = bh % (int)gridDim.y; b = bh / (int)gridDim.y;
	}
	constexpr int QW = 16 * QT, QB = NWV * QW;        // queries per wave / per workgroup
	const int li = lane & 15, g = lane >> 4;
	TTK_WSTAMP(p.stamps, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 0);
	// ragged batch: sequence b holds TL valid rows in its slot of p.T rows.  Every bound below is the sequence's own (TL); p.T is only the stride.
	const int TL = p.tlen ? p.tlen[b] : p.T;
	int q0 = bx * QB + wave * QW;             // first query row of this wave
	int cnt = NWV;                                    // BAL: 16-query tiles (= working waves) of this workgroup
	if (BAL) {
		const int tiles = (TL + 15) / 16, G = gridDim.x, per = tiles / G, extra = tiles - per * G;
		cnt = per + (bx < extra ? 1 : 0);
		q0 = 16 * (bx * per + min(bx, extra) + wave);
		if (cnt == 0) return;
	} else if (bx * QB >= TL) return;    // a query block of padding rows (uniform for the workgroup: before any barrier)
	const T* base = (const T*)p.qkv + (int64_t)b * p.T * p.ld;
	const int qc = p.q_off + h * p.head_stride, kc = p.k_off + h * p.head_stride, vc = p.v_off + h * p.head_stride;

	__shared__ unsigned pf_sink[64 * NWV];
	if (p.pf) {   // the following projection's weights into L2; workgroups are numbered x-fastest over the (query block, head, batch) grid
		const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
		l2_touch_for_next(p.pf, p.pf_bytes, p.pf_taps, __builtin_amdgcn_readfirstlane(lds_byte_addr(pf_sink) + (threadIdx.x >> 6) * 256), lin, gridDim.x * gridDim.y * gridDim.z, threadIdx.x, 64 * NWV);
	}
	if (BIAS) {
		if (tid < 129) bias_s[tid] = p.bias[h * 129 + tid] * LOG2E;   // scores live in the log2 domain (exp2 = one v_exp_f32)
		for (int i = tid; i < 288; i += 64 * NWV) { const int rel = i - 144; band_s[i] = p.bias[h * 129 + (rel < -64 ? -64 : (rel > 64 ? 64 : rel)) + 64] * LOG2E; }
	}

	// Q fragments (B operand of S^T = K Q^T): lane holds Q[q][32ks + 8g .. +8], scaled.  Four named values (an indexed
	// array of vectors ends up in scratch memory).
	auto load_q = [&](int qt, int ks) -> FragT {
		int row = q0 + 16 * qt + li;
		row = row < TL ? row : TL - 1;
		const FragT f = *(const FragT*)(base + (int64_t)row * p.ld + qc + 32 * ks + 8 * g);
		FragT o;
#pragma unroll
		for (int j = 0; j < 8; ++j) o[j] = cvt<T>((float)f[j] * p.scale);
		return o;
	};
	const FragT q00 = load_q(0, 0), q01 = load_q(0, 1), q10 = load_q(QT - 1, 0), q11 = load_q(QT - 1, 1);

	f32x4 o[QT][4];
	float m_run[QT], l_run[QT];
#pragma unroll
	for (int qt = 0; qt < QT; ++qt) {
		m_run[qt] = NEG_BIG; l_run[qt] = 0.f;
#pragma unroll
		for (int dt = 0; dt < 4; ++dt) o[qt][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
	}

	int nkt = (TL + 63) / 64;
	if (CAUSAL) { const int last_q = min(bx * QB + QB - 1, TL - 1); nkt = min(nkt, last_q / 64 + 1); }

	// staging registers as named scalars (indexed arrays captured by the lambdas were placed in scratch memory)
	uint4 rk0, rk1, rk2, rk3, rv0, rv1, rv2, rv3;
	rk1 = rv1 = rk2 = rk3 = rv2 = rv3 = make_uint4(0, 0, 0, 0);
	// a thread's chunk of a tile: its address inside tile 0 once (row * ld is a 64-bit multiply: three quarter-rate instructions), then one wave-uniform
	// offset per tile; only a last, partly valid tile takes the per-row clamp
	auto thr_ptr = [&](int i) -> const T* { const int id = tid + NTH * i; return base + (int64_t)(id / NCH) * p.ld + (id % NCH) * (16 / ES); };
	const T* const sp0 = thr_ptr(0);
	const T* const sp1 = thr_ptr(CPT > 1 ? 1 : 0);
	const T* const sp2 = thr_ptr(CPT > 2 ? 2 : 0);
	const T* const sp3 = thr_ptr(CPT > 2 ? 3 : 0);
	const int64_t tile_step = (int64_t)64 * p.ld;
	auto ld1 = [&](int kt, int i, uint4& k, uint4& v) {
		const T* src;
		if (kt * 64 + 64 <= TL) {
			src = (i == 0 ? sp0 : (i == 1 ? sp1 : (i == 2 ? sp2 : sp3))) + kt * tile_step;
		} else {
			const int id = tid + NTH * i, row = id / NCH, c = id % NCH;
			int key = kt * 64 + row;
			key = key < TL ? key : TL - 1;
			src = base + (int64_t)key * p.ld + c * (16 / ES);
		}
		k = *(const uint4*)(src + kc);
		v = *(const uint4*)(src + vc);
	};
	auto st1 = [&](int i, const uint4& k, const uint4& v) {
		const int id = tid + NTH * i, row = id / NCH, c = id % NCH;
		const int off = row * ROWB + ((c ^ (row & SWM)) << 4);
		*(uint4*)(Ks + off) = k;
		*(uint4*)(Vs + off) = v;
	};
	auto load_kv = [&](int kt) {
		if (BAL && tid >= NTH) return;
		ld1(kt, 0, rk0, rv0);
		if (CPT > 1) ld1(kt, 1, rk1, rv1);
		if (CPT > 2) { ld1(kt, 2, rk2, rv2); ld1(kt, 3, rk3, rv3); }
	};
	auto store_kv = [&]() {
		if (BAL && tid >= NTH) return;
		st1(0, rk0, rv0);
		if (CPT > 1) st1(1, rk1, rv1);
		if (CPT > 2) { st1(2, rk2, rv2); st1(3, rk3, rv3); }
	};

	load_kv(0);
#ifdef TTK_CLOCK_STAMPS
	const unsigned long long clk0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef TTK_ATTN_PHASES      // diagnostic builds (tests/diag/ddim_chain.cpp): shader cycles of this wave summed per phase of a key tile -- barriers + staging / K reads + QK^T / softmax / V reads + PV
	unsigned long long ph_t_ = 0; unsigned ph_sync_ = 0, ph_qk_ = 0, ph_sm_ = 0, ph_pv_ = 0;
#define TTK_PH_MARK(acc_, dep_) do { asm volatile("s_nop 0" :: "v"(dep_)); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += (unsigned)(n_ - ph_t_); ph_t_ = n_; } while (0)
#else
#define TTK_PH_MARK(acc_, dep_) do {} while (0)
#endif
	for (int kt = 0; kt < nkt; ++kt) {
#ifdef TTK_ATTN_PHASES
		ph_t_ = __builtin_amdgcn_s_memtime();
#endif
		// (TTK_DIAG_ATTN, diagnostic builds of tests/diag/ddim_chain.cpp only, results wrong on purpose: 1 = no workgroup barriers in the key loop, 2 = K / V staged once and
		// never again: profiles/r05_ddim_chain_attn_ablation.log)
#if !(defined(TTK_DIAG_ATTN) && (TTK_DIAG_ATTN & 1))
		__syncthreads();            // previous tile's readers are done
#endif
#if defined(TTK_DIAG_ATTN) && (TTK_DIAG_ATTN & 2)
		if (kt == 0)
#endif
		store_kv();
#if !(defined(TTK_DIAG_ATTN) && (TTK_DIAG_ATTN & 1))
		__syncthreads();
#endif
#if !(defined(TTK_DIAG_ATTN) && (TTK_DIAG_ATTN & 2))
		load_kv(kt + 1 < nkt ? kt + 1 : kt);   // unconditional (the last tile is re-read): keeps the staging registers out of scratch
#endif

		const int k0 = kt * 64;
		const bool wave_active = (!CAUSAL || k0 <= q0 + QW - 1) && (!BAL || wave < cnt);
		TTK_PH_MARK(ph_sync_, k0);
		if (wave_active) {
			// ---- S^T tile: 4 key sub-tiles x 2 query tiles
			f32x4 s[QT][4];
#pragma unroll
			for (int qt = 0; qt < QT; ++qt)
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) s[qt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
				for (int nt = 0; nt < 4; ++nt) {
					union { FragT v; uint4 q[FCH]; } kf;
					const int row = 16 * nt + li;
					const int c0 = (32 * ks + 8 * g) * ES / 16;
#pragma unroll
					for (int f = 0; f < FCH; ++f) kf.q[f] = *(const uint4*)(Ks + row * ROWB + (((c0 + f) ^ (row & SWM)) << 4));
					s[0][nt] = mma16<T>(kf.v, ks == 0 ? q00 : q01, s[0][nt]);
					if (QT > 1) s[QT - 1][nt] = mma16<T>(kf.v, ks == 0 ? q10 : q11, s[QT - 1][nt]);
				}
			}
			TTK_PH_MARK(ph_qk_, s[0][3][3]);
			// ---- bias, masks, online softmax in the log2 domain (lane owns query column li of each q tile; keys 16nt + 4g + r).
			// Wave-uniform fast path: a tile with no sequence edge, no causal diagonal and every |key - q| >= 64 (the T5 bucket is
			// saturated there) needs one fma per score; only the ~3 tiles around the diagonal take the per-element path.
			const bool edge = k0 + 64 > TL;
			const bool diag = CAUSAL && (k0 + 63 > q0);
#pragma unroll
			for (int qt = 0; qt < QT; ++qt) {
				const int qi = q0 + 16 * qt + li;
				// the band test and the saturated bias per 16-QUERY TILE, not per wave: a tile then takes the same path (and rounds the same way)
				// whether its wave owns one tile or two -- a sequence gives the same bits in a batch that is large enough for 128-query blocks
				const int qa = q0 + 16 * qt;
				const bool near_band = BIAS && (k0 - (qa + 15) < 64) && (qa - (k0 + 63) < 64);
				const float cbias = BIAS ? (k0 > qa ? bias_s[128] : bias_s[0]) : 0.f;
				const bool slow = edge || diag || near_band;
				float tmax = NEG_BIG;
				if (BIAS && near_band && !edge && !diag) {
					// a tile inside the bias band with nothing to mask (3 - 4 of the 17 key tiles at T = 1088): the same values as the general path below --
					// bias from the table, one fma -- without its per-element clamps and compares (~80 of its ~110 VALU instructions per tile; the SQ
					// counters put this kernel's critical SIMD 60 % in VALU issue, profiles/r03_pmc_attn.txt): key - query indexes the saturated table
					// directly, the 16 offsets of a lane are immediates
					const float* bt = band_s + (k0 - qi + 4 * g + 144);
#pragma unroll
					for (int nt = 0; nt < 4; ++nt)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const float v = fmaf(s[qt][nt][r], LOG2E, bt[16 * nt + r]);
							s[qt][nt][r] = v;
							tmax = fmaxf(tmax, v);
						}
				} else if (slow) {   // per-element bias lookup and masks; scores become log2-domain values in place
#pragma unroll
					for (int nt = 0; nt < 4; ++nt)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const int key = k0 + 16 * nt + 4 * g + r;
							float bv = 0.f;
							if (BIAS) { int rel = key - qi; rel = rel < -64 ? -64 : (rel > 64 ? 64 : rel); bv = bias_s[rel + 64]; }
							float v = fmaf(s[qt][nt][r], LOG2E, bv);
							if (key >= TL || (CAUSAL && key > qi)) v = NEG_BIG;
							s[qt][nt][r] = v;
							tmax = fmaxf(tmax, v);
						}
				} else {      // one max over the raw scores; the affine map is folded into the exp2 argument below
#pragma unroll
					for (int nt = 0; nt < 4; ++nt)
#pragma unroll
						for (int r = 0; r < 4; ++r) tmax = fmaxf(tmax, s[qt][nt][r]);
					tmax = fmaf(tmax, LOG2E, cbias);
				}
				tmax = fold32_max(fold16_max(tmax));      // (4 VALU instructions; two __shfl_xor cost 22 and two LDS round trips)
				// deferred rescale (cdna_hip_programming.md T13): keep the running max unless this tile exceeds it by more than 2^8;
				// P then stays <= 256, exact enough in bf16 (same relative precision) with f32 accumulation.  Everything at the old
				// scale (O, l) is multiplied exactly once, before any P of this tile exists.
				if (__any(tmax > m_run[qt] + 8.0f)) {
					const float m_new = fmaxf(m_run[qt], tmax);
					const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
					m_run[qt] = m_new;
					l_run[qt] *= alpha;
#pragma unroll
					for (int dt = 0; dt < 4; ++dt) o[qt][dt] *= alpha;
				}
				const float mq = m_run[qt];
				float psum = 0.f;
				// (v_pk_fma_f32 / v_pk_add_f32 for the arguments and the row sum -- 8 fewer VALU instructions per tile -- measured SLOWER, 25.1 against 24.2 us per
				// launch on one box: packed f32 operations cost more than the two scalar ones they replace beside MFMAs, MI355X_MICROARCH.md constants table)
				if (slow) {
#pragma unroll
					for (int nt = 0; nt < 4; ++nt)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const float pv = __builtin_amdgcn_exp2f(s[qt][nt][r] - mq);
							s[qt][nt][r] = pv;
							psum += pv;
						}
				} else {
					const float cm = cbias - mq;
#pragma unroll
					for (int nt = 0; nt < 4; ++nt)
#pragma unroll
						for (int r = 0; r < 4; ++r) {
							const float pv = __builtin_amdgcn_exp2f(fmaf(s[qt][nt][r], LOG2E, cm));
							s[qt][nt][r] = pv;
							psum += pv;
						}
				}
				l_run[qt] += psum;
			}
			TTK_PH_MARK(ph_sm_, l_run[0]);
			// ---- O^T += V^T P^T
#pragma unroll
			for (int sidx = 0; sidx < 2; ++sidx) {
				FragT pf[QT];
#pragma unroll
				for (int qt = 0; qt < QT; ++qt)
#pragma unroll
					for (int j = 0; j < 8; ++j) pf[qt][j] = cvt<T>(s[qt][2 * sidx + (j >> 2)][j & 3]);
#pragma unroll
				for (int dt = 0; dt < 4; ++dt) {
					const FragT vf = VT<T>::load(Vs, 32 * sidx, 16 * dt, lane);
#pragma unroll
					for (int qt = 0; qt < QT; ++qt) o[qt][dt] = mma16<T>(vf, pf[qt], o[qt][dt]);
				}
			}
			TTK_PH_MARK(ph_pv_, o[0][3][3]);
		}
	}
#ifdef TTK_ATTN_PHASES
	if (p.stamps && lane == 0) {
		unsigned long long* st_ = p.stamps + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8;
		st_[1] = ((unsigned long long)ph_sync_ << 32) | ph_qk_; st_[2] = ((unsigned long long)ph_sm_ << 32) | ph_pv_;
	}
#endif

#ifdef TTK_CLOCK_STAMPS
	if (p.stamps && lane == 0) {      // shader cycles and 100 MHz ticks of this wave's key loop (diagnostic build: MI355X_MICROARCH.md, DVFS give-back item 6)
		unsigned long long* st_ = p.stamps + ((size_t)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8;
		st_[6] = ((__builtin_amdgcn_s_memtime() - clk0_) << 32) | ((__builtin_amdgcn_s_memrealtime() - rt0_) & 0xffffffffull);
	}
#endif
	// ---- normalise and store: lane holds d = 16dt + 4g + r of query column li
#pragma unroll
	for (int qt = 0; qt < QT; ++qt) {
		float l = l_run[qt];
		l = fold32_add(fold16_add(l));
		const float inv = 1.0f / l;
		const int qi = q0 + 16 * qt + li;
		if (qi < TL && (!BAL || wave < cnt)) {
			if (p.out_f8) {      // operand of an fp8 projection GEMM: four consecutive head dims = one 32-bit store
				unsigned char* dst = (unsigned char*)p.out + ((int64_t)b * p.T + qi) * p.ldo + h * HD;
#pragma unroll
				for (int dt = 0; dt < 4; ++dt)
					*(unsigned*)(dst + 16 * dt + 4 * g) = pack4_fp8(o[qt][dt][0] * inv, o[qt][dt][1] * inv, o[qt][dt][2] * inv, o[qt][dt][3] * inv);
			} else {
				T* dst = (T*)p.out + ((int64_t)b * p.T + qi) * p.ldo + h * HD;
#pragma unroll
				for (int dt = 0; dt < 4; ++dt)
#pragma unroll
					for (int r = 0; r < 4; ++r) dst[16 * dt + 4 * g + r] = cvt<T>(o[qt][dt][r] * inv);
			}
		}
	}
	TTK_WSTAMPD(p.stamps, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 4, o[0][0][0]);
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	TTK_WSTAMP(p.stamps, (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x, 5);
#endif
}

template <typename T>
static void launch_attn_fwd_t(const AttnParams& p, hipStream_t s) {
	static const int force_qt = [] { const char* e = getenv("TTK_ATTN_QT"); return e ? atoi(e) : 0; }();   // tuning knob
	const bool big = force_qt ? force_qt == 2 : (int64_t)((p.T + 127) / 128) * p.H * p.nb >= 512;   // enough 128-query blocks for 2 waves per SIMD
	// 128-query blocks as 8 waves x 16 queries once they cover most of the chip (TTK_ATTN_W8=0: off); non-causal only: a causal block of 8 waves
	// would idle its upper waves on the diagonal tiles
	// (measured at T = 1088: 141.3 vs 139.9 ms per DDIM loop -- 288 such blocks on 256 CUs leave 32 CUs with two, the same quantisation that
	// costs the 64-query form 544 blocks on 768 slots; kept as a knob, off by default)
	static const int w8 = [] { const char* e = getenv("TTK_ATTN_W8"); return e ? atoi(e) : 0; }();
	if (w8 && !big && !force_qt && !p.causal && (int64_t)((p.T + 127) / 128) * p.H * p.nb >= 192) {
		dim3 grid((p.T + 127) / 128, p.H, p.nb);
		if (p.bias) hipLaunchKernelGGL((k_attn_fwd<T, false, true, 1, 8>), grid, dim3(512), 0, s, p);
		else hipLaunchKernelGGL((k_attn_fwd<T, false, false, 1, 8>), grid, dim3(512), 0, s, p);
		return;
	}
	// balanced form: batch x heads x G workgroups == 256 (the CUs), each with 8..9 tiles
	static const int bal = [] { const char* e = getenv("TTK_ATTN_BAL"); return e ? atoi(e) : 1; }();
	if (bal && !force_qt && !p.causal && !p.tlen && p.nb * p.H <= 256 && 256 % (p.nb * p.H) == 0) {
		const int G = 256 / (p.nb * p.H), tiles = (p.T + 15) / 16;
		if ((tiles + G - 1) / G <= 9 && tiles / G >= 6) {
			dim3 grid(G, p.H, p.nb);
			if (p.bias) hipLaunchKernelGGL((k_attn_fwd<T, false, true, 1, 9>), grid, dim3(576), 0, s, p);
			else hipLaunchKernelGGL((k_attn_fwd<T, false, false, 1, 9>), grid, dim3(576), 0, s, p);
			return;
		}
	}
	if (big) {
		dim3 grid((p.T + 127) / 128, p.H, p.nb);
		if (p.causal) hipLaunchKernelGGL((k_attn_fwd<T, true, false, 2>), grid, dim3(256), 0, s, p);
		else if (p.bias) hipLaunchKernelGGL((k_attn_fwd<T, false, true, 2>), grid, dim3(256), 0, s, p);
		else hipLaunchKernelGGL((k_attn_fwd<T, false, false, 2>), grid, dim3(256), 0, s, p);
	} else {
		dim3 grid((p.T + 63) / 64, p.H, p.nb);
		if (p.causal) hipLaunchKernelGGL((k_attn_fwd<T, true, false, 1>), grid, dim3(256), 0, s, p);
		else if (p.bias) hipLaunchKernelGGL((k_attn_fwd<T, false, true, 1>), grid, dim3(256), 0, s, p);
		else hipLaunchKernelGGL((k_attn_fwd<T, false, false, 1>), grid, dim3(256), 0, s, p);
	}
}
void launch_attn_fwd(int dt, const AttnParams& p, hipStream_t s) {
	ProfScope prof(PROF_ATTN_FWD, 4.0 * p.nb * p.H * (double)p.T * p.T * HD * (p.causal ? 0.5 : 1.0), s);
	if (dt == DT_BF16) launch_attn_fwd_t<bf16>(p, s);
	else if (dt == DT_F16) launch_attn_fwd_t<f16>(p, s);
	else launch_attn_fwd_t<float>(p, s);
}

// ------------------------------------------------------------------------------------------------ decode
// grid (H, B), NW waves.  Lane (slot = lane>>3, dg = lane&7) owns 8 head dims of one key; a wave instruction reads 8 consecutive
// cache rows = 1 KiB (bf16).  Key groups (8 keys) are dealt round-robin to the waves and a wave requests UN groups of K and V at
// once, so for up to NW*UN*8 keys (384 with the values below: every position of the benchmark's 64-token prompt + 250 mel tokens)
// the whole cache slice of this (b, h) is in flight after ONE dependent step (the position scalar), instead of one round trip
// per 128 keys.  Online softmax per (wave, slot), partials merged through LDS in two levels.
// Algorithmic bytes: 2 * (pos+1) * 64 * sizeof(T) per (b, h).
// where output element (candidate b, head h, dim e) goes: row-major [B][H*64], or the A-fragment order the skinny GEMV reads
// ([m_tile][k_step][lane = k_group * 16 + row][8]), in which a wave instruction of the projection reads 1 KiB contiguous
__device__ __forceinline__ int64_t decode_out_index(const AttnDecodeParams& p, int b, int h, int e) {
	if (!p.out_frag) return ((int64_t)b * p.H + h) * HD + e;
	const int n = h * HD + e;
	return ((((int64_t)(b >> 4) * (p.H * HD / 32) + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (b & 15)) * 8 + (n & 7));
}

#if defined(TTK_STAMPS) && TTK_STAMPS == 2   // tests/diag/ar_chain.cpp: every wave stamps, [workgroup][wave (16 slots)][8]; slot 7 = XCC id
#define TTK_ASTAMP(i) do { if (p.stamps && (threadIdx.x & 63) == 0) { unsigned long long* st_ = p.stamps + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8; \
	st_[(i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0) st_[7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); } } while (0)
#define TTK_ASTAMPD(i, dep) do { if (p.stamps) { unsigned tmp_; unsigned long long t_; \
	asm volatile("s_nop 7\n\tv_readfirstlane_b32 %0, %2\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tmp_), "=s"(t_) : "v"(dep) : "memory"); \
	if ((threadIdx.x & 63) == 0) p.stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = t_; } } while (0)
#else
#define TTK_ASTAMP(i) do {} while (0)
#define TTK_ASTAMPD(i, dep) do {} while (0)
#endif

// The position line (round 5).  Every K / V address of the decode attention depends on the cache length, which lives in device memory (one captured step serves every
// token), behind a pointer that arrives with the kernel arguments: two DEPENDENT round trips -- arguments, then position -- before the first key is asked for (1.52 us from a
// wave's first instruction to its requests, profiles/r04_ar_chain.log; the GEMV launches, with one round trip, need 0.68).  The words of up to 8 handles therefore sit in ONE
// 64-byte line of this code object: its address is known at link time, so lane i asks for word i of the line with the wave's first instructions, beside the argument loads, and the
// slot index (an argument) only selects the lane to read when both have arrived.  Writers (prefill, the head launch's bump) reach the words through the ordinary pointer.
__device__ __attribute__((aligned(64))) int g_pos_line[16];
static unsigned g_pos_slots_used = 0;      // slot mask of the ONE device this process drives (one process per GPU: a second device would have its own copy of g_pos_line but share this mask)
static std::mutex g_pos_slots_mutex;
int attn_pos_slot_acquire(int** words_out) {
	std::lock_guard<std::mutex> lock(g_pos_slots_mutex);
	void* base = nullptr;
	if (hipGetSymbolAddress(&base, HIP_SYMBOL(g_pos_line)) != hipSuccess) { (void)hipGetLastError(); return -1; }
	for (int s = 0; s < 8; ++s)
		if (!(g_pos_slots_used & (1u << s))) { g_pos_slots_used |= 1u << s; *words_out = (int*)base + 2 * s; return s; }
	return -1;
}
void attn_pos_slot_release(int slot) {
	std::lock_guard<std::mutex> lock(g_pos_slots_mutex);
	if (slot >= 0 && slot < 8) g_pos_slots_used &= ~(1u << slot);
}

template <typename T, int NW, int UN, bool ROWS = false, bool SLOT = false>
__global__ __launch_bounds__(64 * NW) void k_attn_decode(AttnDecodeParams p) {
	typedef typename Frag<T>::type FragT;
	constexpr int NP = NW * 8;           // (wave, slot) partial softmaxes
	int line_word = 0;
	if (SLOT) line_word = __builtin_nontemporal_load(g_pos_line + (threadIdx.x & 15));      // (link-time address: nothing to wait for)
	TTK_PIN_ARGS(TTK_S(p.qbuf), TTK_S(p.kcache), TTK_S(p.vcache), TTK_S(p.d_pos), TTK_S(p.H), TTK_S(p.max_ctx), TTK_S(p.out), TTK_S(p.out_frag),
				 TTK_S(p.row_info), TTK_S(p.shared_rows));
	const int h = blockIdx.x, b = blockIdx.y;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int slot = lane >> 3, dg = lane & 7;
	TTK_ASTAMP(0);
#ifdef TTK_ABL      // diagnostic builds only (tests/diag/ar_ablate.sh): 256 = the whole kernel, 128 = the K / V loads
	if (TTK_ABL & 256) return;
#endif
	// ROWS: several text lines in one batch, prefixes right-aligned.  Candidate b's keys are cache rows [start, d_pos + 1) -- numbered from 0 here,
	// so they are dealt to the waves exactly as in a batch of its own -- and its shared prefix rows live in the slice of its line's first candidate
	int start = 0, grp = 0;
	if (ROWS) { const int2 ri = p.row_info[b]; start = ri.x; grp = ri.y; }
	// {valid cache rows, rows of the shared prefix} as ONE 8-byte request.  Written as two reads, the second under `p.shared_rows`, hipcc issued two DEPENDENT
	// loads -- position, wait, prefix length, wait -- in front of every K / V address: 1.76 us from a wave's first instruction to its requests
	// (profiles/r03_ar_chain_lean.log).  Both words sit in device memory, not in the kernel arguments: a captured token step is replayed for later
	// calls with other prefix lengths.
	int2 dp;
	if (SLOT) { const int w = 2 * (p.pos_slot_p1 - 1); dp.x = __builtin_amdgcn_readlane(line_word, w); dp.y = __builtin_amdgcn_readlane(line_word, w + 1); }
	else dp = *(const int2*)p.d_pos;
	const int n = min(dp.x + 1, p.max_ctx) - start;
	// rows [0, shared) are read from the line's first candidate
	const int sh0 = p.shared_rows ? dp.y : 0;
	const int shared = sh0 > 0 ? sh0 - start : 0;
	const T* Kc = (const T*)p.kcache + (((int64_t)b * p.H + h) * p.max_ctx + start) * HD;
	const T* Vc = (const T*)p.vcache + (((int64_t)b * p.H + h) * p.max_ctx + start) * HD;
	const int64_t to_shared = -(int64_t)(b - grp) * p.H * p.max_ctx * HD;      // element offset from this candidate's slice to its line's first
	const int groups = (n + 7) / 8;
	// The first round's K / V requests leave right behind the position words, BEFORE the query is asked for: placed after the query (whose
	// scaling the compiler hoists in front of the loop, with its wait) they were a third dependent round trip -- arguments, position, query, keys.
	FragT kf[UN], vf[UN];
	auto request = [&](int gb) {
#pragma unroll
		for (int u = 0; u < UN; ++u) {     // unconditional, clamped: all 2*UN requests leave before the first use
			int key = (gb + u * NW) * 8 + slot;
			key = key < n ? key : n - 1;
#ifdef TTK_ABL
			if (TTK_ABL & 128) { kf[u] = FragT{}; vf[u] = FragT{}; continue; }
#endif
			const int64_t off = (int64_t)key * HD + 8 * dg + (key < shared ? to_shared : 0);
			kf[u] = *(const FragT*)(Kc + off);
			vf[u] = *(const FragT*)(Vc + off);
		}
	};
	request(wave);      // waves beyond the last key group re-read the last row (never used)
	__builtin_amdgcn_sched_barrier(0);
	float q[8];
	{
		const float* qp = p.qbuf + ((int64_t)b * p.H + h) * HD + 8 * dg;
#pragma unroll
		for (int j = 0; j < 8; ++j) q[j] = qp[j] * LOG2E;   // log2-domain scores: exp2 is a single v_exp_f32
	}
	TTK_ASTAMPD(1, (float)(n + shared) + q[0]);      // cache length, shared-prefix length and the query are there
	float m = NEG_BIG, l = 0.f, acc[8];
#pragma unroll
	for (int j = 0; j < 8; ++j) acc[j] = 0.f;
	for (int gb = wave; gb < groups;) {
#pragma unroll
		for (int u = 0; u < UN; ++u) {
			const int key = (gb + u * NW) * 8 + slot;
			float sdot = 0.f;
#pragma unroll
			for (int j = 0; j < 8; ++j) sdot += q[j] * (float)kf[u][j];
			sdot = dpp_add<0xB1>(sdot);    // the 8 lanes of a key: quad swaps + row_half_mirror, all DPP
			sdot = dpp_add<0x4E>(sdot);
			sdot = dpp_add<0x141>(sdot);
			if (key < n) {
				const float m_new = fmaxf(m, sdot);
				const float alpha = __builtin_amdgcn_exp2f(m - m_new), pv = __builtin_amdgcn_exp2f(sdot - m_new);
				l = l * alpha + pv;
#pragma unroll
				for (int j = 0; j < 8; ++j) acc[j] = acc[j] * alpha + pv * (float)vf[u][j];
				m = m_new;
			}
		}
		gb += NW * UN;
		if (gb < groups) request(gb);
	}
	TTK_ASTAMPD(2, acc[0] + l);                      // this wave's keys are in and reduced
	// merge the NP partial softmaxes through LDS in two levels.  Level 1 is wave-private: a wave folds its own 8 slots (lane = head dim) as soon as
	// its keys are done -- no workgroup barrier in front of it, so it runs inside the time the wave would otherwise wait for the slowest one -- and
	// publishes ONE partial; after the only barrier 64 threads fold the NW wave partials.  (First version: barrier, 4 waves folding 32 partials
	// each while 12 idled, barrier, 4 more: 36 dependent steps behind two barriers against 8 + NW behind one.)
	__shared__ float sm[NP], sl[NP], sacc[NP][HD + 1];
	__shared__ float sm2[NW], sl2[NW], so2[NW][HD];
	const int ps = wave * 8 + slot;
	if (dg == 0) { sm[ps] = m; sl[ps] = l; }
#pragma unroll
	for (int j = 0; j < 8; ++j) sacc[ps][8 * dg + j] = acc[j];
	__builtin_amdgcn_wave_barrier();      // same wave, LDS is in order: the reads below see the writes above
	{
		float mn = NEG_BIG;
#pragma unroll
		for (int i = 0; i < 8; ++i) mn = fmaxf(mn, sm[wave * 8 + i]);
		float lt = 0.f, ot = 0.f;
#pragma unroll
		for (int i = 0; i < 8; ++i) { const float a = __builtin_amdgcn_exp2f(sm[wave * 8 + i] - mn); lt += sl[wave * 8 + i] * a; ot += sacc[wave * 8 + i][lane] * a; }
		if (lane == 0) { sm2[wave] = mn; sl2[wave] = lt; }
		so2[wave][lane] = ot;
	}
	__syncthreads();
	TTK_ASTAMP(3);
	if (tid < HD) {
		float mn = NEG_BIG;
#pragma unroll
		for (int i = 0; i < NW; ++i) mn = fmaxf(mn, sm2[i]);
		float lt = 0.f, ot = 0.f;
#pragma unroll
		for (int i = 0; i < NW; ++i) { const float a = __builtin_amdgcn_exp2f(sm2[i] - mn); lt += sl2[i] * a; ot += so2[i][tid] * a; }
		((T*)p.out)[decode_out_index(p, b, h, tid)] = cvt<T>(ot / lt);
	}
	TTK_ASTAMP(5);
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	TTK_ASTAMP(6);
#endif
}

template <typename T, int NW, int UN>
static void launch_attn_decode_t(const AttnDecodeParams& p, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
	static const bool no_line = [] { const char* e = getenv("TTK_ATTN_POS_LINE"); return e && atoi(e) == 0; }();      // A/B knob: 0 = always through d_pos
	const bool slot = p.pos_slot_p1 > 0 && !no_line;
	if (p.row_info) { if (slot) hipExtLaunchKernelGGL((k_attn_decode<T, NW, UN, true, true>), dim3(p.H, p.B), dim3(64 * NW), 0, s, ea, eb, 0, p);
					  else hipExtLaunchKernelGGL((k_attn_decode<T, NW, UN, true>), dim3(p.H, p.B), dim3(64 * NW), 0, s, ea, eb, 0, p); }
	else if (slot) hipExtLaunchKernelGGL((k_attn_decode<T, NW, UN, false, true>), dim3(p.H, p.B), dim3(64 * NW), 0, s, ea, eb, 0, p);
	else hipExtLaunchKernelGGL((k_attn_decode<T, NW, UN>), dim3(p.H, p.B), dim3(64 * NW), 0, s, ea, eb, 0, p);
}
void launch_attn_decode(int dt, const AttnDecodeParams& p, hipStream_t s) {
	hipEvent_t ea = nullptr, eb = nullptr;      // kernel start / stop timestamps when profiling (prof_pair)
	if (g_prof_on) prof_pair(PROF_ATTN_DECODE, 2.0 * p.B * p.H * (double)p.ctx_hint * HD * dtype_size(dt), &ea, &eb);
	static const int variant = [] { const char* e = getenv("TTK_ATTN_DECODE"); return e ? atoi(e) : 0; }();   // tuning knob: 0 = default
	if (dt == DT_BF16) {
		// 250-token AR loop, B=16, one box: 4 waves x 4 groups (first version, 128 keys per round trip) 261.9 ms; 8 x 6 257.5 ms;
		// 16 x 3 256.8 ms
		if (variant == 1) launch_attn_decode_t<bf16, 4, 4>(p, s, ea, eb);
		else if (variant == 2) launch_attn_decode_t<bf16, 8, 6>(p, s, ea, eb);
		else launch_attn_decode_t<bf16, 16, 3>(p, s, ea, eb);
	} else if (dt == DT_F16) {
		launch_attn_decode_t<f16, 16, 3>(p, s, ea, eb);
	} else {
		launch_attn_decode_t<float, 8, 6>(p, s, ea, eb);
	}
}

// ------------------------------------------------------------------------------------------------ prefill KV -> cache
template <typename T>
__global__ void k_kv_scatter(const T* qkv, int B, int S, int H, T* kc, T* vc, int max_ctx, int t0) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per 8 elements
	const int d = H * HD;
	const int64_t total = (int64_t)B * S * (d / 8);
	if (idx >= total) return;
	const int c8 = (int)(idx % (d / 8));
	const int64_t row = idx / (d / 8);
	const int b = (int)(row / S), t = (int)(row - (int64_t)b * S);
	const int c = c8 * 8, h = c >> 6, dd = c & 63;
	typedef typename Frag<T>::type FragT;
	const T* src = qkv + row * (3 * d);
	const int64_t dst = (((int64_t)b * H + h) * max_ctx + t0 + t) * HD + dd;      // t0: first cache row (right-aligned prefixes of a line batch)
	*(FragT*)(kc + dst) = *(const FragT*)(src + d + c);
	*(FragT*)(vc + dst) = *(const FragT*)(src + 2 * d + c);
}
void launch_kv_scatter(int dt, const void* qkv, int B, int S, int H, void* kcache, void* vcache, int max_ctx, hipStream_t s, int t0) {
	const int64_t total = (int64_t)B * S * (H * HD / 8);
	const int grid = (int)((total + 255) / 256);
	if (dt == DT_BF16) hipLaunchKernelGGL((k_kv_scatter<bf16>), dim3(grid), dim3(256), 0, s, (const bf16*)qkv, B, S, H, (bf16*)kcache, (bf16*)vcache, max_ctx, t0);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_kv_scatter<f16>), dim3(grid), dim3(256), 0, s, (const f16*)qkv, B, S, H, (f16*)kcache, (f16*)vcache, max_ctx, t0);
	else hipLaunchKernelGGL((k_kv_scatter<float>), dim3(grid), dim3(256), 0, s, (const float*)qkv, B, S, H, (float*)kcache, (float*)vcache, max_ctx, t0);
}

}  // namespace ttk
