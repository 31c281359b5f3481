// Skinny (M <= 64 rows) weight-streaming GEMM for the KV-cached AR decode step -- the HBM-bound half of the path.
//
//   out[M, N] = epilogue( LN?(x)[M, K] * W[N, K]^T + bias )
//
// One workgroup per 16-column n-tile; its waves split K; every wave streams its slice of the weights exactly once,
// straight from HBM into MFMA B-operand registers (no LDS round trip: "GEMV / M <= 16 decode weights" row of
// cdna_hip_programming.md section 5).  Weights are stored in MFMA-fragment order  Wp[n_tile][k_step][lane][8]
// so one wave instruction reads 1 KiB (bf16) of contiguous memory.  The batch rows are the MFMA M dimension
// (16 candidates = one 16x16 tile, MT tiles for bigger shards).
//
// Fused prologues/epilogues (one launch per GPT-2 sub-block, reference ops in parentheses):
//   LN1 + c_attn + bias -> q (pre-scaled) / K,V appended to the cache  (HF:models/gpt2/modeling_gpt2.py:144-226,
//        DynamicCache.update)                                                         SK_QKV
//   c_proj + bias + residual add                                                       SK_RESIDUAL
//   LN2 + c_fc + bias + gelu_new  (HF:activations.py:59-66)                            SK_ACT_T
//   ln_f + final_norm + mel_head  (unified_voice.py:106,239)                           SK_STORE_F32, ln_count = 2
// Algorithmic bytes per launch: N*K*sizeof(T) weight bytes (+ M*K*4 activations from L2).
#include <hip/hip_ext.h>

#include "ttk_common.h"
#include "ttk_kernels.h"
#include "ttk_rng.h"

namespace ttk {

// Diagnostic build only (-DTTK_STAMPS, tests/diag/skinny_stamps.cpp): wave 0 / lane 0 of every workgroup records a 100 MHz
// timestamp per phase.  Expands to nothing in the product build.
// weight-stream cache policy: non-temporal (streamed once per token) unless built with -DTTK_NT=0
#ifndef TTK_NT
#define TTK_NT 1
#endif
#if TTK_NT
#define TTK_WLOAD(p) __builtin_nontemporal_load(p)
#else
#define TTK_WLOAD(p) (*(p))
#endif

// Diagnostic builds only (-DTTK_ABL=<bits>, tests/diag/ar_ablate.sh): leave out one part of the kernel to price it in the real decode loop
// (results are then wrong on purpose).  1 LayerNorm statistics + affine, 2 gamma / beta loads, 4 cross-wave reduction, 8 weight loads,
// 16 activation loads, 32 output stores, 64 the whole kernel.  Expands to nothing in the product build.
#ifndef TTK_ABL
#define TTK_ABL 0
#endif

#if defined(TTK_STAMPS) && TTK_STAMPS == 2   // tests/diag/ar_chain.cpp: every wave stamps, [workgroup][wave (16 slots)][8]; slot 7 = XCC id
#define TTK_STAMP(i) do { if (p.stamps && (threadIdx.x & 63) == 0) { unsigned long long* st_ = p.stamps + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8; \
	st_[(i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0) st_[7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); } } while (0)
// stamp taken only once `dep` (a VGPR value: an accumulator, a loaded word) is really there: the asm reads it, so the wave stalls on the MFMA /
// the load that produces it first -- a bare s_memrealtime has no data dependency and floats above the arithmetic it is meant to follow
#define TTK_STAMPD(i, dep) do { if (p.stamps) { unsigned tmp_; unsigned long long t_; \
	asm volatile("s_nop 7\n\tv_readfirstlane_b32 %0, %2\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tmp_), "=s"(t_) : "v"(dep) : "memory"); \
	if ((threadIdx.x & 63) == 0) p.stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = t_; } } while (0)
#elif defined(TTK_STAMPS)
#define TTK_STAMP(i) do { if (p.stamps && threadIdx.x == 0) p.stamps[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define TTK_STAMPD(i, dep) TTK_STAMP(i)
#else
#define TTK_STAMP(i) do {} while (0)
#define TTK_STAMPD(i, dep) do {} while (0)
#endif

// Weight fragment as it sits in memory: the MFMA operand itself, or (W8, bf16 arithmetic only) 8 fp8-e4m3 bytes that are widened to
// bf16 -- exactly, e4m3 has 3 mantissa bits -- next to their MFMA; the power-of-two tensor scale is applied to the f32 sums.
template <typename T, bool W8> struct WFrag {
	typedef typename Frag<T>::type raw;
	static __device__ __forceinline__ typename Frag<T>::type dec(raw r) { return r; }
};
template <> struct WFrag<bf16, true> {
	typedef unsigned raw __attribute__((ext_vector_type(2)));      // a clang vector: the non-temporal load builtin takes no HIP struct types
	static __device__ __forceinline__ bf16x8 dec(raw r) {
		typedef float f2 __attribute__((ext_vector_type(2)));
		const f2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[0], true);
		const f2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[1], true);
		return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)b[0], (bf16)b[1], (bf16)c[0], (bf16)c[1], (bf16)d[0], (bf16)d[1]};
	}
};

// KC = float4 chunks of a row per lane in the LayerNorm prologue (K <= 256 * KC); LN kernels run <= 8 waves (2 per SIMD,
// 256 VGPRs), plain ones up to 16.
// FOLD (plain A operand only): the LayerNorm in front of this matrix folded into it -- see SkinnyParams.g1.  The waves sum x and x^2 of the
// fragments they feed the MFMA (lane = one row x 8 k), the per-wave row partials go through LDS next to the accumulators, and the
// epilogue finishes (acc - mean * csum[n]) * rstd + bias.  No normalised copy of the rows is ever written, no LayerNorm barrier, no
// gamma / beta traffic: the launch is shaped like the plain output projections.
template <typename T, int MT, bool LN, int KC, bool W8, bool FOLD = false>
__global__ __launch_bounds__(LN ? 512 : 1024) void k_skinny(SkinnyParams p) {
	typedef typename Frag<T>::type FragT;
	typedef WFrag<T, W8> WF;
	typedef typename WF::raw WRaw;
	constexpr int ES = sizeof(T);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
	if (TTK_ABL & 64) return;
	// Plain mode: ksplit workgroups share one 16-column n-tile.  Narrow mode (p.narrow = 4, or 2): that many workgroups share an n-tile, each
	// owning 4 (8) of its columns over the whole of K -- 4x (2x) the workgroups streaming the matrix without any split-K combine.  The MFMA still runs
	// 16 columns wide: lane (g, n) fetches the fragment of column 4*sub + (n & 3), so columns 4..15 of the product are copies that the
	// epilogue ignores (matrix throughput is irrelevant here, the weight stream is the work).  The four workgroups of a tile read the
	// same 128-byte lines, so they are given ids that are equal mod 8: round-robin dispatch then puts them on one XCD and its L2
	// fetches every line once.
	int nt, kslice = 0, sub = 0;
	const int G = p.narrow > 1 ? p.narrow : 1;          // workgroups per 16-column tile (narrow mode: 2 or 4), 16 / G columns each
	const int NC = 16 / G;
	if (p.narrow) {
		const int b = blockIdx.x, ntiles = (p.N + 15) / 16;
		if ((ntiles & 7) == 0) { nt = (b / (8 * G)) * 8 + (b & 7); sub = (b >> 3) % G; }
		else { nt = b / G; sub = b % G; }
	} else {
		nt = blockIdx.x / p.ksplit; kslice = blockIdx.x - nt * p.ksplit;
	}
	const int KS = p.K / 32;
	const int RS = p.K * ES + 16;                       // padded LDS row stride (bytes), LN mode only
	char* a_lds = smem;
	float* red = (float*)(smem + (LN ? 16 * MT * RS : 0));
	float* rstat = red + nw * MT * 64 * 4;              // FOLD: [wave][m_tile][16 rows][sum, sum of squares]

	// Order of the memory requests matters (vmcnt retires in order): first the activation rows this wave normalises, then
	// its whole first batch of weight fragments, so the HBM latency of the weights hides behind the LayerNorm arithmetic.
	constexpr int RP = 2;                                // rows per LayerNorm pass of a wave
	const int nchunk = p.K / 4;
	float4 v[RP][KC];
	auto ln_load = [&](int r0) {
#pragma unroll
		for (int j = 0; j < RP; ++j) {
			const int r = r0 + j * nw;
			const bool live = r < p.M;
#pragma unroll
			for (int i = 0; i < KC; ++i) {   // unconditional loads (clamped address + select): a branch around a load makes hipcc
				const int c = lane + 64 * i;     // fall back to s_waitcnt vmcnt(0), which would also wait for the weight stream
				const int cc = c < nchunk ? c : nchunk - 1;
				const int rr = live ? r : 0;
				const float4 t = (TTK_ABL & 16) ? make_float4(1.f, 2.f, 3.f, 4.f) : *(const float4*)(p.x + (int64_t)rr * p.ldx + 4 * cc);
				v[j][i] = (live && c < nchunk) ? t : make_float4(0, 0, 0, 0);
			}
		}
	};
	// LayerNorm of RP rows at once: per-lane partial (sum, sum of squares) of every row first, then ONE butterfly in which the
	// 2*RP independent chains interleave (a dependent 6-step cross-lane chain costs ~0.3 us; the first version ran four of them
	// back to back per row and the prologue took 4 us).  Variance = E[x^2] - mean^2 in f32 (K <= 2048, |mean| << rms).
	// gamma/beta of the first LayerNorm are requested together with the rows, BEFORE the weight stream: vmcnt retires in order,
	// so a gamma load issued after the weights would make the normalisation wait for the whole HBM stream.
	float4 g0[KC], b0[KC];
	auto ln_load_affine = [&]() {
#pragma unroll
		for (int i = 0; i < KC; ++i) {
			const int c = lane + 64 * i;
			const int cc = c < nchunk ? c : nchunk - 1;
			const float4 tg = (TTK_ABL & 2) ? make_float4(1.f, 1.f, 1.f, 1.f) : *(const float4*)(p.g1 + 4 * cc), tb = (TTK_ABL & 2) ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(p.b1 + 4 * cc);
			g0[i] = c < nchunk ? tg : make_float4(0, 0, 0, 0);
			b0[i] = c < nchunk ? tb : make_float4(0, 0, 0, 0);
		}
	};
	auto ln_finish = [&](int r0) {
		for (int pass = 0; pass < ((TTK_ABL & 1) ? 0 : p.ln_count); ++pass) {
			float4 gg[KC], bb[KC];
#pragma unroll
			for (int i = 0; i < KC; ++i) {
				const int c = lane + 64 * i;
				if (pass == 0) { gg[i] = g0[i]; bb[i] = b0[i]; }
				else {
					gg[i] = c < nchunk ? *(const float4*)(p.g2 + 4 * c) : make_float4(0, 0, 0, 0);
					bb[i] = c < nchunk ? *(const float4*)(p.b2 + 4 * c) : make_float4(0, 0, 0, 0);
				}
			}
			float s1[RP], s2[RP];
#pragma unroll
			for (int j = 0; j < RP; ++j) {
				s1[j] = 0.f; s2[j] = 0.f;
#pragma unroll
				for (int i = 0; i < KC; ++i) {
					s1[j] += v[j][i].x + v[j][i].y + v[j][i].z + v[j][i].w;
					s2[j] += v[j][i].x * v[j][i].x + v[j][i].y * v[j][i].y + v[j][i].z * v[j][i].z + v[j][i].w * v[j][i].w;
				}
			}
#pragma unroll
			for (int j = 0; j < RP; ++j) { s1[j] = wave_sum(s1[j]); s2[j] = wave_sum(s2[j]); }   // independent chains: they interleave
#pragma unroll
			for (int j = 0; j < RP; ++j) {
				const float mean = s1[j] / (float)p.K;
				const float var = fmaxf(s2[j] / (float)p.K - mean * mean, 0.f);
				const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
				for (int i = 0; i < KC; ++i) {   // lanes beyond the row hold zeros and zero gamma/beta: harmless
					v[j][i].x = (v[j][i].x - mean) * rstd * gg[i].x + bb[i].x;
					v[j][i].y = (v[j][i].y - mean) * rstd * gg[i].y + bb[i].y;
					v[j][i].z = (v[j][i].z - mean) * rstd * gg[i].z + bb[i].z;
					v[j][i].w = (v[j][i].w - mean) * rstd * gg[i].w + bb[i].w;
				}
			}
		}
#pragma unroll
		for (int j = 0; j < RP; ++j) {
			const int r = r0 + j * nw;
			if (r >= 16 * MT) continue;
			const bool live = r < p.M;
#pragma unroll
			for (int i = 0; i < KC; ++i) {
				const int c = lane + 64 * i;
				if (c < nchunk) {
					T* dst = (T*)(a_lds + r * RS) + 4 * c;
					dst[0] = cvt<T>(live ? v[j][i].x : 0.f);
					dst[1] = cvt<T>(live ? v[j][i].y : 0.f);
					dst[2] = cvt<T>(live ? v[j][i].z : 0.f);
					dst[3] = cvt<T>(live ? v[j][i].w : 0.f);
					if (p.ln_out && nt == 0 && live) *(float4*)(p.ln_out + (int64_t)r * p.K + 4 * c) = v[j][i];
				}
			}
		}
	};
	TTK_STAMP(0);
	// the cache row the c_attn epilogue appends at: requested now (first in the vmcnt order), not as one more dependent round trip at the end
	int kv_pos = 0;
	if (p.mode == SK_QKV) kv_pos = *p.d_pos;
	// mel_head launch of a decode step: one thread advances the cache length for the next step (nothing in this launch reads it)
	if (p.mode == SK_STORE_F32 && p.d_pos && blockIdx.x == 0 && threadIdx.x == 0) *(int*)p.d_pos += 1;
	// Epilogue role of the first 256 threads: element (row 4*(l2>>4)+r of each m-tile, column l2&15) of the 16-wide output tile.  Its
	// bias and, for the residual modes, the current value of the output are requested now, ahead of everything else, so the epilogue
	// finds them in registers instead of paying one more dependent L2 round trip after the reduction.
	const int l2 = tid & 63, r = tid >> 6;
	const int n = p.narrow ? nt * 16 + NC * sub + (l2 & (NC - 1)) : nt * 16 + (l2 & 15);
	const bool mine = tid < 256 && (!p.narrow || (l2 & 15) < NC) && n < p.N;
	float bias = 0.f, fcs = 0.f, res[MT];
	RngArgs rng = {};           // mel head drawing the multinomial noise: generator state and this thread's draw counters
	int64_t draw[MT];
	{
		const int nn = n < p.N ? n : p.N - 1;
		if (p.bias) bias = p.bias[nn];
		if (FOLD) fcs = p.g1[nn];            // column sum of the folded matrix (cold, like the bias: left to the epilogue it is a dependent trip to HBM)
		const bool noise = p.mode == SK_STORE_F32 && p.qbuf;
		if (noise) rng = *(const RngArgs*)p.slab;
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) {
			res[mt] = 0.f; draw[mt] = 0;
			int m = mt * 16 + 4 * (l2 >> 4) + (r & 3);
			m = m < p.M ? m : p.M - 1;
			if (p.mode == SK_RESIDUAL) res[mt] = p.out_f32[(int64_t)m * p.ldc + nn];
			if (noise) draw[mt] = ((const int64_t*)p.tickets)[m];
		}
	}
	if (LN) { ln_load(wave); ln_load_affine(); }

	// ---- this wave's K slice of the weights; the first PRE fragments are requested now
	const int kw0 = (KS * kslice) / p.ksplit, kw1 = (KS * (kslice + 1)) / p.ksplit;   // this workgroup's k-steps
	const int ks0 = kw0 + ((kw1 - kw0) * wave) / nw, ks1 = kw0 + ((kw1 - kw0) * (wave + 1)) / nw;
	const WRaw* wp = (const WRaw*)p.Wp + ((int64_t)nt * KS) * 64 + (p.narrow ? ((lane & ~15) | (NC * sub + (lane & (NC - 1)))) : lane);
	// Weight fragments requested per batch.  LN mode: 8 x 1 KiB (bf16) per wave.  Plain mode: the A fragments of the same k-steps are requested
	// right beside them (B0 A0 B1 A1 ...), all before the first use -- fetched next to their MFMA instead, every k-step was one more
	// dependent L2 round trip behind a full vmcnt(0) (8 in a row in mlp.c_proj: 4.7 of its 6.9 us) -- so the batch is sized to the register
	// budget of a 1024-thread workgroup (128 VGPRs): <= 64 registers of operands in flight.
	constexpr int OPREGS = (MT + 1) * (ES == 4 ? 8 : 4);                       // A (per m-tile) + B registers of one k-step
	constexpr int PRE = LN ? 8 : (64 / OPREGS >= 8 ? 8 : (64 / OPREGS >= 4 ? 4 : (64 / OPREGS >= 2 ? 2 : 1)));
	WRaw bpre[PRE];
	FragT apre[LN ? 1 : PRE][MT];
	const int npre = min(ks1 - ks0, PRE);
	const int arow = lane & 15, ag = lane >> 4;
	auto load_a_global = [&](int mt, int ks) -> FragT {
		if (TTK_ABL & 16) return FragT{};
		if (p.a_frag) return *(const FragT*)((const T*)p.a + (((int64_t)mt * KS + ks) * 64 + lane) * 8);   // rows >= M hold zeros
		int row = mt * 16 + arow;
		row = row < p.M ? row : p.M - 1;
		return *(const FragT*)((const T*)p.a + (int64_t)row * p.lda + 32 * ks + 8 * ag);
	};
#pragma unroll
	for (int u = 0; u < PRE; ++u) {   // unconditional: slots beyond npre re-read the last fragment (never multiplied)
		const int kk = max(ks0 + (u < npre ? u : npre - 1), kw0);
		if (TTK_ABL & 8) bpre[u] = WRaw{}; else bpre[u] = TTK_WLOAD(wp + (int64_t)kk * 64);
		if (!LN) {
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) apre[u][mt] = load_a_global(mt, kk);
		}
	}

	if (LN) {
		ln_finish(wave);
		TTK_STAMP(1);
		for (int r0 = wave + RP * nw; r0 < 16 * MT; r0 += RP * nw) { ln_load(r0); ln_finish(r0); }
		__syncthreads();
	}
	TTK_STAMP(2);

	f32x4 acc[MT];
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
	auto load_a_lds = [&](int mt, int ks) -> FragT {
		union { FragT v; uint4 q[ES / 2]; } u;
		const char* src = a_lds + (mt * 16 + arow) * RS + (32 * ks + 8 * ag) * ES;
#pragma unroll
		for (int f = 0; f < ES / 2; ++f) u.q[f] = *(const uint4*)(src + 16 * f);
		return u.v;
	};
	float fs1[MT], fs2[MT];          // FOLD: this lane's sum / sum of squares over the k it multiplies (row lane & 15 of each m-tile)
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) { fs1[mt] = 0.f; fs2[mt] = 0.f; }
	auto mma_step = [&](int mt, const FragT& a, const WRaw& b) {
		if (FOLD) {
#pragma unroll
			for (int j = 0; j < 8; ++j) { const float f = (float)a[j]; fs1[mt] += f; fs2[mt] = fmaf(f, f, fs2[mt]); }
		}
		acc[mt] = mma16<T>(a, WF::dec(b), acc[mt]);
	};
#pragma unroll
	for (int u = 0; u < PRE; ++u)
		if (u < npre) {
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) mma_step(mt, LN ? load_a_lds(mt, ks0 + u) : apre[LN ? 0 : u][mt], bpre[u]);
		}
	constexpr int UN = PRE;
	int ks = ks0 + npre;
	for (; ks + UN <= ks1; ks += UN) {
		WRaw b[UN];
		FragT a[LN ? 1 : UN][MT];
#pragma unroll
		for (int u = 0; u < UN; ++u) {
			if (TTK_ABL & 8) b[u] = WRaw{}; else b[u] = TTK_WLOAD(wp + (int64_t)(ks + u) * 64);
			if (!LN) {
#pragma unroll
				for (int mt = 0; mt < MT; ++mt) a[u][mt] = load_a_global(mt, ks + u);
			}
		}
		__builtin_amdgcn_sched_barrier(0);      // keep the requests in front: sunk next to their MFMAs they become dependent round trips again
#pragma unroll
		for (int u = 0; u < UN; ++u)
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) mma_step(mt, LN ? load_a_lds(mt, ks + u) : a[LN ? 0 : u][mt], b[u]);
	}
	for (; ks < ks1; ++ks) {
		const WRaw b = (TTK_ABL & 8) ? WRaw{} : TTK_WLOAD(wp + (int64_t)ks * 64);
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) mma_step(mt, LN ? load_a_lds(mt, ks) : load_a_global(mt, ks), b);
	}
	if (FOLD) {   // the four lanes of a row (k-groups) -> every one holds the wave's partial; lane group 0 publishes it
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) {
			fs1[mt] = fold16_add(fs1[mt]); fs2[mt] = fold16_add(fs2[mt]);
			fs1[mt] = fold32_add(fs1[mt]); fs2[mt] = fold32_add(fs2[mt]);
			if (lane < 16) *(float2*)(rstat + ((wave * MT + mt) * 16 + lane) * 2) = make_float2(fs1[mt], fs2[mt]);
		}
	}

	TTK_STAMPD(3, acc[0][0]);
	// ---- cross-wave reduction through LDS, then epilogue by the first 256 threads
	float vsum[MT];
	if (TTK_ABL & 4) {
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) vsum[mt] = acc[mt][r & 3];
	} else {
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) *(f32x4*)(red + ((wave * MT + mt) * 64 + lane) * 4) = acc[mt];
		__syncthreads();
		TTK_STAMP(4);
		if (tid < 256) {
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) {
				float v = 0.f;
				for (int w = 0; w < nw; ++w) v += red[((w * MT + mt) * 64 + l2) * 4 + r];
				vsum[mt] = v;
			}
		}
	}
	if (p.ksplit > 1) {
		// Split-K over workgroups with a deterministic in-launch combine (cdna_hip_programming.md section 5, "In-launch split-K
		// reduction", write-through form): every slice stores its 16x16 partial with sc1 stores, drains them, and one lane
		// takes a ticket; the workgroup that draws the last ticket reads all slices back with sc1 loads IN SLICE ORDER (bitwise
		// reproducible, unlike float atomics) and runs the epilogue.  The ticket counter is reset by its last user.
		float* slab = p.slab + ((int64_t)nt * p.ksplit) * 256 * MT;
		if (tid < 256) {
#pragma unroll
			for (int mt = 0; mt < MT; ++mt)
				__hip_atomic_store(slab + ((int64_t)kslice * MT + mt) * 256 + tid, vsum[mt], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		int* flag = (int*)red;                  // LDS word reused as the "I am last" broadcast
		if (tid == 0) {
			const int ticket = __hip_atomic_fetch_add(p.tickets + nt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			*flag = ticket == p.ksplit - 1;
			if (ticket == p.ksplit - 1) __hip_atomic_store(p.tickets + nt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
		__syncthreads();
		if (!*flag) return;
		if (tid < 256) {
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) {
				float v = 0.f;
				for (int sl = 0; sl < p.ksplit; ++sl)
					v += __hip_atomic_load(slab + ((int64_t)sl * MT + mt) * 256 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				vsum[mt] = v;
			}
		}
	}
	float fmean[MT], frstd[MT];
	if (FOLD && tid < 256) {
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) {
			float a1 = 0.f, a2 = 0.f;
			for (int w = 0; w < nw; ++w) { const float2 t = *(const float2*)(rstat + ((w * MT + mt) * 16 + 4 * (l2 >> 4) + r) * 2); a1 += t.x; a2 += t.y; }
			const float mean = a1 / (float)p.K;
			fmean[mt] = mean;
			frstd[mt] = rsqrtf(fmaxf(a2 / (float)p.K - mean * mean, 0.f) + 1e-5f);      // E[x^2] - mean^2 in f32, as the LN prologue does
		}
	}
	if (!mine) return;
	if (TTK_ABL & 32) { if (vsum[0] == 1.2345e-30f) p.out_f32[0] = vsum[0]; return; }
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) {
		const int m = mt * 16 + 4 * (l2 >> 4) + r;
		if (m >= p.M) continue;
		const float v = FOLD ? (vsum[mt] - fmean[mt] * fcs) * frstd[mt] + bias : (W8 ? vsum[mt] * p.wscale : vsum[mt]) + bias;
		if (p.mode == SK_STORE_F32) {
			p.out_f32[(int64_t)m * p.ldc + n] = v;
			// mel head: the multinomial noise of the sampling launch that follows, one value per logit (SkinnyParams.qbuf in this mode)
			if (p.qbuf) {
				int mrow = p.max_ctx + m;                  // max_ctx: first row of this launch's row group
				const int grp = (int)rng.group;            // line batch: every line draws the same rows (a handful of subtractions, not a division routine in every instantiation)
				if (grp > 0) while (mrow >= grp) mrow -= grp;
				p.qbuf[(int64_t)m * p.ldc + n] = torch_exponential_at(rng, draw[mt], (rng.row0 + mrow) * (int64_t)p.N + n);
			}
		} else if (p.mode == SK_RESIDUAL) {
			p.out_f32[(int64_t)m * p.ldc + n] = res[mt] + v;
			if (p.out_T) ((T*)p.out_T)[((((int64_t)mt * (p.N / 32) + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (m & 15)) * 8 + (n & 7))] = cvt<T>(res[mt] + v);
		} else if (p.mode == SK_ACT_T) {
			const int64_t o = p.out_frag ? ((((int64_t)mt * (p.N / 32) + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (m & 15)) * 8 + (n & 7)) : (int64_t)m * p.N + n;
			((T*)p.out_T)[o] = cvt<T>(apply_act(v, p.act));
		} else {   // SK_QKV
			const int d = p.N / 3;
			const int which = n / d, c = n - which * d;
			if (which == 0) {
				p.qbuf[(int64_t)m * d + c] = v * p.q_scale;
			} else {
				const int h = c >> 6, dd = c & 63;
				T* cache = (T*)(which == 1 ? p.kcache : p.vcache);
				const int pos = kv_pos;
				if (pos < p.max_ctx) cache[(((int64_t)m * p.H + h) * p.max_ctx + pos) * 64 + dd] = cvt<T>(v);   // guard: never write past the cache
			}
		}
	}
	TTK_STAMP(5);
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // stores acknowledged
	TTK_STAMP(6);
#endif
}

template <typename T, int MT, bool W8>
static void launch_skinny_mt(const SkinnyParams& p, int waves, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
	const int grid = p.narrow ? ((p.N + 15) / 16) * p.narrow : ((p.N + 15) / 16) * p.ksplit;
	const size_t red = (size_t)waves * MT * 64 * 4 * sizeof(float);
	if (p.ln_count > 0) {
		if (waves > 8) waves = 8;
		const size_t lds = (size_t)16 * MT * (p.K * sizeof(T) + 16) + (size_t)waves * MT * 64 * 4 * sizeof(float);
		if (p.K <= 1024) {
			if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_skinny<T, MT, true, 4, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			hipExtLaunchKernelGGL((k_skinny<T, MT, true, 4, W8>), dim3(grid), dim3(64 * waves), (unsigned)lds, s, ea, eb, 0, p);
		} else {
			if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_skinny<T, MT, true, 8, W8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			hipExtLaunchKernelGGL((k_skinny<T, MT, true, 8, W8>), dim3(grid), dim3(64 * waves), (unsigned)lds, s, ea, eb, 0, p);
		}
	} else if (p.g1 && !W8) {
		const size_t lds = red + (size_t)waves * MT * 16 * 2 * sizeof(float);
		hipExtLaunchKernelGGL((k_skinny<T, MT, false, 1, false, true>), dim3(grid), dim3(64 * waves), (unsigned)lds, s, ea, eb, 0, p);
	} else {
		hipExtLaunchKernelGGL((k_skinny<T, MT, false, 1, W8>), dim3(grid), dim3(64 * waves), (unsigned)red, s, ea, eb, 0, p);
	}
}

template <typename T, bool W8>
static void launch_skinny_t(const SkinnyParams& p, int waves, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
	const int mt = decode_row_tiles(p.M);
	if (mt == 1) launch_skinny_mt<T, 1, W8>(p, waves, s, ea, eb);
	else if (mt == 2) launch_skinny_mt<T, 2, W8>(p, waves, s, ea, eb);
	else if (mt == 3) launch_skinny_mt<T, 3, W8>(p, waves, s, ea, eb);
	else launch_skinny_mt<T, 4, W8>(p, waves, s, ea, eb);
}

void launch_skinny(int dt, const SkinnyParams& p_in, int waves, hipStream_t s) {
	SkinnyParams p = p_in;
	if (p.ksplit < 1 || !p.slab || !p.tickets || (p.mode == SK_STORE_F32 && p.qbuf)) p.ksplit = 1;      // (mel head: slab / tickets carry the noise arguments)
	if (p.ln_count > 0 || p.mode == SK_QKV) p.narrow = 0;      // every workgroup of an LN kernel normalises all rows: more of them only adds work
	if (p.ln_count == 0 && p.g1) { p.narrow = 0; p.ksplit = 1; }   // folded LayerNorm: the row statistics need all of K inside the workgroup
	if (p.narrow) { p.ksplit = 1; p.narrow = p.narrow == 2 ? 2 : 4; }
	if (waves < 4) waves = 4;
	// algorithmic bytes: the weight matrix once + bias + the M activation rows in and out
	if (dt != DT_BF16) p.w8 = 0;                                // fp8 weights exist for the bf16 arithmetic only
	// profiling: the event pair travels with the dispatch packet (kernel start / stop timestamps), see prof_pair
	hipEvent_t ea = nullptr, eb = nullptr;
	if (g_prof_on) prof_pair(PROF_SKINNY, (double)p.N * p.K * (p.w8 ? 1 : dtype_size(dt)) + 4.0 * p.N + 4.0 * p.M * p.K + 4.0 * p.M * p.N, &ea, &eb);
	if (dt == DT_BF16) { if (p.w8) launch_skinny_t<bf16, true>(p, waves, s, ea, eb); else launch_skinny_t<bf16, false>(p, waves, s, ea, eb); }
	else if (dt == DT_F16) launch_skinny_t<f16, false>(p, waves, s, ea, eb);
	else launch_skinny_t<float, false>(p, waves, s, ea, eb);
}

}  // namespace ttk
