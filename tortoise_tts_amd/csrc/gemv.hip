// Lean decode-step GEMV: the four weight-streaming launches of a GPT-2 layer and the mel head at the benchmarked geometry, one
// instantiation per role with every shape decision taken at compile time.
//
// Why a second kernel family beside k_skinny (skinny.hip): in-kernel timestamps of every wave (tests/diag/ar_chain.cpp, profiles/
// r03_ar_chain_*.log) showed the generic kernel spending 1.2 - 1.5 us between a wave's first instruction and its first weight request -- as
// long as the weights then take to arrive.  Its prologue is ~350 instructions of run-time shape arithmetic (integer divisions by ksplit / narrow /
// waves, mode branches) issued by one wave per SIMD, and it reads its 248 bytes of kernel arguments in FOUR dependent scalar-load round trips
// (the compiler sinks each field's load next to its first use, behind branches), the second of which also waits for the cache-length word.
// Here the role (epilogue), the waves per workgroup, the k-steps per wave and the tile geometry are template parameters, the arguments are 37
// dwords pinned into SGPRs by ONE batch of scalar loads, and the first weight request leaves ~40 instructions after the wave starts.
// The products are k_skinny's, operation for operation (same fragments, same k order per wave, same wave order in the cross-wave sum), so the
// projections and the head are bit-identical to the generic kernel's; the folded launches differ from it only in how a lane sums its row
// statistics (dot2 instructions, see fold_stats) -- a few ulp of mean / rstd.  k_skinny stays the path for every other geometry (small models,
// row groups, LayerNorm-prologue form, split-K) and is what tests/test_gpu_gemv.py compares against.
//
//   GV_QKV   ln_1 (folded) + c_attn + bias -> q (pre-scaled) / K, V appended to the cache   (HF:models/gpt2/modeling_gpt2.py:144-226)
//   GV_PROJ  c_proj / mlp.c_proj + bias + residual, 4-column workgroups; also writes the T-typed fragment-order copy of the rows
//   GV_FC    ln_2 (folded) + c_fc + bias + gelu_new -> fragment-order T                    (HF:activations.py:59-66)
//   GV_HEAD  mel_head over the normalised rows (+ the multinomial noise of the sampling launch, + the cache-length bump)
//                                                                                            (unified_voice.py:106,239)
// Algorithmic bytes per launch: N*K*sizeof(T) weight bytes (+ M*K*sizeof(T) activations from L2).
#include <hip/hip_ext.h>

#include "ttk_common.h"
#include "ttk_kernels.h"
#include "ttk_rng.h"

namespace ttk {

#ifndef TTK_NT
#define TTK_NT 1
#endif
#if TTK_NT
#define GV_WLOAD(p) __builtin_nontemporal_load(p)
#else
#define GV_WLOAD(p) (*(p))
#endif

#if defined(TTK_STAMPS) && TTK_STAMPS == 2   // tests/diag/ar_chain.cpp: every wave stamps, [workgroup][wave (16 slots)][8]; slot 7 = XCC id
#define GV_STAMP(i) do { if (p.stamps && (threadIdx.x & 63) == 0) { unsigned long long* st_ = p.stamps + ((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8; \
	st_[(i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0) st_[7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); } } while (0)
#define GV_STAMPD(i, dep) do { if (p.stamps) { unsigned tmp_; unsigned long long t_; \
	asm volatile("s_nop 7\n\tv_readfirstlane_b32 %0, %2\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tmp_), "=s"(t_) : "v"(dep) : "memory"); \
	if ((threadIdx.x & 63) == 0) p.stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = t_; } } while (0)
#else
#define GV_STAMP(i) do {} while (0)
#define GV_STAMPD(i, dep) do {} while (0)
#endif

// weight fragment as stored: the MFMA operand, or (W8, bf16 arithmetic) 8 fp8-e4m3 bytes widened exactly next to their MFMA (skinny.hip: WFrag)
template <typename T, bool W8> struct GvW {
	typedef typename Frag<T>::type raw;
	static __device__ __forceinline__ typename Frag<T>::type dec(raw r) { return r; }
};
template <> struct GvW<bf16, true> {
	typedef unsigned raw __attribute__((ext_vector_type(2)));
	static __device__ __forceinline__ bf16x8 dec(raw r) {
		typedef float f2 __attribute__((ext_vector_type(2)));
		const f2 a = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[0], false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[0], true);
		const f2 c = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[1], false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)r[1], true);
		return bf16x8{(bf16)a[0], (bf16)a[1], (bf16)b[0], (bf16)b[1], (bf16)c[0], (bf16)c[1], (bf16)d[0], (bf16)d[1]};
	}
};

// Folded LayerNorm: a lane's contribution to its row's sum and sum of squares from the 8 elements of one A fragment.  16-bit types: two
// v_dot2c_f32_{bf16,f16} per element pair (x.x and x.1, f32 accumulate) instead of unpack + add + fma per element -- 8 VALU issues per
// fragment against 40; with one wave per SIMD the 320 dependent issues of a wave's 8 fragments were 0.35 us of every folded launch.
template <typename T> __device__ __forceinline__ void fold_stats(const typename Frag<T>::type& a, float& s1, float& s2) {
#pragma unroll
	for (int j = 0; j < 8; ++j) { const float f = (float)a[j]; s1 += f; s2 = fmaf(f, f, s2); }
}
template <> __device__ __forceinline__ void fold_stats<bf16>(const bf16x8& a, float& s1, float& s2) {
	typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
	const bf16x2 one = {(bf16)1.0f, (bf16)1.0f};
#pragma unroll
	for (int j = 0; j < 8; j += 2) {
		const bf16x2 x = {a[j], a[j + 1]};
		s2 = __builtin_amdgcn_fdot2_f32_bf16(x, x, s2, false);
		s1 = __builtin_amdgcn_fdot2_f32_bf16(x, one, s1, false);
	}
}
template <> __device__ __forceinline__ void fold_stats<f16>(const f16x8& a, float& s1, float& s2) {
	typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
	const f16x2 one = {(f16)1.0f, (f16)1.0f};
#pragma unroll
	for (int j = 0; j < 8; j += 2) {
		const f16x2 x = {a[j], a[j + 1]};
		s2 = __builtin_amdgcn_fdot2(x, x, s2, false);
		s1 = __builtin_amdgcn_fdot2(x, one, s1, false);
	}
}

// NW waves per workgroup, each multiplying KPW consecutive k-steps (K = 32 * NW * KPW); MT 16-row tiles of candidates.
template <typename T, int MT, int ROLE, int NW, int KPW, bool W8>
__global__ __launch_bounds__(64 * NW) void k_gemv(GemvParams p) {
	typedef typename Frag<T>::type FragT;
	typedef GvW<T, W8> WF;
	typedef typename WF::raw WRaw;
	constexpr int ES = sizeof(T);
	constexpr bool FOLD = ROLE == GV_QKV || ROLE == GV_FC;
	constexpr bool NARROW = ROLE == GV_PROJ;
	constexpr int KS = NW * KPW;                            // k-steps of the whole matrix
	extern __shared__ __attribute__((aligned(16))) float gv_red[];
	float* red = gv_red;                                     // [wave][m_tile][lane][4]
	float* rstat = red + NW * MT * 64 * 4;                   // FOLD: [wave][m_tile][16 rows][sum, sum of squares]
	// every argument into SGPRs NOW, as one batch of scalar loads behind one wait (left alone the compiler fetches each field next to its first
	// use: four dependent round trips before the first weight request)
	asm volatile("" :: "s"(p.Wp), "s"(p.a), "s"(p.bias), "s"(p.csum), "s"(p.out_f32), "s"(p.out_T), "s"(p.qbuf), "s"(p.kcache), "s"(p.vcache),
				 "s"(p.d_pos), "s"(p.noise), "s"(p.rng), "s"(p.draws), "s"(p.health), "s"(p.M), "s"(p.N), "s"(p.max_ctx), "s"(p.H), "s"(p.row0), "s"(p.q_scale), "s"(p.wscale));
	GV_STAMP(0);
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	int nt, sub = 0;
	if (NARROW) { const int b = blockIdx.x; nt = ((b >> 5) << 3) + (b & 7); sub = (b >> 3) & 3; }   // the 4 workgroups of a tile: ids equal mod 8 -> one XCD, one L2
	else nt = blockIdx.x;

	// ---- epilogue role of the first 256 threads: element (row 4*(l2>>4)+r of each m-tile, column l2&15 [narrow: &3]) of the output tile.  Bias,
	// column sum, residual value, cache position and noise counters are requested first (oldest in the vmcnt order): the epilogue finds them there.
	const int l2 = lane, r = wave & 3;
	const int n = NARROW ? nt * 16 + 4 * sub + (l2 & 3) : nt * 16 + (l2 & 15);
	const bool mine = tid < 256 && (!NARROW || (l2 & 15) < 4) && n < p.N;
	const int nn = n < p.N ? n : p.N - 1;
	float bias = 0.f, fcs = 0.f, res[MT];
	RngArgs rng = {};
	int64_t draw[MT];
	int kv_pos = 0;
	if (ROLE == GV_QKV) kv_pos = *p.d_pos;
	if (ROLE == GV_HEAD && p.d_pos && blockIdx.x == 0 && tid == 0) *(int*)p.d_pos += 1;      // nothing in this launch reads it; the next step does
	const bool noise = ROLE == GV_HEAD && p.noise != nullptr;
	if (p.bias) bias = p.bias[nn];
	if (FOLD) fcs = p.csum[nn];
	if (noise) rng = *(const RngArgs*)p.rng;
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) {
		res[mt] = 0.f; draw[mt] = 0;
		int m = mt * 16 + 4 * (l2 >> 4) + r;
		m = m < p.M ? m : p.M - 1;
		if (ROLE == GV_PROJ) res[mt] = p.out_f32[(int64_t)m * p.N + nn];
		if (noise) draw[mt] = p.draws[m];
	}

	// ---- operands: weights and rows in fragment order, requested in batches of PRE k-steps (B0 A0 B1 A1 ...), all before the first use
	constexpr int OPREGS = (MT + 1) * (ES == 4 ? 8 : 4);
	constexpr int PRE0 = 64 / OPREGS >= 8 ? 8 : (64 / OPREGS >= 4 ? 4 : (64 / OPREGS >= 2 ? 2 : 1));
	constexpr int PRE = PRE0 < KPW ? PRE0 : KPW;
	static_assert(KPW % PRE == 0, "k-steps per wave must be a multiple of the operand batch");
	const int ks0 = wave * KPW;
	const WRaw* wp = (const WRaw*)p.Wp + ((int64_t)nt * KS + ks0) * 64 + (NARROW ? ((lane & ~15) | (4 * sub + (lane & 3))) : lane);
	const FragT* ap = (const FragT*)p.a + (int64_t)ks0 * 64 + lane;      // [m_tile][KS][lane]: rows >= M hold zeros
	f32x4 acc[MT];
	float fs1[MT], fs2[MT];
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) { acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f}; fs1[mt] = 0.f; fs2[mt] = 0.f; }
	// Narrow tiles (GV_PROJ), 16-bit types: a lane's B fragment is column 4*sub + (n & 3), so requesting it per lane asks for every 16-byte chunk
	// four times over -- 1 KiB of requests per k-step for 256 distinct bytes, and with the 128 KB of rows a workgroup of mlp.c_proj pulls, the CU's
	// 64 B/clk request path was what bounded the launch (256 KB of requests: ~2 us).  Instead one DENSE request per four k-steps (lane = (k-step,
	// k-group, column): 64 distinct chunks), parked in a wave-private LDS strip and read back with the broadcast pattern the MFMA wants (four
	// lanes per address: free on LDS).  Same bits into the same MFMAs.
	constexpr bool WLDS = NARROW && ES == 2;
	if constexpr (WLDS) {
		static_assert(KPW % 4 == 0, "dense weight requests cover four k-steps");
		constexpr int WL = KPW / 4;
		WRaw* wl = (WRaw*)(rstat) + wave * (KPW * 16);                     // [k-step][k-group][column] chunks of this wave (behind the reduce area)
		const WRaw* wd = (const WRaw*)p.Wp + ((int64_t)nt * KS + ks0 + (lane >> 4)) * 64 + ((lane >> 2) & 3) * 16 + 4 * sub + (lane & 3);
		WRaw wq[WL];
#pragma unroll
		for (int i = 0; i < WL; ++i) wq[i] = GV_WLOAD(wd + i * 4 * 64);
		FragT a0[PRE][MT];
#pragma unroll
		for (int u = 0; u < PRE; ++u)
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) a0[u][mt] = ap[((int64_t)mt * KS + u) * 64];
		GV_STAMP(2);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int i = 0; i < WL; ++i) wl[i * 64 + lane] = wq[i];
		const int rd = ((lane >> 4) << 2) + (lane & 3);                    // chunk of (k-group, column) inside a k-step's 16
#pragma unroll
		for (int kb = 0; kb < KPW; kb += PRE) {
			FragT a[PRE][MT];
			if (kb + PRE < KPW) {      // the next batch of rows leaves before this one is multiplied
#pragma unroll
				for (int u = 0; u < PRE; ++u)
#pragma unroll
					for (int mt = 0; mt < MT; ++mt) a[u][mt] = ap[((int64_t)mt * KS + kb + PRE + u) * 64];
			}
			__builtin_amdgcn_sched_barrier(0);
#pragma unroll
			for (int u = 0; u < PRE; ++u) {
				const WRaw b = wl[(kb + u) * 16 + rd];
#pragma unroll
				for (int mt = 0; mt < MT; ++mt) acc[mt] = mma16<T>(a0[u][mt], WF::dec(b), acc[mt]);
			}
			if (kb + PRE < KPW) {
#pragma unroll
				for (int u = 0; u < PRE; ++u)
#pragma unroll
					for (int mt = 0; mt < MT; ++mt) a0[u][mt] = a[u][mt];
			}
		}
	} else {
#pragma unroll
	for (int kb = 0; kb < KPW; kb += PRE) {
		WRaw b[PRE];
		FragT a[PRE][MT];
#pragma unroll
		for (int u = 0; u < PRE; ++u) {
			b[u] = GV_WLOAD(wp + (kb + u) * 64);
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) a[u][mt] = ap[((int64_t)mt * KS + kb + u) * 64];
		}
		if (kb == 0) GV_STAMP(2);
		__builtin_amdgcn_sched_barrier(0);      // keep the requests in front: sunk next to their MFMAs they become dependent round trips
#pragma unroll
		for (int u = 0; u < PRE; ++u)
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) {
				if (FOLD) fold_stats<T>(a[u][mt], fs1[mt], fs2[mt]);
				acc[mt] = mma16<T>(a[u][mt], WF::dec(b[u]), acc[mt]);
			}
	}
	}
	if (FOLD) {   // the four lanes of a row (k-groups) -> the wave's partial; lane group 0 publishes it
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) {
			fs1[mt] = fold16_add(fs1[mt]); fs2[mt] = fold16_add(fs2[mt]);
			fs1[mt] = fold32_add(fs1[mt]); fs2[mt] = fold32_add(fs2[mt]);
			if (lane < 16) *(float2*)(rstat + ((wave * MT + mt) * 16 + lane) * 2) = make_float2(fs1[mt], fs2[mt]);
		}
	}
	GV_STAMPD(3, acc[0][0]);
	// ---- cross-wave sum through LDS (wave order), epilogue by the first 256 threads
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) *(f32x4*)(red + ((wave * MT + mt) * 64 + lane) * 4) = acc[mt];
	__syncthreads();
	GV_STAMP(4);
	if (!mine) return;
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) {
		const int m = mt * 16 + 4 * (l2 >> 4) + r;
		if (m >= p.M) continue;
		float vs = 0.f;
#pragma unroll
		for (int w = 0; w < NW; ++w) vs += red[((w * MT + mt) * 64 + l2) * 4 + r];
		float v;
		if (FOLD) {
			float a1 = 0.f, a2 = 0.f;
#pragma unroll
			for (int w = 0; w < NW; ++w) { const float2 t = *(const float2*)(rstat + ((w * MT + mt) * 16 + 4 * (l2 >> 4) + r) * 2); a1 += t.x; a2 += t.y; }
			const float mean = a1 / (float)(32 * KS);
			const float var = fmaxf(fmaf(-mean, mean, a2 / (float)(32 * KS)), 0.f);      // the fused form k_skinny's `a2 / K - mean * mean` contracts to
			const float rstd = rsqrtf(var + 1e-5f);      // E[x^2] - mean^2 in f32, as the LN prologue does
			v = (vs - mean * fcs) * rstd + bias;
			// what the fold cannot represent well is reported, not hidden (one workgroup looks; the branch is never taken on a healthy stream)
			if (p.health && blockIdx.x == 0 && (l2 & 15) == 0) {
				const int bad = (mean * mean > 64.f * (var + 1e-5f) ? 1 : 0) | (a2 - a2 != 0.f ? 2 : 0);
				if (bad) atomicOr(p.health, bad);
			}
		} else {
			v = (W8 ? vs * p.wscale : vs) + bias;
		}
		if (ROLE == GV_HEAD) {
			p.out_f32[(int64_t)m * p.N + n] = v;
			if (noise) {
				int mrow = p.row0 + m;
				const int grp = (int)rng.group;            // line batch: every line draws the same rows
				if (grp > 0) while (mrow >= grp) mrow -= grp;
				p.noise[(int64_t)m * p.N + n] = torch_exponential_at(rng, draw[mt], (rng.row0 + mrow) * (int64_t)p.N + n);
			}
		} else if (ROLE == GV_PROJ) {
			p.out_f32[(int64_t)m * p.N + n] = res[mt] + v;
			if (p.out_T) ((T*)p.out_T)[((((int64_t)mt * (p.N >> 5) + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (m & 15)) * 8 + (n & 7))] = cvt<T>(res[mt] + v);
		} else if (ROLE == GV_FC) {
			((T*)p.out_T)[((((int64_t)mt * (p.N >> 5) + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (m & 15)) * 8 + (n & 7))] = cvt<T>(gelu_new_f(v));
		} else {   // GV_QKV: n in [0, 3d), d = 32 * KS
			constexpr int d = 32 * KS;
			const int which = n / d, c = n - which * d;
			if (which == 0) {
				p.qbuf[(int64_t)m * d + c] = v * p.q_scale;
			} else {
				const int h = c >> 6, dd = c & 63;
				T* cache = (T*)(which == 1 ? p.kcache : p.vcache);
				if (kv_pos < p.max_ctx) cache[(((int64_t)m * p.H + h) * p.max_ctx + kv_pos) * 64 + dd] = cvt<T>(v);   // guard: never write past the cache
			}
		}
	}
	GV_STAMP(5);
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	GV_STAMP(6);
#endif
}

template <typename T, int MT, int ROLE, int NW, int KPW, bool W8>
static void gemv_go(const GemvParams& p, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
	constexpr bool FOLD = ROLE == GV_QKV || ROLE == GV_FC;
	const int tiles = (p.N + 15) / 16;
	const int grid = ROLE == GV_PROJ ? tiles * 4 : tiles;
	const size_t lds = (size_t)NW * MT * 64 * 4 * sizeof(float) + (FOLD ? (size_t)NW * MT * 16 * 2 * sizeof(float) : 0)
					   + (ROLE == GV_PROJ && sizeof(T) == 2 ? (size_t)NW * KPW * 16 * (W8 ? 8 : 16) : 0);      // wave-private weight strips of the narrow tiles
	hipExtLaunchKernelGGL((k_gemv<T, MT, ROLE, NW, KPW, W8>), dim3(grid), dim3(64 * NW), (unsigned)lds, s, ea, eb, 0, p);
}

template <typename T, int ROLE, int NW, int KPW, bool W8>
static void gemv_mt(const GemvParams& p, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
	const int mt = decode_row_tiles(p.M);
	if (mt == 1) gemv_go<T, 1, ROLE, NW, KPW, W8>(p, s, ea, eb);
	else if (mt == 2) gemv_go<T, 2, ROLE, NW, KPW, W8>(p, s, ea, eb);
	else if (mt == 3 && sizeof(T) == 2) gemv_go<T, sizeof(T) == 2 ? 3 : 2, ROLE, NW, KPW, W8>(p, s, ea, eb);
	else if (sizeof(T) == 2) gemv_go<T, sizeof(T) == 2 ? 4 : 2, ROLE, NW, KPW, W8>(p, s, ea, eb);      // (f32 batches end at 32 rows: gemv_supported)
}

// the (waves, k-steps per wave) the decode step uses per role and type: 16-bit 4 x 8 (K = 1024) / 8 x 16 (K = 4096); f32 batches are 4
// k-steps, so 8 x 4 / 8 x 16; the head runs 4 waves in every type (513 tiles: four-wave workgroups keep them one round)
template <typename T, int ROLE, bool W8>
static bool gemv_role(const GemvParams& p, hipStream_t s, hipEvent_t ea, hipEvent_t eb) {
	constexpr bool F32 = sizeof(T) == 4;
	if (ROLE == GV_HEAD) {
		if (p.K != 1024) return false;
		gemv_mt<T, ROLE, 4, 8, W8>(p, s, ea, eb);
	} else if (p.K == 1024) {
		gemv_mt<T, ROLE, F32 ? 8 : 4, F32 ? 4 : 8, W8>(p, s, ea, eb);
	} else if (p.K == 4096 && ROLE == GV_PROJ) {
		gemv_mt<T, ROLE, 8, 16, W8>(p, s, ea, eb);
	} else {
		return false;
	}
	return true;
}

bool gemv_supported(int dt, int role, const GemvParams& p) {
	if (dt != DT_BF16 && dt != DT_F16 && dt != DT_F32) return false;
	if (p.M < 1 || p.M > (dt == DT_F32 ? 32 : 64)) return false;
	if (p.w8 && (dt != DT_BF16 || role != GV_PROJ)) return false;
	if (role == GV_HEAD) return p.K == 1024;
	if (role == GV_PROJ) return (p.K == 1024 || p.K == 4096) && p.N % 128 == 0;       // narrow tile order needs n-tiles % 8 == 0
	if (role == GV_QKV) return p.K == 1024 && p.N == 3 * p.K;
	return p.K == 1024 && p.N % 32 == 0;
}

bool launch_gemv(int dt, int role, const GemvParams& p, hipStream_t s) {
	if (!gemv_supported(dt, role, p)) return false;
	hipEvent_t ea = nullptr, eb = nullptr;      // profiling: the event pair travels with the dispatch packet (kernel start / stop timestamps)
	if (g_prof_on) prof_pair(PROF_SKINNY, (double)p.N * p.K * (p.w8 ? 1 : dtype_size(dt)) + 4.0 * p.N + 4.0 * p.M * p.K + 4.0 * p.M * p.N, &ea, &eb);
#define GV_DISPATCH(T, W8) \
	(role == GV_QKV ? gemv_role<T, GV_QKV, false>(p, s, ea, eb) : role == GV_FC ? gemv_role<T, GV_FC, false>(p, s, ea, eb) : \
	 role == GV_HEAD ? gemv_role<T, GV_HEAD, false>(p, s, ea, eb) : gemv_role<T, GV_PROJ, W8>(p, s, ea, eb))
	if (dt == DT_BF16) return p.w8 ? GV_DISPATCH(bf16, true) : GV_DISPATCH(bf16, false);
	if (dt == DT_F16) return GV_DISPATCH(f16, false);
	return GV_DISPATCH(float, false);
#undef GV_DISPATCH
}

}  // namespace ttk
