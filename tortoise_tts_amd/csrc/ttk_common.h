// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libttk.
// One arithmetic template parameter T in {float, __bf16, _Float16}: T is the storage/MFMA-operand type of weights and
// of GEMM/attention input activations; accumulators, residual streams, norm statistics and softmax are f32.
//   T = __bf16 : v_mfma_f32_16x16x32_bf16              (performance mode, BASELINE config 2)
//   T = _Float16 : v_mfma_f32_16x16x32_f16             (the reference's other autocast dtype, inference.py:331 / config.py:625-637)
//   T = float  : 8 x v_mfma_f32_16x16x4_f32 per k-step (exact-f32 parity mode; same tiling, same code)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef _Float16 f16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) float f32x8;

namespace ttk {

// A "fragment" is 8 consecutive-k elements of T held by one lane: for the 16x16 MFMA tile, lane l holds
// row/col (l & 15) and k-group g = l >> 4, i.e. k = 32*ks + 8*g + j, j = 0..7.
template <typename T> struct Frag;
template <> struct Frag<float> { typedef f32x8 type; };
template <> struct Frag<bf16> { typedef bf16x8 type; };
template <> struct Frag<f16> { typedef f16x8 type; };
// fp8-e4m3 (OCP) storage tag for the dense GEMM's operands in the fp8 mode of the diffusion network: one byte per element; a lane's
// 16-byte LDS read then holds its k elements for TWO v_mfma_f32_16x16x32_fp8_fp8 steps (low / high 8 bytes).
struct f8 { unsigned char v; };
template <> struct Frag<f8> { typedef uint4 type; };
// what a "T-typed" output means: the arithmetic type itself, bf16 for fp8 operands
template <typename T> struct OutOf { typedef T type; };
template <> struct OutOf<f8> { typedef bf16 type; };
// four f32 -> four fp8-e4m3 bytes (v_cvt_pk_fp8_f32: round to nearest even, saturating at +-448), element 0 in the low byte
__device__ __forceinline__ unsigned pack4_fp8(float a, float b, float c, float d) {
	int w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
	w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
	return (unsigned)w;
}

template <typename T>
__device__ __forceinline__ f32x4 mma16(typename Frag<T>::type a, typename Frag<T>::type b, f32x4 c);
template <>
__device__ __forceinline__ f32x4 mma16<bf16>(bf16x8 a, bf16x8 b, f32x4 c) {
	return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <>
__device__ __forceinline__ f32x4 mma16<f16>(f16x8 a, f16x8 b, f32x4 c) {
	return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
// f32: instruction j contracts k = {8g + j : g = 0..3}; the eight instructions together cover the same 32 k as
// the bf16 form with the same per-lane addressing (exact f32 fma chain, MI355X_MICROARCH "FP32-input MFMA").
template <>
__device__ __forceinline__ f32x4 mma16<float>(f32x8 a, f32x8 b, f32x4 c) {
#pragma unroll
	for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c, 0, 0, 0);
	return c;
}

template <typename T> __device__ __forceinline__ T cvt(float x);
template <> __device__ __forceinline__ float cvt<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 cvt<bf16>(float x) { return (bf16)x; }   // v_cvt_pk_bf16_f32, RNE, NaN-safe
template <> __device__ __forceinline__ f16 cvt<f16>(float x) { return (f16)x; }      // v_cvt_f16_f32, RNE; |x| > 65504 becomes inf, as in torch
// four f32 -> four 2-byte T as one 8-byte word (element 0 lowest)
template <typename T> __device__ __forceinline__ uint2 pack4_16(float a, float b, float c, float d) { return make_uint2(0u, 0u); }
template <> __device__ __forceinline__ uint2 pack4_16<bf16>(float a, float b, float c, float d) {
	union { bf16x4 v; uint2 u; } pk; pk.v = bf16x4{(bf16)a, (bf16)b, (bf16)c, (bf16)d}; return pk.u;
}
template <> __device__ __forceinline__ uint2 pack4_16<f16>(float a, float b, float c, float d) {
	union { f16x4 v; uint2 u; } pk; pk.v = f16x4{(f16)a, (f16)b, (f16)c, (f16)d}; return pk.u;
}
// the same with the element type chosen at run time (kernels without a T parameter that write a "T-typed" copy): ttk::ElemKind
enum ElemKind { EK_BF16 = 0, EK_F32 = 1, EK_F16 = 2 };
__device__ __forceinline__ void store4_kind(void* base, int64_t idx, float4 v, int kind) {
	if (kind == EK_F32) *(float4*)((float*)base + idx) = v;
	else if (kind == EK_F16) *(uint2*)((f16*)base + idx) = pack4_16<f16>(v.x, v.y, v.z, v.w);
	else *(uint2*)((bf16*)base + idx) = pack4_16<bf16>(v.x, v.y, v.z, v.w);
}

// Wave64 sum, result in every lane.  DPP inside each 16-lane row (quad swaps, row_half_mirror, row_mirror: VALU-rate, no LDS
// crossbar), then the four row sums are read back through SGPRs.  ~12 instructions instead of six dependent ds_bpermute.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
	return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum(float v) {
	v = dpp_add<0xB1>(v);    // quad_perm [1,0,3,2]
	v = dpp_add<0x4E>(v);    // quad_perm [2,3,0,1]
	v = dpp_add<0x141>(v);   // row_half_mirror
	v = dpp_add<0x140>(v);   // row_mirror
	const int iv = __float_as_int(v);
	const float r0 = __int_as_float(__builtin_amdgcn_readlane(iv, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(iv, 16));
	const float r2 = __int_as_float(__builtin_amdgcn_readlane(iv, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(iv, 48));
	return (r0 + r1) + (r2 + r3);
}
// Folds over lane bit 4 / bit 5 (pairs {lane, lane ^ 16} / {lane, lane ^ 32}) for commutative operations without ds_bpermute's address arithmetic (8 VALU
// instructions per __shfl_xor) and LDS round trip: v_permlane16_swap / v_permlane32_swap (gfx950) of a register with a copy of itself leave the pair's two
// values in two registers of every lane.  Same operands as `op(v, __shfl_xor(v, 16 | 32))`, so the same bits.
__device__ __forceinline__ float fmax_raw(float a, float b) {   // v_max_f32 without fmaxf's canonicalising v_max x, x in front (operands here are never sNaN)
	float r;
	asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
	return r;
}
__device__ __forceinline__ float fold16_max(float v) {
	const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
	return fmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fold32_max(float v) {
	const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
	return fmax_raw(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fold16_add(float v) {
	const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
	return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold32_add(float v) {
	const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
	return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
	return v;
}

// Branch-free activations on the hardware exp (v_exp_f32, ~1 ulp): they sit in GEMM epilogues, 64 values per lane.
__device__ __forceinline__ float gelu_new_f(float x) {   // HF:activations.py:59-66; tanh(u) = 1 - 2 / (1 + e^{2u})
	const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
	const float th = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * u));
	return 0.5f * x * (1.0f + th);
}
__device__ __forceinline__ float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float silu_precise(float x) { return silu_f(x); }

// GroupNorm32 "apply" arithmetic, ONE definition for every kernel that normalises (k_gn_apply, k_gn_apply_c1024 and the dense GEMM's fused form): the merge of
// a group's chunk triples (count, mean, M2) by 8 lanes (Chan et al.; lane `sub` takes chunks sub, sub + 8, ...; DPP sums leave the result in all 8) and the
// per-channel fold y = x * a + d.  Contraction is written out (fmaf where a product feeds a sum, nothing else fused): left to the compiler, two instantiations
// of one source line have rounded differently in the last bit (tests/diag/role_check.cpp), and a sequence must come out the same whichever kernel serves it.
// LOAD(ptr) fetches one float of the triples (plain load, or an agent-scope load where the triples were written by this very launch).
// Split in two since round 6: the REQUESTS of a lane's triples (gn_load_triples) and the arithmetic on them (gn_merge_loaded), so that a kernel can ask for the triples
// FIRST -- vmcnt retires in order: requested behind the rows, the triples could not be used before every row had landed, and the merge (a chain of DPP sums, a
// division and an rsqrt) started only then -- and so that chunk groups beyond nch are not requested at all (a lane asked for 24 dwords whatever nch was; at
// T = 1088, nch = 17, nine are real).  A skipped group contributes exact zeros, as the clamped loads' `ok ? x : 0` did: same bits.
template <int NG, typename LoadF>
__device__ __forceinline__ void gn_load_triples_n(const float* part, int nch, int sub, LoadF load, float (&cn)[8], float (&cm)[8], float (&c2)[8]) {
#pragma unroll
	for (int i = 0; i < 8; ++i) {   // up to 64 chunks per group of channels; the first NG groups of eight are requested (all at once: no branch between them)
		cn[i] = 0.f; cm[i] = 0.f; c2[i] = 0.f;
		if (i >= NG) continue;
		const int k = sub + 8 * i;
		const bool ok = k < nch;
		const int kk = ok ? k : 0;
		const float a0 = load(part + 3 * kk), a1 = load(part + 3 * kk + 1), a2 = load(part + 3 * kk + 2);
		cn[i] = ok ? a0 : 0.f; cm[i] = ok ? a1 : 0.f; c2[i] = ok ? a2 : 0.f;
	}
}
template <typename LoadF>
__device__ __forceinline__ void gn_load_triples(const float* part, int nch, int sub, LoadF load, float (&cn)[8], float (&cm)[8], float (&c2)[8]) {
	gn_load_triples_n<8>(part, nch, sub, load, cn, cm, c2);
}
__device__ __forceinline__ void gn_merge_loaded(const float (&cn)[8], const float (&cm)[8], const float (&c2)[8], float& mean, float& rstd) {
#pragma clang fp contract(off)
	float nt = 0.f, wsum = 0.f;
#pragma unroll
	for (int i = 0; i < 8; ++i) { nt += cn[i]; wsum = __builtin_fmaf(cn[i], cm[i], wsum); }
	nt = dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(nt)));
	wsum = dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(wsum)));
	mean = wsum / nt;
	float m2 = 0.f;
#pragma unroll
	for (int i = 0; i < 8; ++i) { const float d = cm[i] - mean; m2 += __builtin_fmaf(cn[i] * d, d, c2[i]); }
	m2 = dpp_add<0x141>(dpp_add<0x4E>(dpp_add<0xB1>(m2)));
	rstd = rsqrtf(m2 / nt + 1e-5f);
}
template <typename LoadF>
__device__ __forceinline__ void gn_merge_triples(const float* part, int nch, int sub, LoadF load, float& mean, float& rstd) {
	float cn[8], cm[8], c2[8];
	gn_load_triples(part, nch, sub, load, cn, cm, c2);
	gn_merge_loaded(cn, cm, c2, mean, rstd);
}
// y = x * a + d with a = rstd * gamma * (1 + scale), d = (beta - mean * rstd * gamma) * (1 + scale) + shift
__device__ __forceinline__ void gn_fold_coef(float mean, float rstd, float gamma, float beta, float scale, float shift, float& a, float& d) {
#pragma clang fp contract(off)
	const float s1 = 1.f + scale;
	a = rstd * gamma * s1;
	d = __builtin_fmaf(beta - mean * rstd * gamma, s1, shift);
}
__device__ __forceinline__ float gn_fold_apply(float x, float a, float d) { return __builtin_fmaf(x, a, d); }

enum Act { ACT_NONE = 0, ACT_GELU_NEW = 1, ACT_SILU = 2 };
__device__ __forceinline__ float apply_act(float v, int act) {
	if (act == ACT_GELU_NEW) return gelu_new_f(v);
	if (act == ACT_SILU) return silu_precise(v);
	return v;
}

// L2 touch of memory a LATER launch will read (weights of the GEMM that follows): every workgroup touches, one 128-byte line per thread,
// its share of the slice that the consumer's workgroups on the same XCD (blockIdx % 8) will fetch -- `taps` blocks of `bytes` each, a
// block's eighth x being XCD x's slice.  The touch is an LDS-DMA load into a sink nobody reads: a load into a VGPR the compiler
// considers dead could land after that register has been given a new value.  `sink` = LDS byte address of >= 256 bytes per wave.
__device__ __forceinline__ void l2_touch_for_next(const void* base, int64_t bytes, int taps, unsigned sink, int block, int nblocks, int tid, int nthreads) {
	const int xcd = block & 7, rank = block >> 3, nx = (nblocks + 7 - xcd) >> 3;      // workgroups on this XCD
	const int64_t slice = bytes / 8, lines = slice / 128;
	const int64_t per = (lines + nx - 1) / nx;
	for (int tap = 0; tap < taps; ++tap)
		for (int64_t l = tid; l < per; l += nthreads) {
			const int64_t line = (int64_t)rank * per + l;
			if (line < lines) {
				const char* a = (const char*)base + (int64_t)tap * bytes + (int64_t)xcd * slice + line * 128;
				unsigned keep;
				asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dword %2, off\n\ts_mov_b32 m0, %0"
							 : "=&s"(keep) : "s"(sink), "v"(a) : "memory");
			}
		}
}
// Kernel arguments into SGPRs as ONE batch of scalar loads behind one wait, at the top of the kernel.  Left alone hipcc sinks every argument's
// load next to its first use, behind branches: the generic decode GEMV made four DEPENDENT round trips to its argument block (~0.3 us each,
// cold scalar cache after every kernel boundary) before its first weight request, the dense GEMM six (tests/diag/ar_chain.cpp).  An empty asm
// that names the fields as SGPR inputs pins them: all loads are issued together, one s_waitcnt follows.  Up to 30 operands per statement.
#define TTK_PIN_ARGS(...) asm volatile("" :: __VA_ARGS__)
#define TTK_S(x) "s"(x)

// Diagnostic builds only (-DTTK_STAMPS=2, tests/diag/*_chain.cpp): every wave records 100 MHz timestamps, [linear workgroup][wave (16 slots)][8]; slot 7 of
// stamp 0 = XCC id.  The D form takes the stamp only once `dep` (a VGPR value) is really there -- the asm reads it, so the wave stalls on the MFMA /
// load that produces it first; a bare s_memrealtime has no data dependency and floats above the arithmetic it is meant to follow.  Expand to nothing
// in the product build.
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
#define TTK_WSTAMP(st, wg, i) do { if ((st) && (threadIdx.x & 63) == 0) { unsigned long long* st_ = (st) + ((size_t)(wg) * 16 + (threadIdx.x >> 6)) * 8; \
	st_[(i)] = __builtin_amdgcn_s_memrealtime(); if ((i) == 0) st_[7] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); } } while (0)
#define TTK_WSTAMPD(st, wg, i, dep) do { if (st) { unsigned tmp_; unsigned long long t_; \
	asm volatile("s_nop 7\n\tv_readfirstlane_b32 %0, %2\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(tmp_), "=s"(t_) : "v"(dep) : "memory"); \
	if ((threadIdx.x & 63) == 0) (st)[((size_t)(wg) * 16 + (threadIdx.x >> 6)) * 8 + (i)] = t_; } } while (0)
#else
#define TTK_WSTAMP(st, wg, i) do {} while (0)
#define TTK_WSTAMPD(st, wg, i, dep) do {} while (0)
#endif

__device__ __forceinline__ unsigned lds_byte_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

}  // namespace ttk
