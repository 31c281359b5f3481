// LayerNorm (GPT-2 ln_1/ln_2/ln_f + final_norm) and GroupNorm32 (+SiLU, +timestep scale/shift) -- HBM/L2-bound
// row passes over f32 residual streams that emit the T-typed operand of the next MFMA GEMM.
//   LayerNorm:   HF:models/gpt2/modeling_gpt2.py:246-310 (eps 1e-5), unified_voice.py:442,519
//   GroupNorm32: /root/reference/tortoise_tts/models/arch_utils.py:24-44 (32 groups, float math, eps 1e-5),
//                used by ResBlock diffusion.py:1338-1372 and AttentionBlock arch_utils.py:163,186
// Internal layout is channels-last [nb][T][C], so a group is (T rows) x (C/32 contiguous channels).
#include <stdint.h>
#include <stdlib.h>

#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

// one wave per row; d <= 4096
// frag != 0: `out` is written in the MFMA A-fragment order of the skinny decode GEMV ([m_tile][d/32][lane][8], skinny.hip) instead of row-major
// NI = float4 slots per lane (d <= 256 * NI).  NI = 4 (d <= 1024: the decode step's ln_f + final_norm launch, one per token): the affine parameters of BOTH
// norms are requested together with the row -- loaded where they are used they were two more dependent round trips behind the reductions (7.3 us per
// launch in the kernel trace for 16 rows); NI = 16 keeps them at their use (64 float4 of parameters per lane would not fit the register file).
template <typename OT, int NI>
__global__ void k_layernorm(const float* x, int64_t ldx, int rows, int d, const float* g1, const float* b1,
							const float* g2, const float* b2, OT* out, int64_t ldo, int frag, float* out2, const int64_t* out2_idx, int64_t out2_stride, const int64_t* out2_base) {
	constexpr bool HOIST = NI <= 4;
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (row >= rows) return;
	// slot of a ring the caller advances on the device (ttk_ar_set_hidden_ring); the ring's BASE comes through device memory as well: a captured launch
	// must serve the buffer of whichever generation replays it, not the one it was captured in
	if (out2_base) out2 = (float*)(uintptr_t)out2_base[0];
	if (out2 && out2_idx) out2 += out2_idx[0] * out2_stride;
	const int nchunk = d / 4;
	float4 v[NI];
	float4 pg[HOIST ? 2 : 1][HOIST ? NI : 1], pb[HOIST ? 2 : 1][HOIST ? NI : 1];
#pragma unroll
	for (int i = 0; i < NI; ++i) {
		const int c = lane + 64 * i;
		v[i] = c < nchunk ? *(const float4*)(x + (int64_t)row * ldx + 4 * c) : make_float4(0, 0, 0, 0);
	}
	if (HOIST) {
#pragma unroll
		for (int i = 0; i < (HOIST ? NI : 0); ++i) {
			const int c = lane + 64 * i < nchunk ? lane + 64 * i : 0;
			pg[0][i] = *(const float4*)(g1 + 4 * c); pb[0][i] = *(const float4*)(b1 + 4 * c);
			if (g2) { pg[HOIST ? 1 : 0][i] = *(const float4*)(g2 + 4 * c); pb[HOIST ? 1 : 0][i] = *(const float4*)(b2 + 4 * c); }
		}
	}
#pragma unroll
	for (int pass = 0; pass < 2; ++pass) {
		if (pass == 1 && !g2) break;
		const float* g = pass ? g2 : g1;
		const float* b = pass ? b2 : b1;
		float sum = 0.f;
#pragma unroll
		for (int i = 0; i < NI; ++i) sum += v[i].x + v[i].y + v[i].z + v[i].w;
		const float mean = wave_sum(sum) / (float)d;
		float sq = 0.f;
#pragma unroll
		for (int i = 0; i < NI; ++i)
			if (lane + 64 * i < nchunk) {
				const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
				sq += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
			}
		const float rstd = rsqrtf(wave_sum(sq) / (float)d + 1e-5f);
#pragma unroll
		for (int i = 0; i < NI; ++i) {
			const int c = lane + 64 * i;
			if (c < nchunk) {
				const float4 gg = HOIST ? pg[HOIST ? pass : 0][HOIST ? i : 0] : *(const float4*)(g + 4 * c), bb = HOIST ? pb[HOIST ? pass : 0][HOIST ? i : 0] : *(const float4*)(b + 4 * c);
				v[i].x = (v[i].x - mean) * rstd * gg.x + bb.x;
				v[i].y = (v[i].y - mean) * rstd * gg.y + bb.y;
				v[i].z = (v[i].z - mean) * rstd * gg.z + bb.z;
				v[i].w = (v[i].w - mean) * rstd * gg.w + bb.w;
			}
		}
	}
#pragma unroll
	for (int i = 0; i < NI; ++i) {
		const int c = lane + 64 * i;
		if (c < nchunk) {
			const int n = 4 * c;
			OT* o = frag ? out + ((((int64_t)(row >> 4) * (d / 32) + (n >> 5)) * 64 + ((n >> 3) & 3) * 16 + (row & 15)) * 8 + (n & 7)) : out + (int64_t)row * ldo + n;
			o[0] = (OT)v[i].x; o[1] = (OT)v[i].y; o[2] = (OT)v[i].z; o[3] = (OT)v[i].w;
			if (out2) *(float4*)(out2 + (int64_t)row * d + n) = v[i];
		}
	}
}

void launch_layernorm(int dt, const float* x, int64_t ldx, int rows, int d, const float* g1, const float* b1,
					  const float* g2, const float* b2, void* out, int64_t ldo, int out_f32, hipStream_t s, int frag, float* out2,
					  const int64_t* out2_idx, int64_t out2_stride, const int64_t* out2_base) {
	ProfScope prof(PROF_LAYERNORM, (double)rows * d * (4.0 + (out_f32 ? 4.0 : dtype_size(dt))), s);
	const int grid = (rows + 3) / 4;
#define LN_GO(OT, NI) hipLaunchKernelGGL((k_layernorm<OT, NI>), dim3(grid), dim3(256), 0, s, x, ldx, rows, d, g1, b1, g2, b2, (OT*)out, ldo, frag, out2, out2_idx, out2_stride, out2_base)
	if (out_f32 || dt == DT_F32) { if (d <= 1024) LN_GO(float, 4); else LN_GO(float, 16); }
	else if (dt == DT_F16) { if (d <= 1024) LN_GO(f16, 4); else LN_GO(f16, 16); }
	else { if (d <= 1024) LN_GO(bf16, 4); else LN_GO(bf16, 16); }
#undef LN_GO
}

// ---------------------------------------------------------------- GroupNorm32
// stats: grid (32 groups, nb, nchunks).  Each workgroup holds its (rows x cpg) chunk (<= 2048 values) in registers, computes the
// chunk's (count, mean, M2) with an exact two-pass, and writes the triple; the consumer merges the chunk triples with Chan's
// parallel-variance formula (k_gn_apply), so no second reduction launch and no atomics (bitwise reproducible).
__global__ __launch_bounds__(256) void k_gn_stats(const float* x, int T, int C, int rows_per_chunk, float* part, const int* tlen, const int* need) {
	const int g = blockIdx.x, b = blockIdx.y, ch = blockIdx.z, cpg = C / 32;
	if (need && !need[b]) return;             // this sequence keeps the triples the producing GEMM's epilogue wrote
	const int Tl = tlen ? tlen[b] : T;        // ragged batch: the sequence's own length inside its T-row slot
	const int t0 = ch * rows_per_chunk;
	if (t0 >= Tl) return;                     // chunk beyond the sequence: k_gn_apply does not read it
	const int rows = min(rows_per_chunk, Tl - t0);
	const float* base = x + ((int64_t)b * T + t0) * C + g * cpg;
	const int n = rows * cpg;
	__shared__ float sh[4];
	float v[8];
	float sum = 0.f;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const int e = threadIdx.x + 256 * i;
		const int t = e / cpg, c = e - t * cpg;
		v[i] = e < n ? base[(int64_t)t * C + c] : 0.f;
		sum += v[i];
	}
	sum = wave_sum(sum);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
	__syncthreads();
	const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)n;
	__syncthreads();
	float sq = 0.f;
#pragma unroll
	for (int i = 0; i < 8; ++i)
		if (threadIdx.x + 256 * i < n) { const float d = v[i] - mean; sq += d * d; }
	sq = wave_sum(sq);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
	__syncthreads();
	if (threadIdx.x == 0) {
		float* o = part + (((int64_t)b * 32 + g) * gridDim.z + ch) * 3;
		o[0] = (float)n; o[1] = mean; o[2] = sh[0] + sh[1] + sh[2] + sh[3];
	}
}

int gn_rows_per_chunk(int C) { const int r = 2048 / (C / 32); return r < 1 ? 1 : r; }
int gn_num_chunks(int T, int C) { const int r = gn_rows_per_chunk(C); return (T + r - 1) / r; }

void launch_gn_stats(const float* x, int nb, int T, int C, float* part, hipStream_t s, const int* tlen, const int* need) {
	ProfScope prof(PROF_GN_STATS, 4.0 * nb * T * C, s);
	hipLaunchKernelGGL(k_gn_stats, dim3(32, nb, gn_num_chunks(T, C)), dim3(256), 0, s, x, T, C, gn_rows_per_chunk(C), part, tlen, need);
}

// apply: a block owns a strip of output rows of ONE batch element.  It first merges that element's 32 groups' chunk statistics
// (Chan et al.) into LDS, then streams its rows: thread = 4 consecutive channels (one group, since C/32 % 4 == 0).
// GN_PASSES rows per thread: a block's fixed cost (merging 32 groups x nchunks triples, the barrier) is paid once per strip of 256 / (C/4) * GN_PASSES
// rows; 2 = 1088 blocks of 2 rows at T = 1088, 8 = 272 blocks of 8 rows (TTK_GN_PASSES, decided by tests/diag/ddim_ab.py)
template <typename OT, int GN_PASSES>
__global__ __launch_bounds__(256) void k_gn_apply(GnApplyParams p) {
	__shared__ float s_mean[32], s_rstd[32];
	TTK_WSTAMP(p.stamps, blockIdx.x, 0);
	const int c4n = p.C / 4;                    // threads per row
	const int rpp = 256 / c4n;                  // rows per pass (C <= 1024)
	const int strip = rpp * GN_PASSES;
	const int strips = (p.Tout + strip - 1) / strip;
	const int b = blockIdx.x / strips, t0 = (blockIdx.x - b * strips) * strip;
	const int Tl = p.tlen ? p.tlen[b] : p.Tout;                                            // ragged batch: valid rows of this sequence
	const int nch = p.tlen ? (Tl + p.chunk_rows - 1) / p.chunk_rows : p.nchunks;           // ... and the chunk triples that describe them
	// the rows this thread normalises are requested FIRST: they do not depend on the statistics, and issued ahead of the chunk triples
	// the two round trips overlap (the loads retire in order, so the triples arrive with or after the rows, never before they were asked)
	const int c = (threadIdx.x % c4n) * 4, rr = threadIdx.x / c4n;
	float4 xv[GN_PASSES];
#pragma unroll
	for (int i = 0; i < GN_PASSES; ++i) {
		const int to = t0 + i * rpp + rr;
		const int ti = to < p.Tout ? (p.row_idx ? p.row_idx[to] : to) : 0;
		xv[i] = *(const float4*)(p.x + ((int64_t)b * p.T + ti) * p.C + c);
	}
	TTK_WSTAMP(p.stamps, blockIdx.x, 1);
	__shared__ unsigned pf_sink[64 * 4];
	if (p.pf)     // the next GEMM's weights into L2 (see GnApplyParams)
		l2_touch_for_next(p.pf, p.pf_bytes, p.pf_taps, __builtin_amdgcn_readfirstlane(lds_byte_addr(pf_sink) + (threadIdx.x >> 6) * 256), blockIdx.x, gridDim.x, threadIdx.x, 256);
	{   // merge: 8 lanes per group, every chunk triple requested up front (one L2 latency, not one per chunk), DPP sums
		const int g = threadIdx.x >> 3, sub = threadIdx.x & 7;
		const float* part = p.ms + ((int64_t)b * 32 + g) * p.nchunks * 3;
		float mean, rstd;
		gn_merge_triples(part, nch, sub, [](const float* q) { return *q; }, mean, rstd);
		if (sub == 0) { s_mean[g] = mean; s_rstd[g] = rstd; }
	}
	__syncthreads();
	TTK_WSTAMP(p.stamps, blockIdx.x, 2);
	const int g = c / (p.C / 32);
	const float mean = s_mean[g], rstd = s_rstd[g];
	const float4 ga = *(const float4*)(p.gamma + c), be = *(const float4*)(p.beta + c);
	float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
	if (p.scale) { sc = *(const float4*)(p.scale + (int64_t)b * p.ss_stride + c); sh = *(const float4*)(p.shift + (int64_t)b * p.ss_stride + c); }
	// fold everything into y = x * a + d per channel
	float a0, a1, a2, a3, d0, d1, d2, d3;
	gn_fold_coef(mean, rstd, ga.x, be.x, sc.x, sh.x, a0, d0); gn_fold_coef(mean, rstd, ga.y, be.y, sc.y, sh.y, a1, d1);
	gn_fold_coef(mean, rstd, ga.z, be.z, sc.z, sh.z, a2, d2); gn_fold_coef(mean, rstd, ga.w, be.w, sc.w, sh.w, a3, d3);
#pragma unroll
	for (int i = 0; i < GN_PASSES; ++i) {
		const int to = t0 + i * rpp + rr;
		if (to >= p.Tout) continue;
		float o0 = gn_fold_apply(xv[i].x, a0, d0), o1 = gn_fold_apply(xv[i].y, a1, d1), o2 = gn_fold_apply(xv[i].z, a2, d2), o3 = gn_fold_apply(xv[i].w, a3, d3);
		if (p.act == ACT_SILU) { o0 = silu_f(o0); o1 = silu_f(o1); o2 = silu_f(o2); o3 = silu_f(o3); }
		if (to >= Tl) { o0 = 0.f; o1 = 0.f; o2 = 0.f; o3 = 0.f; }      // padding rows of a ragged batch: zeros, what a k = 3 conv reads beyond a sequence's end
		OT* dst = (OT*)p.out + ((int64_t)b * p.Tout + to) * p.C + c;
		if (sizeof(OT) == 1) {
			*(unsigned*)dst = pack4_fp8(o0, o1, o2, o3);
		} else if (sizeof(OT) == 2) {
			*(uint2*)dst = pack4_16<OT>(o0, o1, o2, o3);
		} else {
			*(float4*)dst = make_float4(o0, o1, o2, o3);
		}
	}
	TTK_WSTAMP(p.stamps, blockIdx.x, 4);
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	TTK_WSTAMP(p.stamps, blockIdx.x, 5);
#endif
}

// The DDIM loop's case of k_gn_apply -- C = 1024 (one thread = 4 channels of ONE row), no row gather -- with the shape decisions at compile time.
// In-kernel stamps of every wave inside the replayed chain (tests/diag/ddim_chain.cpp, profiles/r03_ddim_chain_*.log) showed the generic kernel at
// 5.5 us for a pass whose rows stream in ~1.5: 1.2 us between a wave's first instruction and its row requests (six run-time integer divisions -- c4n,
// rows per pass, strips, the batch index, tid / c4n, tid % c4n -- and the arguments fetched in several dependent scalar round trips), and the
// statistics merge waiting for the weight touches: those were issued between the row requests and the triples, vmcnt retires in order, so the triples
// could not be used before the touched HBM lines had come back.  Here: grid (strips, batch) -- no division at all --, the arguments pinned into SGPRs
// by one batch of scalar loads, every request (rows, affine parameters, triples) issued up front, and the touches on a FIFTH wave that does nothing
// else and leaves before the barrier (a terminated wave does not take part in s_barrier), so nobody waits for them but the kernel's end -- which the
// four working waves reach later anyway.  Same arithmetic in the same order as k_gn_apply: bit-identical output.
template <typename OT, int GN_PASSES, bool GN_TOUCH_WAVE>
__global__ __launch_bounds__(GN_TOUCH_WAVE ? 320 : 256) void k_gn_apply_c1024(GnApplyParams p) {
	constexpr int C = 1024;
	__shared__ unsigned pf_sink[64 * 4];
	TTK_PIN_ARGS(TTK_S(p.x), TTK_S(p.ms), TTK_S(p.gamma), TTK_S(p.beta), TTK_S(p.scale), TTK_S(p.shift), TTK_S(p.ss_stride), TTK_S(p.T), TTK_S(p.nchunks),
				 TTK_S(p.act), TTK_S(p.out), TTK_S(p.tlen), TTK_S(p.chunk_rows), TTK_S(p.pf), TTK_S(p.pf_bytes), TTK_S(p.pf_taps));
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 0);
	const int tid = threadIdx.x;
	if (GN_TOUCH_WAVE && tid >= 256) {      // the touch wave: the following GEMM's weights into L2 (see GnApplyParams), then gone
		if (p.pf) l2_touch_for_next(p.pf, p.pf_bytes, p.pf_taps, __builtin_amdgcn_readfirstlane(lds_byte_addr(pf_sink)), blockIdx.y * gridDim.x + blockIdx.x,
									gridDim.x * gridDim.y, tid - 256, 64);
		return;
	}
	const int b = blockIdx.y, t0 = blockIdx.x * GN_PASSES;
	const int Tl = p.tlen ? p.tlen[b] : p.T;                                               // ragged batch: valid rows of this sequence
	const int nch = p.tlen ? (Tl + p.chunk_rows - 1) / p.chunk_rows : p.nchunks;           // ... and the chunk triples that describe them
	const int c = tid * 4;
	float4 xv[GN_PASSES];
#pragma unroll
	for (int i = 0; i < GN_PASSES; ++i) {
		const int to = t0 + i;
		xv[i] = *(const float4*)(p.x + ((int64_t)b * p.T + (to < p.T ? to : 0)) * C + c);
	}
	const float4 ga = *(const float4*)(p.gamma + c), be = *(const float4*)(p.beta + c);
	float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
	if (p.scale) { sc = *(const float4*)(p.scale + (int64_t)b * p.ss_stride + c); sh = *(const float4*)(p.shift + (int64_t)b * p.ss_stride + c); }
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 1);
	float mean, rstd;
	{   // merge: 8 lanes per group, every chunk triple requested up front, DPP sums (k_gn_apply's arithmetic).  With 4 channels per thread the 8 lanes
		// that merge group g are exactly the 8 threads whose channels lie in it (c >> 5 == tid >> 3), and the DPP sums leave the result in all 8: no LDS
		// hand-over, no workgroup barrier.
		const int g = tid >> 3, sub = tid & 7;
		const float* part = p.ms + ((int64_t)b * 32 + g) * p.nchunks * 3;
		gn_merge_triples(part, nch, sub, [](const float* q) { return *q; }, mean, rstd);
	}
	TTK_WSTAMPD(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 2, rstd);
	// (no fifth wave: the touches leave HERE -- every load this thread waits for has been consumed, nothing below waits on vmcnt, so they cost the
	// workgroup only their issue slots and keep it alive until they land)
	if (!GN_TOUCH_WAVE && p.pf) l2_touch_for_next(p.pf, p.pf_bytes, p.pf_taps, __builtin_amdgcn_readfirstlane(lds_byte_addr(pf_sink) + (tid >> 6) * 64), blockIdx.y * gridDim.x + blockIdx.x,
												   gridDim.x * gridDim.y, tid, 256);
	float a0, a1, a2, a3, d0, d1, d2, d3;
	gn_fold_coef(mean, rstd, ga.x, be.x, sc.x, sh.x, a0, d0); gn_fold_coef(mean, rstd, ga.y, be.y, sc.y, sh.y, a1, d1);
	gn_fold_coef(mean, rstd, ga.z, be.z, sc.z, sh.z, a2, d2); gn_fold_coef(mean, rstd, ga.w, be.w, sc.w, sh.w, a3, d3);
#pragma unroll
	for (int i = 0; i < GN_PASSES; ++i) {
		const int to = t0 + i;
		if (to >= p.T) continue;
		float o0 = gn_fold_apply(xv[i].x, a0, d0), o1 = gn_fold_apply(xv[i].y, a1, d1), o2 = gn_fold_apply(xv[i].z, a2, d2), o3 = gn_fold_apply(xv[i].w, a3, d3);
		if (p.act == ACT_SILU) { o0 = silu_f(o0); o1 = silu_f(o1); o2 = silu_f(o2); o3 = silu_f(o3); }
		if (to >= Tl) { o0 = 0.f; o1 = 0.f; o2 = 0.f; o3 = 0.f; }
		OT* dst = (OT*)p.out + ((int64_t)b * p.T + to) * C + c;
		if (sizeof(OT) == 1) *(unsigned*)dst = pack4_fp8(o0, o1, o2, o3);
		else if (sizeof(OT) == 2) *(uint2*)dst = pack4_16<OT>(o0, o1, o2, o3);
		else *(float4*)dst = make_float4(o0, o1, o2, o3);
	}
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 4);
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 5);
#endif
}

// k_gn_apply_c1024 with the rows dealt EVENLY over the chip: grid (strips, batch) with strips x batch <= 256 workgroups, strip w owning q rows + one more for the
// first `rem` strips (q = T / strips, rem = T % strips, both from the host: no division here) -- one workgroup per CU, 8 or 9 rows each at the DDIM step's T = 1088.
// The fixed 4-row strips make 544 workgroups there, 2.125 per CU: the 32 CUs that hold three set the launch (4.4 us against 3.7 us for the same launch at T = 1024,
// where 512 workgroups are two per CU; profiles/r04_ddim_chain_T1024_vs_T1088.log).  Up to GN_MAXR rows per thread, every request up front; elementwise, same
// arithmetic: bit-identical output.
template <typename OT, int GN_MAXR, int NG = 8>      // NG: groups of eight statistics chunks a lane asks for (3 serves sequences up to 1536 frames: 9 dwords per lane instead of 24)
__global__ __launch_bounds__(320) void k_gn_apply_c1024_even(GnApplyParams p, int q, int rem) {
	constexpr int C = 1024;
	__shared__ unsigned pf_sink[64 * 4];
	TTK_PIN_ARGS(TTK_S(p.x), TTK_S(p.ms), TTK_S(p.gamma), TTK_S(p.beta), TTK_S(p.scale), TTK_S(p.shift), TTK_S(p.ss_stride), TTK_S(p.T), TTK_S(p.nchunks),
				 TTK_S(p.act), TTK_S(p.out), TTK_S(p.pf), TTK_S(p.pf_bytes), TTK_S(p.pf_taps), TTK_S(q), TTK_S(rem));
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 0);
	const int tid = threadIdx.x;
	if (tid >= 256) {      // the touch wave: the following GEMM's weights into L2 (see GnApplyParams), then gone
		if (p.pf) l2_touch_for_next(p.pf, p.pf_bytes, p.pf_taps, __builtin_amdgcn_readfirstlane(lds_byte_addr(pf_sink)), blockIdx.y * gridDim.x + blockIdx.x,
									gridDim.x * gridDim.y, tid - 256, 64);
		return;
	}
	const int b = blockIdx.y, w = blockIdx.x;
	const int t0 = w * q + (w < rem ? w : rem), nrows = q + (w < rem ? 1 : 0);
	const int c = tid * 4;
	// the statistics triples are requested FIRST (round 6): the merge then runs while the rows are still on their way (see gn_load_triples)
	float cn[8], cm[8], c2[8];
	gn_load_triples_n<NG>(p.ms + ((int64_t)b * 32 + (tid >> 3)) * p.nchunks * 3, p.nchunks, tid & 7, [](const float* qq) { return *qq; }, cn, cm, c2);
	__builtin_amdgcn_sched_barrier(0);
	float4 xv[GN_MAXR];
#pragma unroll
	for (int i = 0; i < GN_MAXR; ++i) {
		// STRAIGHT (NG < 8 instantiations, GN_MAXR = the strip's row count or one more): every row request leaves unconditionally -- a strip one row short re-reads its last row --
		// so that the requests form one straight line and the wait in front of the merge can be counted: under `if (i < nrows)` the compiler cannot know how many requests
		// follow the triples and waits for (nearly) all of them
		if (NG < 8) xv[i] = *(const float4*)(p.x + ((int64_t)b * p.T + t0 + (i < nrows ? i : nrows - 1)) * C + c);
		else if (i < nrows) xv[i] = *(const float4*)(p.x + ((int64_t)b * p.T + t0 + i) * C + c);
	}
	const float4 ga = *(const float4*)(p.gamma + c), be = *(const float4*)(p.beta + c);
	float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
	if (NG < 8) {      // (straight line as well: without scale / shift the two requests re-read gamma / beta and the values are dropped)
		const float4 s1 = *(const float4*)((p.scale ? p.scale + (int64_t)b * p.ss_stride : p.gamma) + c), s2 = *(const float4*)((p.scale ? p.shift + (int64_t)b * p.ss_stride : p.beta) + c);
		if (p.scale) { sc = s1; sh = s2; }
	} else if (p.scale) { sc = *(const float4*)(p.scale + (int64_t)b * p.ss_stride + c); sh = *(const float4*)(p.shift + (int64_t)b * p.ss_stride + c); }
	__builtin_amdgcn_sched_barrier(0);
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 1);
	float mean, rstd;
	gn_merge_loaded(cn, cm, c2, mean, rstd);
	__builtin_amdgcn_sched_barrier(0);      // (nothing of the fold -- it needs scale / shift, the youngest requests -- may be scheduled into the merge)
	TTK_WSTAMPD(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 2, rstd);
	float a0, a1, a2, a3, d0, d1, d2, d3;
	gn_fold_coef(mean, rstd, ga.x, be.x, sc.x, sh.x, a0, d0); gn_fold_coef(mean, rstd, ga.y, be.y, sc.y, sh.y, a1, d1);
	gn_fold_coef(mean, rstd, ga.z, be.z, sc.z, sh.z, a2, d2); gn_fold_coef(mean, rstd, ga.w, be.w, sc.w, sh.w, a3, d3);
#pragma unroll
	for (int i = 0; i < GN_MAXR; ++i) {
		if (i >= nrows) continue;
		float o0 = gn_fold_apply(xv[i].x, a0, d0), o1 = gn_fold_apply(xv[i].y, a1, d1), o2 = gn_fold_apply(xv[i].z, a2, d2), o3 = gn_fold_apply(xv[i].w, a3, d3);
		if (p.act == ACT_SILU) { o0 = silu_f(o0); o1 = silu_f(o1); o2 = silu_f(o2); o3 = silu_f(o3); }
		OT* dst = (OT*)p.out + ((int64_t)b * p.T + t0 + i) * C + c;
		if (sizeof(OT) == 1) *(unsigned*)dst = pack4_fp8(o0, o1, o2, o3);
		else if (sizeof(OT) == 2) *(uint2*)dst = pack4_16<OT>(o0, o1, o2, o3);
		else *(float4*)dst = make_float4(o0, o1, o2, o3);
	}
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 4);
#if defined(TTK_STAMPS) && TTK_STAMPS == 2
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	TTK_WSTAMP(p.stamps, blockIdx.y * gridDim.x + blockIdx.x, 5);
#endif
}

template <int PASSES>
static void launch_gn_apply_p(int dt, const GnApplyParams& p, hipStream_t s) {
	const int strip = (256 / (p.C / 4)) * PASSES;
	const int grid = p.nb * ((p.Tout + strip - 1) / strip);
	if (p.out_f8) hipLaunchKernelGGL((k_gn_apply<f8, PASSES>), dim3(grid), dim3(256), 0, s, p);
	else if (p.out_f32 || dt == DT_F32) hipLaunchKernelGGL((k_gn_apply<float, PASSES>), dim3(grid), dim3(256), 0, s, p);
	else if (dt == DT_F16) hipLaunchKernelGGL((k_gn_apply<f16, PASSES>), dim3(grid), dim3(256), 0, s, p);
	else hipLaunchKernelGGL((k_gn_apply<bf16, PASSES>), dim3(grid), dim3(256), 0, s, p);
}
void launch_gn_apply(int dt, const GnApplyParams& p, hipStream_t s) {
	ProfScope prof(PROF_GN_APPLY, (double)p.nb * p.Tout * p.C * (4.0 + (p.out_f8 ? 1.0 : p.out_f32 ? 4.0 : dtype_size(dt))), s);
	static const int passes = [] { const char* e = getenv("TTK_GN_PASSES"); return e ? atoi(e) : 2; }();
	static const int fast = [] { const char* e = getenv("TTK_GN_FAST"); return e ? atoi(e) : 1; }();      // 0: the generic kernel everywhere
	if (fast && p.C == 1024 && !p.row_idx && p.Tout == p.T && passes == 2 && p.nb <= 65535) {      // the DDIM loop's launches: k_gn_apply_c1024
		// rows per thread: with 2 the 1088 five-wave workgroups of a DDIM step do not fit the chip at once (two waves of each land on one SIMD: 4 per CU, 1024 slots) and
		// the last 64 start 3 us late; with 4 all 544 are resident within 0.5 us (tests/diag/ddim_chain: 4.5 / 5.1 / 5.2 -> 4.2 / 4.7 / 4.8 us per launch)
		// rows dealt evenly, one workgroup per CU (k_gn_apply_c1024_even), when a sequence's rows split into <= 256 / nb strips of 4 .. 18 rows and the batch is not ragged
		static const int even = [] { const char* e = getenv("TTK_GN_EVEN"); return e ? atoi(e) : 1; }();
		if (even && !p.tlen && p.nb >= 1 && p.nb <= 64) {
			const int strips = 256 / p.nb, q = p.T / strips, rem = p.T % strips;
			if (q >= 4 && q + (rem ? 1 : 0) <= 18) {
				const dim3 grid(strips, p.nb);
#define GN_EVEN(OT) do { const int mr = q + (rem ? 1 : 0); \
						if (p.nchunks <= 24 && mr <= 9) hipLaunchKernelGGL((k_gn_apply_c1024_even<OT, 9, 3>), grid, dim3(320), 0, s, p, q, rem); \
						else if (p.nchunks <= 24 && mr <= 12) hipLaunchKernelGGL((k_gn_apply_c1024_even<OT, 12, 3>), grid, dim3(320), 0, s, p, q, rem); \
						else hipLaunchKernelGGL((k_gn_apply_c1024_even<OT, 18, 8>), grid, dim3(320), 0, s, p, q, rem); } while (0)
				if (p.out_f8) GN_EVEN(f8);
				else if (p.out_f32 || dt == DT_F32) GN_EVEN(float);
				else if (dt == DT_F16) GN_EVEN(f16);
				else GN_EVEN(bf16);
#undef GN_EVEN
				return;
			}
		}
		static const int cp = [] { const char* e = getenv("TTK_GN_C1024_PASSES"); return e && atoi(e) == 2 ? 2 : 4; }();
		const dim3 grid((p.T + cp - 1) / cp, p.nb);
		// the weight touches on a fifth wave (default) or, TTK_GN_TOUCH_WAVE=0, issued by the four working waves once their own loads are consumed:
		// 122.8 - 123.0 against 123.5 - 123.7 us per layer in tests/diag/ddim_chain (both far ahead of touches issued between the row requests and the triples)
		static const int tw = [] { const char* e = getenv("TTK_GN_TOUCH_WAVE"); return e ? atoi(e) : 1; }();
#define GN_GO(OT) do { if (cp == 4) { if (tw) hipLaunchKernelGGL((k_gn_apply_c1024<OT, 4, true>), grid, dim3(320), 0, s, p); else hipLaunchKernelGGL((k_gn_apply_c1024<OT, 4, false>), grid, dim3(256), 0, s, p); } \
	else if (tw) hipLaunchKernelGGL((k_gn_apply_c1024<OT, 2, true>), grid, dim3(320), 0, s, p); else hipLaunchKernelGGL((k_gn_apply_c1024<OT, 2, false>), grid, dim3(256), 0, s, p); } while (0)
		if (p.out_f8) GN_GO(f8);
		else if (p.out_f32 || dt == DT_F32) GN_GO(float);
		else if (dt == DT_F16) GN_GO(f16);
		else GN_GO(bf16);
#undef GN_GO
		return;
	}
	// few rows in all (short clips, the latent conditioner): keep the strips small so the launch still spreads over the chip
	const int rows = p.nb * p.Tout * (p.C / 4) / 256;
	if (passes >= 8 && rows >= 8 * 256) launch_gn_apply_p<8>(dt, p, s);
	else if (passes >= 4 && rows >= 4 * 256) launch_gn_apply_p<4>(dt, p, s);
	else if (passes == 1) launch_gn_apply_p<1>(dt, p, s);
	else launch_gn_apply_p<2>(dt, p, s);
}

}  // namespace ttk
