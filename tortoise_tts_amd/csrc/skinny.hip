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
#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

// KC = float4 chunks of a row per lane in the LayerNorm prologue (K <= 256 * KC); LN kernels run <= 8 waves (2 per SIMD,
// 256 VGPRs), plain ones up to 16.
template <typename T, int MT, bool LN, int KC>
__global__ __launch_bounds__(LN ? 512 : 1024) void k_skinny(SkinnyParams p) {
	typedef typename Frag<T>::type FragT;
	constexpr int ES = sizeof(T);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
	const int nt = blockIdx.x;
	const int KS = p.K / 32;
	const int RS = p.K * ES + 16;                       // padded LDS row stride (bytes), LN mode only
	char* a_lds = smem;
	float* red = (float*)(smem + (LN ? 16 * MT * RS : 0));

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
			for (int i = 0; i < KC; ++i) {
				const int c = lane + 64 * i;
				v[j][i] = (live && c < nchunk) ? *(const float4*)(p.x + (int64_t)r * p.ldx + 4 * c) : make_float4(0, 0, 0, 0);
			}
		}
	};
	auto ln_finish = [&](int r0) {
#pragma unroll
		for (int j = 0; j < RP; ++j) {
			const int r = r0 + j * nw;
			if (r >= 16 * MT) continue;
			const bool live = r < p.M;
			for (int pass = 0; pass < p.ln_count; ++pass) {
				const float* gp = pass ? p.g2 : p.g1;
				const float* bp = pass ? p.b2 : p.b1;
				float sum = 0.f;
#pragma unroll
				for (int i = 0; i < KC; ++i) sum += v[j][i].x + v[j][i].y + v[j][i].z + v[j][i].w;
				const float mean = wave_sum(sum) / (float)p.K;
				float sq = 0.f;
#pragma unroll
				for (int i = 0; i < KC; ++i) {
					if (lane + 64 * i < nchunk) {
						const float a0 = v[j][i].x - mean, a1 = v[j][i].y - mean, a2 = v[j][i].z - mean, a3 = v[j][i].w - mean;
						sq += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
					}
				}
				const float rstd = rsqrtf(wave_sum(sq) / (float)p.K + 1e-5f);
#pragma unroll
				for (int i = 0; i < KC; ++i) {
					const int c = lane + 64 * i;
					if (c < nchunk) {
						const float4 g = *(const float4*)(gp + 4 * c), b = *(const float4*)(bp + 4 * c);
						v[j][i].x = (v[j][i].x - mean) * rstd * g.x + b.x;
						v[j][i].y = (v[j][i].y - mean) * rstd * g.y + b.y;
						v[j][i].z = (v[j][i].z - mean) * rstd * g.z + b.z;
						v[j][i].w = (v[j][i].w - mean) * rstd * g.w + b.w;
					}
				}
			}
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
	if (LN) ln_load(wave);

	// ---- this wave's K slice of the weights; the first PRE fragments are requested now
	const int ks0 = (KS * wave) / nw, ks1 = (KS * (wave + 1)) / nw;
	const FragT* wp = (const FragT*)p.Wp + ((int64_t)nt * KS) * 64 + lane;
	constexpr int PRE = 8;   // 8 x 1 KiB (bf16) in flight per wave
	FragT bpre[PRE];
	const int npre = min(ks1 - ks0, PRE);
#pragma unroll
	for (int u = 0; u < PRE; ++u)
		if (u < npre) bpre[u] = __builtin_nontemporal_load(wp + (int64_t)(ks0 + u) * 64);

	if (LN) {
		ln_finish(wave);
		for (int r0 = wave + RP * nw; r0 < 16 * MT; r0 += RP * nw) { ln_load(r0); ln_finish(r0); }
		__syncthreads();
	}

	f32x4 acc[MT];
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
	const int arow = lane & 15, ag = lane >> 4;
	auto load_a = [&](int mt, int ks) -> FragT {
		if (LN) {
			union { FragT v; uint4 q[ES / 2]; } u;
			const char* src = a_lds + (mt * 16 + arow) * RS + (32 * ks + 8 * ag) * ES;
#pragma unroll
			for (int f = 0; f < ES / 2; ++f) u.q[f] = *(const uint4*)(src + 16 * f);
			return u.v;
		} else {
			int row = mt * 16 + arow;
			row = row < p.M ? row : p.M - 1;
			return *(const FragT*)((const T*)p.a + (int64_t)row * p.lda + 32 * ks + 8 * ag);
		}
	};
#pragma unroll
	for (int u = 0; u < PRE; ++u)
		if (u < npre) {
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) acc[mt] = mma16<T>(load_a(mt, ks0 + u), bpre[u], acc[mt]);
		}
	constexpr int UN = 8;
	int ks = ks0 + npre;
	for (; ks + UN <= ks1; ks += UN) {
		FragT b[UN];
#pragma unroll
		for (int u = 0; u < UN; ++u) b[u] = __builtin_nontemporal_load(wp + (int64_t)(ks + u) * 64);
#pragma unroll
		for (int u = 0; u < UN; ++u)
#pragma unroll
			for (int mt = 0; mt < MT; ++mt) acc[mt] = mma16<T>(load_a(mt, ks + u), b[u], acc[mt]);
	}
	for (; ks < ks1; ++ks) {
		const FragT b = __builtin_nontemporal_load(wp + (int64_t)ks * 64);
#pragma unroll
		for (int mt = 0; mt < MT; ++mt) acc[mt] = mma16<T>(load_a(mt, ks), b, acc[mt]);
	}

	// ---- cross-wave reduction through LDS, then epilogue by the first 256 threads
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) *(f32x4*)(red + ((wave * MT + mt) * 64 + lane) * 4) = acc[mt];
	__syncthreads();
	if (tid >= 256) return;
	const int l2 = tid & 63, r = tid >> 6;
	const int n = nt * 16 + (l2 & 15);
	if (n >= p.N) return;
	const float bias = p.bias ? p.bias[n] : 0.f;
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) {
		const int m = mt * 16 + 4 * (l2 >> 4) + r;
		if (m >= p.M) continue;
		float v = bias;
		for (int w = 0; w < nw; ++w) v += red[((w * MT + mt) * 64 + l2) * 4 + r];
		if (p.mode == SK_STORE_F32) {
			p.out_f32[(int64_t)m * p.ldc + n] = v;
		} else if (p.mode == SK_RESIDUAL) {
			p.out_f32[(int64_t)m * p.ldc + n] += v;
		} else if (p.mode == SK_ACT_T) {
			((T*)p.out_T)[(int64_t)m * p.N + n] = cvt<T>(apply_act(v, p.act));
		} else {   // SK_QKV
			const int d = p.N / 3;
			const int which = n / d, c = n - which * d;
			if (which == 0) {
				p.qbuf[(int64_t)m * d + c] = v * p.q_scale;
			} else {
				const int h = c >> 6, dd = c & 63;
				T* cache = (T*)(which == 1 ? p.kcache : p.vcache);
				const int pos = *p.d_pos;
				if (pos < p.max_ctx) cache[(((int64_t)m * p.H + h) * p.max_ctx + pos) * 64 + dd] = cvt<T>(v);   // guard: never write past the cache
			}
		}
	}
}

template <typename T, int MT>
static void launch_skinny_mt(const SkinnyParams& p, int waves, hipStream_t s) {
	const int grid = (p.N + 15) / 16;
	const size_t red = (size_t)waves * MT * 64 * 4 * sizeof(float);
	if (p.ln_count > 0) {
		if (waves > 8) waves = 8;
		const size_t lds = (size_t)16 * MT * (p.K * sizeof(T) + 16) + (size_t)waves * MT * 64 * 4 * sizeof(float);
		if (p.K <= 1024) {
			if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_skinny<T, MT, true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			hipLaunchKernelGGL((k_skinny<T, MT, true, 4>), dim3(grid), dim3(64 * waves), lds, s, p);
		} else {
			if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_skinny<T, MT, true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
			hipLaunchKernelGGL((k_skinny<T, MT, true, 8>), dim3(grid), dim3(64 * waves), lds, s, p);
		}
	} else {
		hipLaunchKernelGGL((k_skinny<T, MT, false, 1>), dim3(grid), dim3(64 * waves), red, s, p);
	}
}

template <typename T>
static void launch_skinny_t(const SkinnyParams& p, int waves, hipStream_t s) {
	if (p.M <= 16) launch_skinny_mt<T, 1>(p, waves, s);
	else if (p.M <= 32) launch_skinny_mt<T, 2>(p, waves, s);
	else launch_skinny_mt<T, 4>(p, waves, s);
}

void launch_skinny(int dt, const SkinnyParams& p, int waves, hipStream_t s) {
	if (waves < 4) waves = 4;
	// algorithmic bytes: the weight matrix once + bias + the M activation rows in and out
	ProfScope prof(PROF_SKINNY, (double)p.N * p.K * dtype_size(dt) + 4.0 * p.N + 4.0 * p.M * p.K + 4.0 * p.M * p.N, s);
	if (dt == DT_BF16) launch_skinny_t<bf16>(p, waves, s);
	else launch_skinny_t<float>(p, waves, s);
}

}  // namespace ttk
