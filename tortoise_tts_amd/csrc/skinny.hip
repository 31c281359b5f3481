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

template <typename T, int MT, bool LN>
__global__ void k_skinny(SkinnyParams p) {
	typedef typename Frag<T>::type FragT;
	constexpr int ES = sizeof(T);
	extern __shared__ __attribute__((aligned(16))) char smem[];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
	const int nt = blockIdx.x;
	const int KS = p.K / 32;
	const int RS = p.K * ES + 16;                       // padded LDS row stride (bytes), LN mode only
	char* a_lds = smem;
	float* red = (float*)(smem + (LN ? 16 * MT * RS : 0));

	if (LN) {
		// rows distributed over waves; each lane owns float4 chunks lane, lane+64, ... of the row (K <= 2048)
		for (int r = wave; r < 16 * MT; r += nw) {
			float4 v[8];
			const int nchunk = p.K / 4;
			const bool live = r < p.M;
			float sum = 0.f;
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int c = lane + 64 * i;
				v[i] = (live && c < nchunk) ? *(const float4*)(p.x + (int64_t)r * p.ldx + 4 * c) : make_float4(0, 0, 0, 0);
				sum += v[i].x + v[i].y + v[i].z + v[i].w;
			}
			const float* gs[2] = {p.g1, p.g2};
			const float* bs[2] = {p.b1, p.b2};
			for (int pass = 0; pass < p.ln_count; ++pass) {
				if (pass > 0) {
					sum = 0.f;
#pragma unroll
					for (int i = 0; i < 8; ++i) sum += v[i].x + v[i].y + v[i].z + v[i].w;
				}
				const float mean = wave_sum(sum) / (float)p.K;
				float sq = 0.f;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					if (lane + 64 * i < nchunk) {
						const float a = v[i].x - mean, b = v[i].y - mean, c2 = v[i].z - mean, d = v[i].w - mean;
						sq += a * a + b * b + c2 * c2 + d * d;
					}
				}
				const float rstd = rsqrtf(wave_sum(sq) / (float)p.K + 1e-5f);
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const int c = lane + 64 * i;
					if (c < nchunk) {
						const float4 g = *(const float4*)(gs[pass] + 4 * c), b = *(const float4*)(bs[pass] + 4 * c);
						v[i].x = (v[i].x - mean) * rstd * g.x + b.x;
						v[i].y = (v[i].y - mean) * rstd * g.y + b.y;
						v[i].z = (v[i].z - mean) * rstd * g.z + b.z;
						v[i].w = (v[i].w - mean) * rstd * g.w + b.w;
					}
				}
			}
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const int c = lane + 64 * i;
				if (c < nchunk) {
					T* dst = (T*)(a_lds + r * RS) + 4 * c;
					dst[0] = cvt<T>(live ? v[i].x : 0.f);
					dst[1] = cvt<T>(live ? v[i].y : 0.f);
					dst[2] = cvt<T>(live ? v[i].z : 0.f);
					dst[3] = cvt<T>(live ? v[i].w : 0.f);
					if (p.ln_out && nt == 0 && live) *(float4*)(p.ln_out + (int64_t)r * p.K + 4 * c) = v[i];
				}
			}
		}
		__syncthreads();
	}

	// ---- stream this wave's K slice
	const int ks0 = (KS * wave) / nw, ks1 = (KS * (wave + 1)) / nw;
	f32x4 acc[MT];
#pragma unroll
	for (int mt = 0; mt < MT; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
	const FragT* wp = (const FragT*)p.Wp + ((int64_t)nt * KS) * 64 + lane;
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
	constexpr int UN = 8;   // weight fragments in flight per wave (8 x 1 KiB bf16)
	int ks = ks0;
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
		const size_t lds = (size_t)16 * MT * (p.K * sizeof(T) + 16) + red;
		if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)k_skinny<T, MT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
		hipLaunchKernelGGL((k_skinny<T, MT, true>), dim3(grid), dim3(64 * waves), lds, s, p);
	} else {
		hipLaunchKernelGGL((k_skinny<T, MT, false>), dim3(grid), dim3(64 * waves), red, s, p);
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
