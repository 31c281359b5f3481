// LayerNorm (GPT-2 ln_1/ln_2/ln_f + final_norm) and GroupNorm32 (+SiLU, +timestep scale/shift) -- HBM/L2-bound
// row passes over f32 residual streams that emit the T-typed operand of the next MFMA GEMM.
//   LayerNorm:   HF:models/gpt2/modeling_gpt2.py:246-310 (eps 1e-5), unified_voice.py:442,519
//   GroupNorm32: /root/reference/tortoise_tts/models/arch_utils.py:24-44 (32 groups, float math, eps 1e-5),
//                used by ResBlock diffusion.py:1338-1372 and AttentionBlock arch_utils.py:163,186
// Internal layout is channels-last [nb][T][C], so a group is (T rows) x (C/32 contiguous channels).
#include "ttk_common.h"
#include "ttk_kernels.h"

namespace ttk {

// one wave per row; d <= 4096
template <typename OT>
__global__ void k_layernorm(const float* x, int64_t ldx, int rows, int d, const float* g1, const float* b1,
							const float* g2, const float* b2, OT* out, int64_t ldo) {
	const int lane = threadIdx.x & 63;
	const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
	if (row >= rows) return;
	const int nchunk = d / 4;
	float4 v[16];
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const int c = lane + 64 * i;
		v[i] = c < nchunk ? *(const float4*)(x + (int64_t)row * ldx + 4 * c) : make_float4(0, 0, 0, 0);
	}
	for (int pass = 0; pass < (g2 ? 2 : 1); ++pass) {
		const float* g = pass ? g2 : g1;
		const float* b = pass ? b2 : b1;
		float sum = 0.f;
#pragma unroll
		for (int i = 0; i < 16; ++i) sum += v[i].x + v[i].y + v[i].z + v[i].w;
		const float mean = wave_sum(sum) / (float)d;
		float sq = 0.f;
#pragma unroll
		for (int i = 0; i < 16; ++i)
			if (lane + 64 * i < nchunk) {
				const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
				sq += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
			}
		const float rstd = rsqrtf(wave_sum(sq) / (float)d + 1e-5f);
#pragma unroll
		for (int i = 0; i < 16; ++i) {
			const int c = lane + 64 * i;
			if (c < nchunk) {
				const float4 gg = *(const float4*)(g + 4 * c), bb = *(const float4*)(b + 4 * c);
				v[i].x = (v[i].x - mean) * rstd * gg.x + bb.x;
				v[i].y = (v[i].y - mean) * rstd * gg.y + bb.y;
				v[i].z = (v[i].z - mean) * rstd * gg.z + bb.z;
				v[i].w = (v[i].w - mean) * rstd * gg.w + bb.w;
			}
		}
	}
#pragma unroll
	for (int i = 0; i < 16; ++i) {
		const int c = lane + 64 * i;
		if (c < nchunk) {
			OT* o = out + (int64_t)row * ldo + 4 * c;
			o[0] = (OT)v[i].x; o[1] = (OT)v[i].y; o[2] = (OT)v[i].z; o[3] = (OT)v[i].w;
		}
	}
}

void launch_layernorm(int dt, const float* x, int64_t ldx, int rows, int d, const float* g1, const float* b1,
					  const float* g2, const float* b2, void* out, int64_t ldo, int out_f32, hipStream_t s) {
	const int grid = (rows + 3) / 4;
	if (out_f32 || dt == DT_F32)
		hipLaunchKernelGGL((k_layernorm<float>), dim3(grid), dim3(256), 0, s, x, ldx, rows, d, g1, b1, g2, b2, (float*)out, ldo);
	else
		hipLaunchKernelGGL((k_layernorm<bf16>), dim3(grid), dim3(256), 0, s, x, ldx, rows, d, g1, b1, g2, b2, (bf16*)out, ldo);
}

// ---------------------------------------------------------------- GroupNorm32
// stats: grid (32, nb), 256 threads; two passes over the (T x cpg) slab (mean, then centred sum of squares).
__global__ void k_gn_stats(const float* x, int T, int C, float* ms) {
	const int g = blockIdx.x, b = blockIdx.y, cpg = C / 32;
	const float* base = x + (int64_t)b * T * C + g * cpg;
	const int n = T * cpg;
	__shared__ float sh[4];
	float sum = 0.f;
	for (int i = threadIdx.x; i < n; i += 256) { const int t = i / cpg, c = i - t * cpg; sum += base[(int64_t)t * C + c]; }
	sum = wave_sum(sum);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sum;
	__syncthreads();
	const float mean = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)n;
	__syncthreads();
	float sq = 0.f;
	for (int i = threadIdx.x; i < n; i += 256) { const int t = i / cpg, c = i - t * cpg; const float d = base[(int64_t)t * C + c] - mean; sq += d * d; }
	sq = wave_sum(sq);
	if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = sq;
	__syncthreads();
	if (threadIdx.x == 0) {
		const float var = (sh[0] + sh[1] + sh[2] + sh[3]) / (float)n;
		ms[(b * 32 + g) * 2 + 0] = mean;
		ms[(b * 32 + g) * 2 + 1] = rsqrtf(var + 1e-5f);
	}
}

void launch_gn_stats(const float* x, int nb, int T, int C, float* ms, hipStream_t s) {
	hipLaunchKernelGGL(k_gn_stats, dim3(32, nb), dim3(256), 0, s, x, T, C, ms);
}

// apply: one thread per 4 channels of one output row
template <typename OT>
__global__ void k_gn_apply(GnApplyParams p) {
	const int c4 = p.C / 4;
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int64_t total = (int64_t)p.nb * p.Tout * c4;
	if (idx >= total) return;
	const int c = (int)(idx % c4) * 4;
	const int64_t orow = idx / c4;
	const int b = (int)(orow / p.Tout), to = (int)(orow - (int64_t)b * p.Tout);
	const int ti = p.row_idx ? p.row_idx[to] : to;
	const float4 xv = *(const float4*)(p.x + ((int64_t)b * p.T + ti) * p.C + c);
	const int cpg = p.C / 32;
	float in[4] = {xv.x, xv.y, xv.z, xv.w}, o[4];
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int cc = c + j, g = cc / cpg;
		const float mean = p.ms[(b * 32 + g) * 2], rstd = p.ms[(b * 32 + g) * 2 + 1];
		float v = (in[j] - mean) * rstd * p.gamma[cc] + p.beta[cc];
		if (p.scale) v = v * (1.0f + p.scale[(int64_t)b * p.ss_stride + cc]) + p.shift[(int64_t)b * p.ss_stride + cc];
		o[j] = apply_act(v, p.act);
	}
	OT* dst = (OT*)p.out + orow * p.C + c;
	dst[0] = (OT)o[0]; dst[1] = (OT)o[1]; dst[2] = (OT)o[2]; dst[3] = (OT)o[3];
}

void launch_gn_apply(int dt, const GnApplyParams& p, hipStream_t s) {
	const int64_t total = (int64_t)p.nb * p.Tout * (p.C / 4);
	const int grid = (int)((total + 255) / 256);
	if (p.out_f32 || dt == DT_F32) hipLaunchKernelGGL((k_gn_apply<float>), dim3(grid), dim3(256), 0, s, p);
	else hipLaunchKernelGGL((k_gn_apply<bf16>), dim3(grid), dim3(256), 0, s, p);
}

}  // namespace ttk
