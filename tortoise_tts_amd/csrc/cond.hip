// ttk_cond: the two conditioning-latent encoders behind the C ABI of include/ttk.h.
//   AR:        ConditioningEncoder       /root/reference/tortoise_tts/models/unified_voice.py:269-293 (1x1 conv + 6 AttentionBlocks, position 0)
//   diffusion: contextual_embedder       /root/reference/tortoise_tts/models/diffusion.py:1441-1447 (two stride-2 k=3 convs + 5 AttentionBlocks
//              of 2048 channels / 16 heads = head width 128, relative position bias), mean over positions :1477-1485
//   block:     AttentionBlock._forward   /root/reference/tortoise_tts/models/arch_utils.py:136-190 (+ QKVAttentionLegacy :59-94)
// One-off per voice (a few hundred mel frames), so this file reuses the dense GEMM and, for head width 64, the MFMA attention of
// the hot path, and adds plain kernels for what those do not cover: the strided stem (im2col), GroupNorm for any channels-per-group,
// attention for head width 128 (one wave per query row), and the pooling.  Layout: channels-last [b*T rows][C], f32 residual stream.
#include <float.h>
#include <stdlib.h>

#include "ttk_common.h"
#include "ttk_host.h"

using namespace ttk;

namespace {

// rows of a strided 'same'-style convolution as GEMM operand: out[(b, t)][c * taps + j] = src[b][c][t * stride + j - pad] (0 outside),
// the k order of `weight.reshape(N, Cin * taps)`.  src is f32 with element strides (sb, sc, st): channels-first mel or channels-last rows.
template <typename T>
__global__ __launch_bounds__(256) void k_cond_im2col(const float* __restrict__ src, int64_t sb, int64_t sc, int64_t st, int Cin, int Tin, int Tout,
													 int taps, int stride, int pad, int Kpad, int64_t total, T* __restrict__ out) {
	const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= total) return;
	const int k = (int)(i % Kpad);
	const int64_t row = i / Kpad;
	const int t = (int)(row % Tout);
	const int64_t b = row / Tout;
	float v = 0.f;
	if (k < Cin * taps) {
		const int c = k / taps, j = k - c * taps, ts = t * stride + j - pad;
		if (ts >= 0 && ts < Tin) v = src[b * sb + c * sc + ts * st];
	}
	out[i] = cvt<T>(v);
}

__device__ __forceinline__ float block_sum256(float v, float* red) {
	v = wave_sum(v);
	__syncthreads();
	if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
	__syncthreads();
	return (red[0] + red[1]) + (red[2] + red[3]);
}

// GroupNorm of one (batch element, group) per workgroup: two-pass mean / biased variance in f32 (F.group_norm, eps 1e-5), any C / groups
template <typename OT>
__global__ __launch_bounds__(256) void k_cond_gn(const float* __restrict__ x, int T, int C, int groups, const float* __restrict__ gamma,
												 const float* __restrict__ beta, OT* __restrict__ out) {
	__shared__ float red[4];
	const int g = blockIdx.x, cpg = C / groups, n = T * cpg;
	const int64_t off = (int64_t)blockIdx.y * T * C + g * cpg;
	float s = 0.f;
	for (int e = threadIdx.x; e < n; e += 256) { const int t = e / cpg, c = e - t * cpg; s += x[off + (int64_t)t * C + c]; }
	const float mean = block_sum256(s, red) / (float)n;
	float q = 0.f;
	for (int e = threadIdx.x; e < n; e += 256) { const int t = e / cpg, c = e - t * cpg; const float d = x[off + (int64_t)t * C + c] - mean; q += d * d; }
	const float rstd = rsqrtf(block_sum256(q, red) / (float)n + 1e-5f);
	for (int e = threadIdx.x; e < n; e += 256) {
		const int t = e / cpg, c = e - t * cpg;
		const int64_t i = off + (int64_t)t * C + c;
		out[i] = cvt<OT>((x[i] - mean) * rstd * gamma[g * cpg + c] + beta[g * cpg + c]);
	}
}

// Attention for head widths the MFMA kernel does not cover.  One wave per query row: lanes split the keys for q.k (+ bias) and the
// softmax statistics, scores live in the wave's LDS strip, then lanes split the head dims for P.V.  f32 throughout.
template <typename T, int HD>
__global__ __launch_bounds__(256) void k_attn_rowwave(AttnParams p) {
	typedef typename Frag<T>::type FragT;
	extern __shared__ float smem[];
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int q = blockIdx.x * 4 + wave, h = blockIdx.y, b = blockIdx.z;
	if (q >= p.T) return;                               // no workgroup barrier below: a wave may leave alone
	float* sc = smem + (int64_t)wave * p.T;
	float* qs = smem + (int64_t)4 * p.T + wave * HD;
	const T* base = (const T*)p.qkv + (int64_t)b * p.T * p.ld + h * p.head_stride;
	for (int d = lane; d < HD; d += 64) qs[d] = (float)base[(int64_t)q * p.ld + p.q_off + d] * p.scale;
	__builtin_amdgcn_wave_barrier();
	float m = -FLT_MAX;
	for (int j = lane; j < p.T; j += 64) {
		const T* kp = base + (int64_t)j * p.ld + p.k_off;
		float s = 0.f;
#pragma unroll 4
		for (int d = 0; d < HD; d += 8) {
			const FragT kf = *(const FragT*)(kp + d);
#pragma unroll
			for (int e = 0; e < 8; ++e) s += qs[d + e] * (float)kf[e];
		}
		if (p.bias) { int rel = j - q; rel = rel < -64 ? -64 : (rel > 64 ? 64 : rel); s += p.bias[h * 129 + rel + 64]; }
		sc[j] = s;
		m = fmaxf(m, s);
	}
	m = wave_max(m);
	float l = 0.f;
	for (int j = lane; j < p.T; j += 64) { const float e = __expf(sc[j] - m); sc[j] = e; l += e; }
	l = wave_sum(l);
	__builtin_amdgcn_wave_barrier();
	constexpr int DPL = HD / 64;
	float o[DPL];
#pragma unroll
	for (int e = 0; e < DPL; ++e) o[e] = 0.f;
	const T* vp = base + p.v_off + lane * DPL;
	for (int j = 0; j < p.T; ++j) {
		const float pj = sc[j];
#pragma unroll
		for (int e = 0; e < DPL; ++e) o[e] += pj * (float)vp[(int64_t)j * p.ld + e];
	}
	T* op = (T*)p.out + ((int64_t)b * p.T + q) * p.ldo + h * HD + lane * DPL;
	const float inv = 1.f / l;
#pragma unroll
	for (int e = 0; e < DPL; ++e) op[e] = cvt<T>(o[e] * inv);
}

// out[b][c] = mean over the T rows (mode 1) or row 0 (mode 0) of x f32 [b][T][C]
__global__ __launch_bounds__(256) void k_cond_pool(const float* __restrict__ x, int T, int C, int mode, float* __restrict__ out) {
	const int c = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
	if (c >= C) return;
	const float* base = x + (int64_t)b * T * C + c;
	float s = base[0];
	if (mode) {
		for (int t = 1; t < T; ++t) s += base[(int64_t)t * C];
		s /= (float)T;
	}
	out[(int64_t)b * C + c] = s;
}

struct Block { float *gn_g = nullptr, *gn_b = nullptr, *relbias = nullptr; Mat qkv, proj; };

}  // namespace

struct ttk_cond {
	ttk_cond_config cfg;
	int dt, groups, hd;
	size_t es;
	Arena arena;
	Mat stem0, stem1;
	std::vector<Block> blocks;
	WsBuf ws;
};

namespace {

int gn_groups(int channels) {    // arch_utils.py:27-44
	int g = 32;
	if (channels <= 16) g = 8;
	else if (channels <= 64) g = 16;
	while (channels % g != 0) g /= 2;
	return g;
}

void gemm_plain(int dt, const void* A, int64_t lda, const Mat& w, int M, const float* residual, void* C, int out_f32, hipStream_t s) {
	GemmParams g = {};
	g.nseg = 1;
	g.seg[0] = {A, lda, 0, 0};
	g.W = w.w; g.ldw = w.Kpad; g.M = M; g.N = w.N; g.K = w.Kpad; g.bias = w.bias;
	g.residual = residual; g.ldr = w.N; g.C = C; g.ldc = w.N; g.out_f32 = out_f32;
	launch_gemm(dt, g, s);
}

template <typename T>
int encode_t(ttk_cond* h, const float* mel, int b, int Tf, float* out, hipStream_t s) {
	const ttk_cond_config& c = h->cfg;
	const int C = c.channels, dt = h->dt;
	const int T1 = c.stem == TTK_COND_STEM_DOWN4 ? (Tf - 1) / 2 + 1 : Tf;
	const int T2 = c.stem == TTK_COND_STEM_DOWN4 ? (T1 - 1) / 2 + 1 : Tf;
	const int64_t rows = (int64_t)b * T2;
	TTK_REQUIRE(h->hd == 64 || T2 <= 3584, TTK_E_ARG, "ttk_cond_encode: %d positions exceed the row-wave attention's LDS strip (3584)", T2);
	// workspace: x f32 [rows][C] | a T [rows][C] | qkv T [rows][3C] | ao T [rows][C]; the stem's operands alias a/qkv/ao (free until block 0)
	const size_t im0 = (size_t)b * T1 * h->stem0.Kpad * h->es, mid = c.stem == TTK_COND_STEM_DOWN4 ? (size_t)b * T1 * (C / 2) * 4 : 0;
	const size_t im1 = c.stem == TTK_COND_STEM_DOWN4 ? (size_t)rows * h->stem1.Kpad * h->es : 0;
	const size_t xbytes = (size_t)rows * C * 4, blk = (size_t)rows * C * 5 * h->es;
	const size_t scratch = std::max(blk, ((im0 + 255) / 256 + (mid + 255) / 256 + (im1 + 255) / 256) * 256);
	TTK_TRY(h->ws.reserve(xbytes + scratch + 1024));
	char* base = (char*)h->ws.p;
	float* x = (float*)base;
	char* sp = base + (xbytes + 255) / 256 * 256;
	{
		T* a0 = (T*)sp;
		const int64_t total = (int64_t)b * T1 * h->stem0.Kpad;
		if (c.stem == TTK_COND_STEM_DOWN4) {
			float* y = (float*)(sp + (im0 + 255) / 256 * 256);
			T* a1 = (T*)((char*)y + (mid + 255) / 256 * 256);
			hipLaunchKernelGGL((k_cond_im2col<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mel, (int64_t)c.in_channels * Tf, (int64_t)Tf, (int64_t)1,
							   c.in_channels, Tf, T1, 3, 2, 1, h->stem0.Kpad, total, a0);
			gemm_plain(dt, a0, h->stem0.Kpad, h->stem0, b * T1, nullptr, y, 1, s);
			const int64_t total1 = rows * h->stem1.Kpad;
			hipLaunchKernelGGL((k_cond_im2col<T>), dim3((unsigned)((total1 + 255) / 256)), dim3(256), 0, s, y, (int64_t)T1 * (C / 2), (int64_t)1, (int64_t)(C / 2),
							   C / 2, T1, T2, 3, 2, 1, h->stem1.Kpad, total1, a1);
			gemm_plain(dt, a1, h->stem1.Kpad, h->stem1, (int)rows, nullptr, x, 1, s);
		} else {
			hipLaunchKernelGGL((k_cond_im2col<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mel, (int64_t)c.in_channels * Tf, (int64_t)Tf, (int64_t)1,
							   c.in_channels, Tf, Tf, 1, 1, 0, h->stem0.Kpad, total, a0);
			gemm_plain(dt, a0, h->stem0.Kpad, h->stem0, (int)rows, nullptr, x, 1, s);
		}
	}
	T* a = (T*)sp;
	T* qkv = a + rows * C;
	T* ao = qkv + rows * 3 * C;
	for (const Block& B : h->blocks) {
		hipLaunchKernelGGL((k_cond_gn<T>), dim3(h->groups, b), dim3(256), 0, s, x, T2, C, h->groups, B.gn_g, B.gn_b, a);
		gemm_plain(dt, a, C, B.qkv, (int)rows, nullptr, qkv, 0, s);
		AttnParams ap = {};
		ap.qkv = qkv; ap.ld = 3 * C; ap.q_off = 0; ap.k_off = h->hd; ap.v_off = 2 * h->hd; ap.head_stride = 3 * h->hd;   // head-major [H][3][hd], arch_utils.py:79
		ap.out = ao; ap.ldo = C; ap.nb = b; ap.T = T2; ap.H = c.num_heads; ap.causal = 0; ap.bias = B.relbias;
		ap.scale = 1.f / sqrtf((float)h->hd);                                      // (q * hd^-1/4) . (k * hd^-1/4)
		if (h->hd == 64) launch_attn_fwd(dt, ap, s);
		else hipLaunchKernelGGL((k_attn_rowwave<T, 128>), dim3((T2 + 3) / 4, c.num_heads, b), dim3(256), (size_t)(4 * T2 + 4 * 128) * 4, s, ap);
		gemm_plain(dt, ao, C, B.proj, (int)rows, x, x, 1, s);
	}
	hipLaunchKernelGGL(k_cond_pool, dim3((C + 255) / 256, b), dim3(256), 0, s, x, T2, C, c.pool, out);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

}  // namespace

extern "C" {

int ttk_cond_create(ttk_cond** out, const ttk_cond_config* cfg, const ttk_weight_view* w, int n_w) {
	TTK_REQUIRE(out && cfg && w, TTK_E_ARG, "ttk_cond_create: null argument");
	TTK_REQUIRE(cfg->dtype == TTK_F32 || cfg->dtype == TTK_BF16, TTK_E_ARG, "ttk_cond_create: bad dtype %d", cfg->dtype);
	TTK_REQUIRE(cfg->stem == TTK_COND_STEM_CONV1 || cfg->stem == TTK_COND_STEM_DOWN4, TTK_E_ARG, "ttk_cond_create: bad stem %d", cfg->stem);
	TTK_REQUIRE(cfg->num_heads >= 1 && cfg->channels % cfg->num_heads == 0 && cfg->channels % 128 == 0 && cfg->in_channels >= 1 && cfg->num_blocks >= 0,
				TTK_E_ARG, "ttk_cond_create: bad widths (channels %d, heads %d, in %d)", cfg->channels, cfg->num_heads, cfg->in_channels);
	const int hd = cfg->channels / cfg->num_heads;
	TTK_REQUIRE(hd == 64 || hd == 128, TTK_E_ARG, "ttk_cond_create: head width %d unsupported (64 or 128)", hd);
	ttk_cond* h = new ttk_cond();
	h->cfg = *cfg;
	h->dt = cfg->dtype;
	h->es = dtype_size(h->dt);
	h->hd = hd;
	h->groups = gn_groups(cfg->channels);
	const int C = cfg->channels;
	WeightMap wm(w, n_w);
	int rc = TTK_OK;
	auto fail = [&](int code) { h->arena.release(); delete h; return code; };
#define C_TRY(expr) do { rc = (expr); if (rc != TTK_OK) return fail(rc); } while (0)
	if (cfg->stem == TTK_COND_STEM_DOWN4) {
		C_TRY(upload_mat(h->arena, wm, h->dt, "stem.0.weight", "stem.0.bias", PK_NK, C / 2, cfg->in_channels * 3, false, &h->stem0));
		C_TRY(upload_mat(h->arena, wm, h->dt, "stem.1.weight", "stem.1.bias", PK_NK, C, (C / 2) * 3, false, &h->stem1));
	} else {
		C_TRY(upload_mat(h->arena, wm, h->dt, "stem.0.weight", "stem.0.bias", PK_NK, C, cfg->in_channels, false, &h->stem0));
	}
	h->blocks.resize(cfg->num_blocks);
	for (int i = 0; i < cfg->num_blocks; ++i) {
		Block& B = h->blocks[i];
		const std::string p = "blocks." + std::to_string(i) + ".";
		C_TRY(upload_f32(h->arena, wm, p + "norm.weight", C, &B.gn_g));
		C_TRY(upload_f32(h->arena, wm, p + "norm.bias", C, &B.gn_b));
		C_TRY(upload_mat(h->arena, wm, h->dt, p + "qkv.weight", p + "qkv.bias", PK_NK, 3 * C, C, false, &B.qkv));
		C_TRY(upload_mat(h->arena, wm, h->dt, p + "proj_out.weight", p + "proj_out.bias", PK_NK, C, C, false, &B.proj));
		if (cfg->relpos) C_TRY(upload_f32(h->arena, wm, p + "__relbias", (int64_t)cfg->num_heads * 129, &B.relbias));
	}
#undef C_TRY
	hipError_t e = hipDeviceSynchronize();
	if (e != hipSuccess) { set_error("ttk_cond_create: %s", hipGetErrorString(e)); return fail(TTK_E_HIP); }
	*out = h;
	return TTK_OK;
}

int ttk_cond_destroy(ttk_cond* h) {
	if (!h) return TTK_OK;
	(void)hipDeviceSynchronize();
	h->ws.release();
	h->arena.release();
	delete h;
	return TTK_OK;
}

int ttk_cond_encode(ttk_cond* h, const float* mel, int b, int T, float* out, void* stream) {
	TTK_REQUIRE(h && mel && out, TTK_E_ARG, "ttk_cond_encode: null argument");
	TTK_REQUIRE(b >= 1 && T >= 1, TTK_E_ARG, "ttk_cond_encode: empty input (b=%d T=%d)", b, T);
	TTK_REQUIRE(T <= 8192, TTK_E_ARG, "ttk_cond_encode: %d frames exceed the 8192-frame limit of a conditioning clip", T);
	return h->dt == DT_BF16 ? encode_t<bf16>(h, mel, b, T, out, (hipStream_t)stream) : encode_t<float>(h, mel, b, T, out, (hipStream_t)stream);
}

}  // extern "C"
