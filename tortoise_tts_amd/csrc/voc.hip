// BigVGAN generator (the vocoder tail of BASELINE config 5) on the kernels of the hot path.
//
// Reference: /root/reference/tortoise_tts/models/bigvgan.py -- BigVGAN.forward :488-510, inference :522-534, AMPBlock1 :306-358,
// Activation1d :158-181 (UpSample1d :113-136, DownSample1d :139-153, kaiser_sinc_filter1d :40-69), SnakeBeta :237-295.
//
// Layout: everything between the first and the last convolution is channels-last [B * L][C], f32 for the residual streams, T for
// GEMM inputs.  Every Conv1d is the hot path's segment GEMM (one segment per tap, row shift = (tap - (k-1)/2) * dilation, zero rows
// outside a batch element); channel counts below one k-tile read past the row into finite neighbouring data that meets zero-padded
// weight columns.  A ConvTranspose1d of stride u is u such GEMMs, one per output phase, each seeing k / u taps and writing its phase
// through the output row stride.  The anti-aliased snake activation (2x zero-stuffing up-sampler with a 12-tap Kaiser low-pass,
// SnakeBeta, 12-tap low-pass + decimation by 2) is ONE kernel: each output sample evaluates the 12 up-sampled activations it needs
// from 13 input rows held in registers.
#include <math.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "ttk_common.h"
#include "ttk_host.h"
#include "ttk_kernels.h"

using namespace ttk;

namespace {

// mel [B][C][T] f32 (log-mel) -> T-typed [B * (T + pad)][ldo]: channels-last, `pad` extra frames of pad_value (BigVGAN.inference :525-526),
// channels beyond C zero (they meet zero weight columns)
template <typename T>
__global__ void k_voc_mel_in(const float* mel, int B, int C, int Tm, int pad, float pad_value, T* out, int ldo) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int L = Tm + pad;
	if (idx >= (int64_t)B * L * ldo) return;
	const int c = (int)(idx % ldo);
	const int64_t row = idx / ldo;
	const int b = (int)(row / L), t = (int)(row - (int64_t)b * L);
	float v = 0.f;
	if (c < C) v = t < Tm ? mel[((int64_t)b * C + c) * Tm + t] : pad_value;
	out[idx] = cvt<T>(v);
}

template <typename T> __device__ __forceinline__ float snake_sin(float x);
template <> __device__ __forceinline__ float snake_sin<float>(float x) { return sinf(x); }        // exact mode: accurate sine
template <> __device__ __forceinline__ float snake_sin<bf16>(float x) { return __sinf(x); }       // v_sin_f32; the result is rounded to bf16

// Activation1d with SnakeBeta.  x f32 [B * L][C] -> y T [B * L][C].  One thread = one row, 4 channels.
//   up[s]  = 2 * sum_j x[clamp(q - 3 + j)] * g[11 - 2j]   (s = 2q even),   2 * sum_j x[clamp(q - 2 + j)] * g[10 - 2j]   (s = 2q + 1)
//   z[s]   = up[s] + inv_b * sin(up[s] * a)^2
//   y[t]   = sum_k z[clamp(2t + k - 5, 0, 2L - 1)] * g[k]
// (the zero-stuffed transposed convolution of UpSample1d written per output phase; clamps are the two replicate paddings)
template <typename T>
__global__ __launch_bounds__(256) void k_snake_aa(const float* x, int L, int C, const float* a_, const float* invb_, const float* g_, T* y, int64_t rows) {
	const int c4 = C / 4;
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= rows * c4) return;
	const int c = (int)(idx % c4) * 4;
	const int64_t row = idx / c4;
	const int t = (int)(row % L);
	const float* xb = x + (row - t) * C + c;              // this batch element, these channels
	float g[12];
#pragma unroll
	for (int k = 0; k < 12; ++k) g[k] = g_[k];
	const float4 a = *(const float4*)(a_ + c), ib = *(const float4*)(invb_ + c);
	float4 xr[13];                                        // rows t-6 .. t+6, replicate-clamped
#pragma unroll
	for (int j = 0; j < 13; ++j) {
		int e = t - 6 + j;
		e = e < 0 ? 0 : (e > L - 1 ? L - 1 : e);
		xr[j] = *(const float4*)(xb + (int64_t)e * C);
	}
	// the 12 up-sampled, activated samples s = 2t + k - 5 this output sees.  Row e of the input sits in register e - (t - 6); because
	// the rows were clamped when loaded, the same register also serves the replicate padding of x, so all indices are compile-time.
	float4 z[12];
#pragma unroll
	for (int k = 0; k < 12; ++k) {
		const int fl = (k - 5) >> 1, odd = (k - 5) & 1;          // s = 2q + odd with q = t + fl
		float4 up = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
		for (int j = 0; j < 6; ++j) {
			const float4 v = xr[fl + 3 + odd + j];
			const float w = g[11 - odd - 2 * j];
			up.x += v.x * w; up.y += v.y * w; up.z += v.z * w; up.w += v.w * w;
		}
		up.x *= 2.f; up.y *= 2.f; up.z *= 2.f; up.w *= 2.f;
		const float sx = snake_sin<T>(up.x * a.x), sy = snake_sin<T>(up.y * a.y), sz = snake_sin<T>(up.z * a.z), sw = snake_sin<T>(up.w * a.w);
		z[k] = make_float4(up.x + ib.x * (sx * sx), up.y + ib.y * (sy * sy), up.z + ib.z * (sz * sz), up.w + ib.w * (sw * sw));
	}
	// replicate padding of the up-sampled signal: samples before s = 0 (k < k0) repeat z[k0], samples after s = 2L - 1 (k > k1) repeat z[k1]
	const int k0 = 5 - 2 * t, k1 = 2 * L + 4 - 2 * t;
	if (k0 > 0 || k1 < 11) {                                         // only the first / last three rows of a sequence
		float4 zlo = z[0], zhi = z[11];
#pragma unroll
		for (int m = 1; m < 12; ++m) { if (m == k0) zlo = z[m]; if (m == k1) zhi = z[m]; }
#pragma unroll
		for (int k = 0; k < 12; ++k) { if (k < k0) z[k] = zlo; if (k > k1) z[k] = zhi; }
	}
	float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
	for (int k = 0; k < 12; ++k) { acc.x += z[k].x * g[k]; acc.y += z[k].y * g[k]; acc.z += z[k].z * g[k]; acc.w += z[k].w * g[k]; }
	T* o = y + row * C + c;
	o[0] = cvt<T>(acc.x); o[1] = cvt<T>(acc.y); o[2] = cvt<T>(acc.z); o[3] = cvt<T>(acc.w);
}

// x = (y0 + y1 [+ y2 ...]) / n  (BigVGAN.forward :498-504), f32 out + T copy for the next transposed convolution
template <typename T>
__global__ void k_voc_mean(const float* y0, const float* y1, const float* y2, const float* y3, int n, float* out, T* out_t, int64_t total) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= total) return;
	float s = y0[i] + y1[i];
	if (n > 2) s += y2[i];
	if (n > 3) s += y3[i];
	s = s / (float)n;
	out[i] = s;
	out_t[i] = cvt<T>(s);
}

// audio[b][t] = clamp(tanh(y[b * L + t]), -1, 1) for t < keep   (forward :507-508, inference :531-533)
__global__ void k_voc_out(const float* y, int B, int L, int keep, float* audio) {
	const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= (int64_t)B * keep) return;
	const int b = (int)(i / keep), t = (int)(i - (int64_t)b * keep);
	float v = tanhf(y[(int64_t)b * L + t]);
	audio[i] = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
}

struct Snake { float *a = nullptr, *invb = nullptr; };
struct AmpBlock { Mat c1[3], c2[3]; Snake act[6]; int k = 3; int dil[3] = {1, 3, 5}; };

}  // namespace

struct ttk_voc {
	ttk_voc_config cfg;
	int dt;
	size_t es;
	Arena arena;
	Mat conv_pre, conv_post;
	std::vector<Mat> ups;
	std::vector<AmpBlock> blocks;
	Snake act_post;
	float* filt = nullptr;
	WsBuf ws;
};

namespace {

template <typename T>
void launch_snake_t(const float* x, int L, int C, const Snake& sn, const float* g, void* y, int64_t rows, hipStream_t s) {
	const int64_t total = rows * (C / 4);
	hipLaunchKernelGGL((k_snake_aa<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, L, C, sn.a, sn.invb, g, (T*)y, rows);
}
void launch_snake(int dt, const float* x, int L, int C, const Snake& sn, const float* g, void* y, int64_t rows, hipStream_t s) {
	if (dt == DT_BF16) launch_snake_t<bf16>(x, L, C, sn, g, y, rows, s);
	else launch_snake_t<float>(x, L, C, sn, g, y, rows, s);
}

// Conv1d(k taps, dilation) over rows: out[M][N] = sum_j A[m + (j - (k-1)/2) * dil] * W_j^T + bias (+ residual)
void conv_rows(int dt, const void* A, int lda, const Mat& w, int k, int dil, int M, int L, const float* residual, void* C, int out_f32, hipStream_t s) {
	GemmParams g = {};
	g.nseg = k;
	for (int j = 0; j < k; ++j) g.seg[j] = {A, lda, (j - (k - 1) / 2) * dil, (int64_t)j * w.Npad * w.Kpad};
	g.W = w.w; g.ldw = w.Kpad; g.M = M; g.N = w.N; g.K = w.Kpad; g.rows_per_batch = L; g.bias = w.bias;
	g.residual = residual; g.ldr = w.N; g.C = C; g.ldc = w.N; g.out_f32 = out_f32;
	launch_gemm(dt, g, s);
}

int upload_snake(Arena& ar, const WeightMap& wm, const std::string& prefix, int C, bool logscale, Snake* out) {
	float *alpha = nullptr, *beta = nullptr;
	TTK_TRY(upload_f32(ar, wm, prefix + "alpha", C, &alpha));
	TTK_TRY(upload_f32(ar, wm, prefix + "beta", C, &beta));
	// a = exp(alpha), inv_b = 1 / (exp(beta) + 1e-9) in f32, the operations of SnakeBeta.forward :287-292
	std::vector<float> ha(C), hb(C);
	TTK_HIP(hipMemcpy(ha.data(), alpha, (size_t)C * 4, hipMemcpyDeviceToHost));
	TTK_HIP(hipMemcpy(hb.data(), beta, (size_t)C * 4, hipMemcpyDeviceToHost));
	for (int i = 0; i < C; ++i) {
		const float a = logscale ? expf(ha[i]) : ha[i], b = logscale ? expf(hb[i]) : hb[i];
		ha[i] = a; hb[i] = 1.0f / (b + 0.000000001f);
	}
	TTK_HIP(hipMemcpy(alpha, ha.data(), (size_t)C * 4, hipMemcpyHostToDevice));
	TTK_HIP(hipMemcpy(beta, hb.data(), (size_t)C * 4, hipMemcpyHostToDevice));
	out->a = alpha; out->invb = beta;
	return TTK_OK;
}

}  // namespace

extern "C" {

int ttk_voc_create(ttk_voc** out, const ttk_voc_config* cfg, const ttk_weight_view* w, int n_w) {
	TTK_REQUIRE(out && cfg && w, TTK_E_ARG, "ttk_voc_create: null argument");
	TTK_REQUIRE(cfg->dtype == TTK_F32 || cfg->dtype == TTK_BF16, TTK_E_ARG, "ttk_voc_create: bad dtype %d", cfg->dtype);
	TTK_REQUIRE(cfg->n_ups >= 1 && cfg->n_ups <= 8 && cfg->n_kernels >= 2 && cfg->n_kernels <= 4, TTK_E_ARG,
				"ttk_voc_create: %d upsamplers / %d resblock kernels unsupported (1..8 / 2..4)", cfg->n_ups, cfg->n_kernels);
	TTK_REQUIRE(cfg->num_mels >= 1 && cfg->num_mels <= 128, TTK_E_ARG, "ttk_voc_create: num_mels %d out of range", cfg->num_mels);
	int ch = cfg->ch0;
	for (int i = 0; i < cfg->n_ups; ++i) {
		TTK_REQUIRE(cfg->up_kernel[i] % cfg->up_rate[i] == 0 && (cfg->up_kernel[i] - cfg->up_rate[i]) % 2 == 0 && cfg->up_kernel[i] / cfg->up_rate[i] <= 12, TTK_E_ARG,
					"ttk_voc_create: upsampler %d (kernel %d, rate %d) unsupported", i, cfg->up_kernel[i], cfg->up_rate[i]);
		TTK_REQUIRE(ch % 2 == 0, TTK_E_ARG, "ttk_voc_create: channel count %d does not halve", ch);
		ch /= 2;
		TTK_REQUIRE(ch % 8 == 0, TTK_E_ARG, "ttk_voc_create: stage %d has %d channels (must be a multiple of 8)", i, ch);
	}
	for (int j = 0; j < cfg->n_kernels; ++j)
		TTK_REQUIRE(cfg->rb_kernel[j] % 2 == 1 && cfg->rb_kernel[j] <= 11, TTK_E_ARG, "ttk_voc_create: resblock kernel %d unsupported (odd, <= 11)", cfg->rb_kernel[j]);
	ttk_voc* h = new ttk_voc();
	h->cfg = *cfg;
	h->dt = cfg->dtype;
	h->es = dtype_size(h->dt);
	WeightMap wm(w, n_w);
	int rc = TTK_OK;
	auto fail = [&](int code) { h->arena.release(); delete h; return code; };
#define V_TRY(expr) do { rc = (expr); if (rc != TTK_OK) return fail(rc); } while (0)
	V_TRY(upload_f32(h->arena, wm, "__aa_filter", 12, &h->filt));
	V_TRY(upload_mat(h->arena, wm, h->dt, "conv_pre.weight", "conv_pre.bias", PK_CONVK, cfg->ch0, cfg->num_mels, false, &h->conv_pre, 7));
	h->ups.resize(cfg->n_ups);
	h->blocks.resize((size_t)cfg->n_ups * cfg->n_kernels);
	ch = cfg->ch0;
	for (int i = 0; i < cfg->n_ups; ++i) {
		const std::string u = "ups." + std::to_string(i) + ".0.";
		V_TRY(upload_mat(h->arena, wm, h->dt, u + "weight", u + "bias", PK_CONVT, ch / 2, ch, false, &h->ups[i], cfg->up_kernel[i]));
		ch /= 2;
		for (int j = 0; j < cfg->n_kernels; ++j) {
			AmpBlock& b = h->blocks[(size_t)i * cfg->n_kernels + j];
			b.k = cfg->rb_kernel[j];
			const std::string p = "resblocks." + std::to_string(i * cfg->n_kernels + j) + ".";
			for (int m = 0; m < 3; ++m) {
				b.dil[m] = cfg->rb_dil[j][m];
				V_TRY(upload_mat(h->arena, wm, h->dt, p + "convs1." + std::to_string(m) + ".weight", p + "convs1." + std::to_string(m) + ".bias", PK_CONVK, ch, ch, false, &b.c1[m], b.k));
				V_TRY(upload_mat(h->arena, wm, h->dt, p + "convs2." + std::to_string(m) + ".weight", p + "convs2." + std::to_string(m) + ".bias", PK_CONVK, ch, ch, false, &b.c2[m], b.k));
			}
			for (int m = 0; m < 6; ++m) V_TRY(upload_snake(h->arena, wm, p + "activations." + std::to_string(m) + ".act.", ch, cfg->snake_logscale != 0, &b.act[m]));
		}
	}
	V_TRY(upload_snake(h->arena, wm, "activation_post.act.", ch, cfg->snake_logscale != 0, &h->act_post));
	V_TRY(upload_mat(h->arena, wm, h->dt, "conv_post.weight", "conv_post.bias", PK_CONVK, 1, ch, false, &h->conv_post, 7));
#undef V_TRY
	*out = h;
	return TTK_OK;
}

int ttk_voc_destroy(ttk_voc* h) {
	if (!h) return TTK_OK;
	h->ws.release();
	h->arena.release();
	delete h;
	return TTK_OK;
}

int ttk_voc_inference(ttk_voc* h, const float* mel, int B, int Tm, float* audio, void* stream) {
	TTK_REQUIRE(h && mel && audio, TTK_E_ARG, "ttk_voc_inference: null argument");
	TTK_REQUIRE(B >= 1 && Tm >= 1, TTK_E_ARG, "ttk_voc_inference: empty input (B=%d T=%d)", B, Tm);
	const ttk_voc_config& c = h->cfg;
	hipStream_t s = (hipStream_t)stream;
	const int dt = h->dt;
	const size_t es = h->es;
	const int PAD = 10;
	int hop = 1;
	for (int i = 0; i < c.n_ups; ++i) hop *= c.up_rate[i];
	const int L0 = Tm + PAD;
	TTK_REQUIRE((int64_t)B * L0 * hop < (int64_t)1 << 30, TTK_E_ARG, "ttk_voc_inference: %d x %d frames is too long for one call", B, Tm);
	// workspace: the largest stage decides (rows * channels is constant or grows by 2 per stage)
	int64_t max_el = (int64_t)B * L0 * c.ch0;
	{
		int64_t L = L0; int ch = c.ch0;
		for (int i = 0; i < c.n_ups; ++i) { L *= c.up_rate[i]; ch /= 2; max_el = std::max(max_el, (int64_t)B * L * ch); }
	}
	TTK_REQUIRE(max_el * 4 < ((int64_t)1 << 31), TTK_E_ARG, "ttk_voc_inference: %d x %d frames exceed the 2 GiB buffer range of one call", B, Tm);
	const int mel_ld = h->conv_pre.Kpad;
	const size_t f32b = (size_t)max_el * 4, tb = (size_t)max_el * es;
	const size_t off_in = 0, off_xt = off_in + (size_t)B * L0 * mel_ld * es, off_at = off_xt + tb, off_y = off_at + tb, off_h = off_y + f32b, off_xb = off_h + f32b;
	TTK_TRY(h->ws.reserve(off_xb + f32b * c.n_kernels + 256));
	char* base = (char*)h->ws.p;
	void* mel_t = base + off_in;      // T [B*L0][mel_ld]
	void* xt = base + off_xt;         // T copy of the stage input (A operand of the transposed convolution)
	void* at = base + off_at;         // T output of the activation kernel (A operand of the AMP convolutions)
	float* y = (float*)(base + off_y);    // f32 transposed-conv output = input of the stage's AMP blocks; later the stage mean
	float* hb = (float*)(base + off_h);   // f32 output of convs1
	float* xb[4];
	for (int j = 0; j < 4; ++j) xb[j] = (float*)(base + off_xb + f32b * (j < c.n_kernels ? j : 0));

	{
		const int64_t total = (int64_t)B * L0 * mel_ld;
		if (dt == DT_BF16) hipLaunchKernelGGL((k_voc_mel_in<bf16>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mel, B, c.num_mels, Tm, PAD, -11.5129f, (bf16*)mel_t, mel_ld);
		else hipLaunchKernelGGL((k_voc_mel_in<float>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, mel, B, c.num_mels, Tm, PAD, -11.5129f, (float*)mel_t, mel_ld);
	}
	conv_rows(dt, mel_t, mel_ld, h->conv_pre, 7, 1, B * L0, L0, nullptr, xt, 0, s);      // conv_pre -> T [B*L0][ch0]
	int L = L0, ch = c.ch0;
	for (int i = 0; i < c.n_ups; ++i) {
		const int u = c.up_rate[i], k = c.up_kernel[i], pd = (k - u) / 2, cout = ch / 2;
		const Mat& W = h->ups[i];
		// transposed convolution, one GEMM per output phase r: y[u*m + r] = sum_i x[m + (r + pd - j_i) / u] * W[:, :, j_i],  j_i = (r + pd) % u + u*i
		for (int r = 0; r < u; ++r) {
			GemmParams g = {};
			g.nseg = k / u;
			for (int t = 0; t < g.nseg; ++t) {
				const int j = (r + pd) % u + u * t;
				g.seg[t] = {xt, ch, (r + pd - j) / u, (int64_t)j * W.Npad * W.Kpad};
			}
			g.W = W.w; g.ldw = W.Kpad; g.M = B * L; g.N = cout; g.K = W.Kpad; g.rows_per_batch = L; g.bias = W.bias;
			g.C = y + (size_t)r * cout; g.ldc = (int64_t)u * cout; g.out_f32 = 1;
			launch_gemm(dt, g, s);
		}
		L *= u; ch = cout;
		const int M = B * L;
		for (int j = 0; j < c.n_kernels; ++j) {
			const AmpBlock& b = h->blocks[(size_t)i * c.n_kernels + j];
			const float* cur = y;
			for (int m = 0; m < 3; ++m) {
				launch_snake(dt, cur, L, ch, b.act[2 * m], h->filt, at, M, s);
				conv_rows(dt, at, ch, b.c1[m], b.k, b.dil[m], M, L, nullptr, hb, 1, s);
				launch_snake(dt, hb, L, ch, b.act[2 * m + 1], h->filt, at, M, s);
				conv_rows(dt, at, ch, b.c2[m], b.k, 1, M, L, cur, xb[j], 1, s);          // + bias + residual (aliases the output from m = 1 on)
				cur = xb[j];
			}
		}
		{
			const int64_t total = (int64_t)M * ch;
			const unsigned grid = (unsigned)((total + 255) / 256);
			if (dt == DT_BF16) hipLaunchKernelGGL((k_voc_mean<bf16>), dim3(grid), dim3(256), 0, s, xb[0], xb[1], xb[2], xb[3], c.n_kernels, y, (bf16*)xt, total);
			else hipLaunchKernelGGL((k_voc_mean<float>), dim3(grid), dim3(256), 0, s, xb[0], xb[1], xb[2], xb[3], c.n_kernels, y, (float*)xt, total);
		}
	}
	const int M = B * L;
	launch_snake(dt, y, L, ch, h->act_post, h->filt, at, M, s);
	conv_rows(dt, at, ch, h->conv_post, 7, 1, M, L, nullptr, hb, 1, s);                  // [M][1] f32
	{
		const int keep = Tm * hop;
		const int64_t total = (int64_t)B * keep;
		hipLaunchKernelGGL(k_voc_out, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, hb, B, L, keep, audio);
	}
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

}  // extern "C"
