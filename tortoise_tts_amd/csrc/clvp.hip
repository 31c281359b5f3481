// CLVP candidate scoring on the kernels of the hot path.
//
// Reference: /root/reference/tortoise_tts/models/clvp.py:100-131 (CLVP.forward, return_loss=False) over two x-transformers encoders
// (models/xtransformers.py: ContinuousTransformerWrapper :1189-1248, AttentionLayers.forward :841-1015, Attention.forward :578-731,
// RMSNorm :337-346, FeedForward / GLU :431-478, rotary :266-290).  Per encoder layer:
//     x += to_out( softmax(rot(q) rot(k)^T / 8) rot(v) ),  q|k|v = rmsnorm(x) W_qkv   (bias-free; rotary on the first 32 features of q, k AND v)
//     x += W2 ( val * gelu(gate) ) + b2,                   val|gate = rmsnorm(x) W1 + b1
// then LayerNorm, mean over the sequence, a bias-free Linear, L2 normalisation; score = <text, speech> * exp(temperature).
// Dense GEMMs, attention and LayerNorm are the hot path's kernels; RMSNorm, rotary, GEGLU and the tail are the small kernels below.
#include <math.h>

#include <string>
#include <vector>

#include "ttk_common.h"
#include "ttk_host.h"
#include "ttk_kernels.h"

using namespace ttk;

namespace {

__global__ void k_embed_rows(const float* emb, const int64_t* ids, int64_t n_ids_per_batch, int batch_stride_ids, float* out, int64_t rows, int d, int vocab) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int d4 = d / 4;
	if (idx >= rows * d4) return;
	const int64_t r = idx / d4;
	const int c = (int)(idx - r * d4) * 4;
	const int64_t b = r / n_ids_per_batch, t = r - b * n_ids_per_batch;
	int64_t id = ids[b * batch_stride_ids + t];
	id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);                 // the host validates ids; never index outside the table
	*(float4*)(out + r * d + c) = *(const float4*)(emb + id * d + c);
}

// RMSNorm (xtransformers.py:337-346): y = x / max(||x|| * d^-0.5, 1e-8) * g.  One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void k_rmsnorm(const float* x, int64_t rows, int d, const float* g, T* y) {
	const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
	const int lane = threadIdx.x & 63;
	if (row >= rows) return;
	const float* xr = x + row * d;
	float ss = 0.f;
	for (int c = lane * 4; c < d; c += 256) { const float4 v = *(const float4*)(xr + c); ss += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
	ss = wave_sum(ss);
	const float norm = fmaxf(sqrtf(ss) * rsqrtf((float)d), 1e-8f);
	T* yr = y + row * d;
	for (int c = lane * 4; c < d; c += 256) {
		const float4 v = *(const float4*)(xr + c), gg = *(const float4*)(g + c);
		yr[c] = cvt<T>(v.x / norm * gg.x); yr[c + 1] = cvt<T>(v.y / norm * gg.y); yr[c + 2] = cvt<T>(v.z / norm * gg.z); yr[c + 3] = cvt<T>(v.w / norm * gg.w);
	}
}

// rotary position embedding on the first 32 features of every head of q, k and v, in place on the fused projection [rows][3 * H * 64]:
// (x1, x2) halves of 16 -> (x1 cos - x2 sin, x2 cos + x1 sin), angle = position * inv_freq[i]   (xtransformers.py:266-290, 622-626)
template <typename T>
__global__ void k_rotary(T* qkv, int64_t rows, int S, int H, const float* inv_freq) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;       // (row, which * H + head, i < 16)
	const int per_row = 3 * H * 16;
	if (idx >= rows * per_row) return;
	const int64_t row = idx / per_row;
	const int rem = (int)(idx - row * per_row);
	const int hh = rem >> 4, i = rem & 15;
	const float ang = (float)(row % S) * inv_freq[i];
	float sn, cs;
	sincosf(ang, &sn, &cs);
	T* p = qkv + row * (int64_t)(3 * H * 64) + hh * 64 + i;
	const float x1 = (float)p[0], x2 = (float)p[16];
	p[0] = cvt<T>(x1 * cs - x2 * sn);
	p[16] = cvt<T>(x2 * cs + x1 * sn);
}

// GLU with exact GELU (xtransformers.py:431-440, nn.GELU()): out[r][c] = h[r][c] * gelu(h[r][inner + c])
template <typename T>
__global__ void k_geglu(const T* h, int64_t rows, int inner, T* out) {
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= rows * inner) return;
	const int64_t r = idx / inner;
	const int c = (int)(idx - r * inner);
	const float v = (float)h[r * 2 * inner + c], gt = (float)h[r * 2 * inner + inner + c];
	out[idx] = cvt<T>(v * (0.5f * gt * (1.0f + erff(gt * 0.70710678118654752f))));
}

// mean over the S positions of each batch element (masked_mean with an all-true mask, clvp.py:17-19,123-124) -> T [B][d]
template <typename T>
__global__ void k_mean_rows(const float* x, int B, int S, int d, T* out) {
	const int idx = blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= B * d) return;
	const int b = idx / d, c = idx - b * d;
	float s = 0.f;
	for (int t = 0; t < S; ++t) s += x[((int64_t)b * S + t) * d + c];
	out[idx] = cvt<T>(s / (float)S);
}

// score[b] = <normalize(text[b or 0]), normalize(speech[b])> * exp(temperature)     (F.normalize eps 1e-12)
__global__ __launch_bounds__(64) void k_clvp_score(const float* tl, int Bt, const float* sl, int d, const float* temperature, float* scores) {
	const int b = blockIdx.x, lane = threadIdx.x;
	const float* t = tl + (int64_t)(Bt == 1 ? 0 : b) * d;
	const float* s = sl + (int64_t)b * d;
	float tt = 0.f, ss = 0.f, ts = 0.f;
	for (int c = lane; c < d; c += 64) { tt += t[c] * t[c]; ss += s[c] * s[c]; ts += t[c] * s[c]; }
	tt = wave_sum(tt); ss = wave_sum(ss); ts = wave_sum(ts);
	if (lane == 0) scores[b] = ts / (fmaxf(sqrtf(tt), 1e-12f) * fmaxf(sqrtf(ss), 1e-12f)) * expf(*temperature);
}

struct EncLayer { float *g_attn = nullptr, *g_ff = nullptr; Mat qkv, out, ff1, ff2; };
struct Encoder { float* emb = nullptr; int vocab = 0; std::vector<EncLayer> L; float *norm_g = nullptr, *norm_b = nullptr; Mat to_latent; };

}  // namespace

struct ttk_clvp {
	ttk_clvp_config cfg;
	int dt;
	size_t es;
	Arena arena;
	Encoder text, speech;
	float *inv_freq = nullptr, *temperature = nullptr;
	WsBuf ws;
};

namespace {

void gemm_plain(int dt, const void* A, int lda, const Mat& w, int M, const float* residual, void* C, int out_f32, hipStream_t s) {
	GemmParams g = {};
	g.nseg = 1;
	g.seg[0] = {A, lda, 0, 0};
	g.W = w.w; g.ldw = w.Kpad; g.M = M; g.N = w.N; g.K = w.Kpad; g.bias = w.bias;
	g.residual = residual; g.ldr = w.N; g.C = C; g.ldc = w.N; g.out_f32 = out_f32;
	launch_gemm(dt, g, s);
}

template <typename T>
void encode_t(ttk_clvp* h, const Encoder& e, const int64_t* ids, int B, int S, int ids_stride, float* latent /* [B][d] f32 */, hipStream_t s) {
	const ttk_clvp_config& c = h->cfg;
	const int d = c.dim, H = c.heads, inner = c.inner, dt = h->dt;
	const int64_t rows = (int64_t)B * S;
	char* base = (char*)h->ws.p;
	float* x = (float*)base;                                        // [rows][d] f32 residual stream
	T* xn = (T*)(base + (size_t)rows * d * 4);                      // [rows][d]
	T* qkv = xn + rows * d;                                         // [rows][3d]  (also the GLU projection [rows][2 * inner])
	T* ao = qkv + rows * (size_t)std::max(3 * d, 2 * inner);        // [rows][max(d, inner)]
	{
		const int64_t total = rows * (d / 4);
		hipLaunchKernelGGL(k_embed_rows, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, e.emb, ids, (int64_t)S, ids_stride, x, rows, d, e.vocab);
	}
	for (const EncLayer& L : e.L) {
		hipLaunchKernelGGL((k_rmsnorm<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, d, L.g_attn, xn);
		gemm_plain(dt, xn, d, L.qkv, (int)rows, nullptr, qkv, 0, s);
		{
			const int64_t total = rows * 3 * H * 16;
			hipLaunchKernelGGL((k_rotary<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, rows, S, H, h->inv_freq);
		}
		AttnParams a = {};
		a.qkv = qkv; a.ld = 3 * d; a.q_off = 0; a.k_off = d; a.v_off = 2 * d; a.head_stride = 64; a.out = ao; a.ldo = d;
		a.nb = B; a.T = S; a.H = H; a.causal = 0; a.bias = nullptr; a.scale = 0.125f;
		launch_attn_fwd(dt, a, s);
		gemm_plain(dt, ao, d, L.out, (int)rows, x, x, 1, s);
		hipLaunchKernelGGL((k_rmsnorm<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, d, L.g_ff, xn);
		gemm_plain(dt, xn, d, L.ff1, (int)rows, nullptr, qkv, 0, s);
		{
			const int64_t total = rows * inner;
			hipLaunchKernelGGL((k_geglu<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, qkv, rows, inner, ao);
		}
		gemm_plain(dt, ao, inner, L.ff2, (int)rows, x, x, 1, s);
	}
	float* xf = (float*)qkv;                                         // final LayerNorm output, f32 [rows][d] (the qkv buffer is free now)
	launch_layernorm(dt, x, d, (int)rows, d, e.norm_g, e.norm_b, nullptr, nullptr, xf, d, 1, s);
	hipLaunchKernelGGL((k_mean_rows<T>), dim3((unsigned)((B * d + 255) / 256)), dim3(256), 0, s, xf, B, S, d, xn);
	gemm_plain(dt, xn, d, e.to_latent, B, nullptr, latent, 1, s);
}

int upload_encoder(ttk_clvp* h, const WeightMap& wm, const std::string& which, int vocab, Encoder* e) {
	const ttk_clvp_config& c = h->cfg;
	const int d = c.dim;
	e->vocab = vocab;
	TTK_TRY(upload_f32(h->arena, wm, which + "_emb.weight", (int64_t)vocab * d, &e->emb));
	const std::string p = which + "_transformer.transformer.";
	e->L.resize(c.depth);
	for (int i = 0; i < c.depth; ++i) {
		EncLayer& L = e->L[i];
		const std::string a = p + "attn_layers.layers." + std::to_string(2 * i) + ".", f = p + "attn_layers.layers." + std::to_string(2 * i + 1) + ".";
		TTK_TRY(upload_f32(h->arena, wm, a + "0.0.g", d, &L.g_attn));
		TTK_TRY(upload_mat(h->arena, wm, h->dt, a + "1.wrap.__qkv.weight", "", PK_NK, 3 * d, d, false, &L.qkv));
		TTK_TRY(upload_mat(h->arena, wm, h->dt, a + "1.wrap.to_out.weight", a + "1.wrap.to_out.bias", PK_NK, d, d, false, &L.out));
		TTK_TRY(upload_f32(h->arena, wm, f + "0.0.g", d, &L.g_ff));
		TTK_TRY(upload_mat(h->arena, wm, h->dt, f + "1.wrap.net.0.proj.weight", f + "1.wrap.net.0.proj.bias", PK_NK, 2 * c.inner, d, false, &L.ff1));
		TTK_TRY(upload_mat(h->arena, wm, h->dt, f + "1.wrap.net.3.weight", f + "1.wrap.net.3.bias", PK_NK, d, c.inner, false, &L.ff2));
	}
	TTK_TRY(upload_f32(h->arena, wm, p + "norm.weight", d, &e->norm_g));
	TTK_TRY(upload_f32(h->arena, wm, p + "norm.bias", d, &e->norm_b));
	TTK_TRY(upload_mat(h->arena, wm, h->dt, "to_" + which + "_latent.weight", "", PK_NK, d, d, false, &e->to_latent));
	return TTK_OK;
}

}  // namespace

extern "C" {

int ttk_clvp_create(ttk_clvp** out, const ttk_clvp_config* cfg, const ttk_weight_view* w, int n_w) {
	TTK_REQUIRE(out && cfg && w, TTK_E_ARG, "ttk_clvp_create: null argument");
	TTK_REQUIRE(cfg->dtype == TTK_F32 || cfg->dtype == TTK_BF16, TTK_E_ARG, "ttk_clvp_create: bad dtype %d", cfg->dtype);
	TTK_REQUIRE(cfg->dim % 64 == 0 && cfg->heads * 64 == cfg->dim, TTK_E_ARG, "ttk_clvp_create: head width must be 64 with heads * 64 == dim (dim %d, heads %d)", cfg->dim, cfg->heads);
	TTK_REQUIRE(cfg->inner % 64 == 0 && cfg->depth >= 1, TTK_E_ARG, "ttk_clvp_create: inner width %d / depth %d unsupported", cfg->inner, cfg->depth);
	ttk_clvp* h = new ttk_clvp();
	h->cfg = *cfg;
	h->dt = cfg->dtype;
	h->es = dtype_size(h->dt);
	WeightMap wm(w, n_w);
	int rc = TTK_OK;
	auto fail = [&](int code) { h->arena.release(); delete h; return code; };
#define C_TRY(expr) do { rc = (expr); if (rc != TTK_OK) return fail(rc); } while (0)
	C_TRY(upload_f32(h->arena, wm, "__rotary_inv_freq", 16, &h->inv_freq));
	C_TRY(upload_f32(h->arena, wm, "temperature", 1, &h->temperature));
	C_TRY(upload_encoder(h, wm, "text", cfg->num_text_tokens, &h->text));
	C_TRY(upload_encoder(h, wm, "speech", cfg->num_speech_tokens, &h->speech));
#undef C_TRY
	*out = h;
	return TTK_OK;
}

int ttk_clvp_destroy(ttk_clvp* h) {
	if (!h) return TTK_OK;
	h->ws.release();
	h->arena.release();
	delete h;
	return TTK_OK;
}

int ttk_clvp_score(ttk_clvp* h, const int64_t* text, int Bt, int Tt, const int64_t* codes, int B, int M, float* scores, void* stream) {
	TTK_REQUIRE(h && text && codes && scores, TTK_E_ARG, "ttk_clvp_score: null argument");
	TTK_REQUIRE(B >= 1 && Tt >= 1 && M >= 1 && (Bt == 1 || Bt == B), TTK_E_ARG, "ttk_clvp_score: bad shape (Bt=%d Tt=%d B=%d M=%d)", Bt, Tt, B, M);
	const ttk_clvp_config& c = h->cfg;
	hipStream_t s = (hipStream_t)stream;
	const int d = c.dim;
	const int64_t rows = std::max((int64_t)Bt * Tt, (int64_t)B * M);
	const size_t per_row = (size_t)d * 4 + (size_t)(d + std::max(3 * d, 2 * c.inner) + std::max(d, c.inner)) * h->es;
	const size_t lat_off = (rows * per_row + 255) / 256 * 256;
	TTK_TRY(h->ws.reserve(lat_off + (size_t)2 * B * d * 4 + 256));
	float* tl = (float*)((char*)h->ws.p + lat_off);
	float* sl = tl + (size_t)B * d;
	if (h->dt == DT_BF16) {
		encode_t<bf16>(h, h->text, text, Bt, Tt, Tt, tl, s);
		encode_t<bf16>(h, h->speech, codes, B, M, M, sl, s);
	} else {
		encode_t<float>(h, h->text, text, Bt, Tt, Tt, tl, s);
		encode_t<float>(h, h->speech, codes, B, M, M, sl, s);
	}
	hipLaunchKernelGGL(k_clvp_score, dim3(B), dim3(64), 0, s, tl, Bt, sl, d, h->temperature, scores);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

}  // extern "C"
