// ttk_diff: the diffusion half of the hot path -- DiffusionTTS and the per-step sampler behind the C ABI of include/ttk.h.
// Internal layout is channels-last ([b*T rows][C]) so every 1x1 conv is an NT GEMM on the conv weight as stored and the
// k=3 convs are 3 row-shifted GEMM segments; the boundary stays the reference's [b, C, T].
// The conditioned and the conditioning-free evaluation of a sampler step run as ONE batch of 2b sequences (same x, same t,
// different code embedding) so each weight is read once per step.
// Reference: /root/reference/tortoise_tts/models/diffusion.py:1316-1574 (ResBlock, DiffusionLayer, DiffusionTTS),
//            :325-431,646-694,510-554 (p_mean_variance, ddim_sample, p_sample); arch_utils.py:136-190 (AttentionBlock).
#include <stdlib.h>

#include "ttk_common.h"
#include "ttk_host.h"

using namespace ttk;

namespace {
struct AttnBlk { float *gn_g, *gn_b; Mat qkv, proj; float* relbias; };
struct ResBlk { float *gn1_g, *gn1_b, *gn2_g, *gn2_b; Mat in, out3; int emb_slot; };
struct DLayer { ResBlk res; AttnBlk attn; };
// the scratch a chain of ResBlocks / AttentionBlocks works in; two of them so that the conditioning integrator of the NEXT sampler step
// can run on a side stream beside the main body of the current one
struct Lane {
	WsBuf a, hf, qkv, ao, ms;
	const void* ms_owner = nullptr;   // tensor whose GroupNorm statistics currently sit in `ms` (written by a GEMM epilogue)
};
}  // namespace

struct ttk_diff {
	ttk_diff_config cfg;
	int prefetch = 1;       // GroupNorm-apply launches touch the following GEMM's weights into L2 (TTK_DIFF_PREFETCH=0: off)
	int dt, wdt;            // kernel arithmetic type / storage type of the block GEMM weights (== dt, DT_FP8W or DT_FP8)
	int a8 = 0;             // DT_FP8: the block GEMMs take fp8 activations too (written by GroupNorm-apply / attention) and run on the fp8 MFMA
	size_t es;
	Arena arena;
	AttnBlk lat_attn[4];
	Mat lat_conv, inp_block, integ, out_conv, time0, time2, emb_cat;
	float *code_g, *code_b, *out_g, *out_b, *uncond, *time_freqs;
	DLayer integrator[3];
	std::vector<DLayer> layers;
	ResBlk tail[3];
	int n_emb;               // number of ResBlocks = rows of emb_cat / 2C
	int in_pad;              // in_channels rounded up to 64
	// workspaces
	Lane lane[2];
	Lane* L = &lane[0];      // the lane the block helpers below enqueue into (host code is sequential: flipped around the side-stream work)
	WsBuf cs, cs2, xs, h0, csT, xcl, outb, ecl, ms_ecl, temb, e1, e2, se, emb_all, lat_T;
	WsBuf hf0, ms_hf0;      // in_layers of the FIRST integrator ResBlock applied to the staged embedding (+ the GroupNorm statistics of the result): the same in every step
	hipStream_t side = nullptr;
	hipEvent_t ev_fork = nullptr, ev_int[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
	int hf0_valid = 0;      // hf0 / ms_hf0 hold the staged embedding's first half-block (TTK_DIFF_HOIST=0: recomputed in every step, as before round 3)
	int pipe = 1;           // ttk_diff_sample_ddim overlaps step i+1's integrator with step i's body (TTK_DIFF_PIPE=0: sequential)
	int cur_b = 0, cur_T = 0, staged = 0;
	int fuse_stats = 1;
	// ragged batch (ttk_diff_sample_ddim_lines): per-sequence valid rows inside the common slot of cur_T rows, [cond b | uncond b]; d_need marks
	// the sequences whose GroupNorm statistics cannot come from a GEMM epilogue (length not a multiple of its 64-row blocks): they get the
	// separate statistics launch, exactly as a batch of their own length would
	int* d_tlen = nullptr; int* d_need = nullptr;
	const int* tlen = nullptr; const int* need = nullptr;      // = d_tlen / d_need while a ragged loop runs, else null
};

// gn_T > 0: the output is a GroupNorm input of gn_T rows per batch element; its statistics are produced in the epilogue when the
// shape allows (gemm_fuses_gn_stats), which saves the separate k_gn_stats launch.
static void want_stats(ttk_diff* h, GemmParams& g, int gn_T) {
	h->L->ms_owner = nullptr;
	if (h->fuse_stats && gn_T > 0 && g.out_f32 && !g.transpose_out && gemm_fuses_gn_stats(g.M, g.N, h->cfg.model_channels, gn_T)) {
		g.gn_part = (float*)h->L->ms.p; g.gn_T = gn_T; h->L->ms_owner = g.C;
	}
}

static void gemm1(ttk_diff* h, const void* A, int64_t lda, const Mat& m, int M, void* C, int64_t ldc, int out_f32, int act,
				  const float* residual, hipStream_t s, int gn_T = 0) {
	GemmParams g = {};
	g.nseg = 1; g.seg[0] = {A, lda, 0, 0};
	g.W = m.w; g.ldw = m.Kpad; g.M = M; g.N = m.N; g.K = m.Kpad; g.bias = m.bias;
	g.residual = residual; g.ldr = ldc; g.C = C; g.ldc = ldc; g.out_f32 = out_f32; g.act = act;
	if (m.wes == 1) g.out_scale = m.wscale;      // fp8 operands (A is fp8 too: written by gn() / the attention for exactly these matrices)
	want_stats(h, g, gn_T);
	launch_gemm(m.wes == 1 ? DT_FP8 : h->dt, g, s);
}

// k=3 'same' conv over rows inside each batch element: tap j multiplies row t + j - 1
static void gemm_conv3(ttk_diff* h, const void* A, int64_t lda, const Mat& m, int M, int Tper, void* C, int64_t ldc, int out_f32,
					   const float* residual, int transpose_out, hipStream_t s, int gn_T = 0) {
	GemmParams g = {};
	g.nseg = 3;
	for (int j = 0; j < 3; ++j) g.seg[j] = {A, lda, j - 1, (int64_t)j * m.Npad * m.Kpad};
	g.W = m.w; g.ldw = m.Kpad; g.M = M; g.N = m.N; g.K = m.Kpad; g.rows_per_batch = Tper; g.bias = m.bias;
	g.residual = residual; g.ldr = ldc; g.C = C; g.ldc = ldc; g.out_f32 = out_f32; g.transpose_out = transpose_out;
	if (m.wes == 1) g.out_scale = m.wscale;
	want_stats(h, g, gn_T);
	launch_gemm(m.wes == 1 ? DT_FP8 : h->dt, g, s);
}

static void gn(ttk_diff* h, const float* x, int nb, int T, const float* gamma, const float* beta, const float* scale, const float* shift,
			   int64_t ss_stride, int act, void* out, int out_f32, const int* row_idx, int Tout, hipStream_t s, const Mat* next = nullptr,
			   const float* ms_pre = nullptr) {
	const int C = h->cfg.model_channels;
	if (!ms_pre) {
		if (h->L->ms_owner != (const void*)x) launch_gn_stats(x, nb, T, C, (float*)h->L->ms.p, s, h->tlen);   // else: left by the producing GEMM
		else if (h->need) launch_gn_stats(x, nb, T, C, (float*)h->L->ms.p, s, h->tlen, h->need);           // ... except for the ragged sequences of a batch
		h->L->ms_owner = nullptr;
	}
	GnApplyParams p = {};
	if (h->tlen && Tout == T && !row_idx) { p.tlen = h->tlen; p.chunk_rows = gn_rows_per_chunk(C); }
	p.x = x; p.ms = ms_pre ? ms_pre : (const float*)h->L->ms.p; p.gamma = gamma; p.beta = beta; p.scale = scale; p.shift = shift; p.ss_stride = ss_stride;
	p.row_idx = row_idx; p.nb = nb; p.T = T; p.Tout = Tout; p.C = C; p.nchunks = gn_num_chunks(T, C); p.act = act; p.out = out; p.out_f32 = out_f32;
	if (next && h->prefetch) { p.pf = next->w; p.pf_bytes = (int64_t)next->Npad * next->Kpad * next->wes; p.pf_taps = next->ntap; }
	if (next && next->wes == 1) p.out_f8 = 1;    // the consumer is an fp8 GEMM
	launch_gn_apply(h->dt, p, s);
}

// x (f32 stream, in place) = x + proj_out(attention(qkv(GN(x))))         arch_utils.py:183-190
static void attn_block(ttk_diff* h, const AttnBlk& A, float* x, int nb, int T, hipStream_t s) {
	const int C = h->cfg.model_channels, rows = nb * T;
	gn(h, x, nb, T, A.gn_g, A.gn_b, nullptr, nullptr, 0, ACT_NONE, h->L->a.p, 0, nullptr, T, s, &A.qkv);
	gemm1(h, h->L->a.p, C, A.qkv, rows, h->L->qkv.p, 3 * C, 0, ACT_NONE, nullptr, s);
	AttnParams a = {};
	a.qkv = h->L->qkv.p; a.ld = 3 * C; a.q_off = 0; a.k_off = 64; a.v_off = 128; a.head_stride = 192;   // head-major [H][3][64], arch_utils.py:79
	a.out = h->L->ao.p; a.ldo = C; a.nb = nb; a.T = T; a.H = h->cfg.num_heads; a.causal = 0; a.bias = A.relbias; a.scale = 0.125f; a.tlen = h->tlen;
	if (h->prefetch) { a.pf = A.proj.w; a.pf_bytes = (int64_t)A.proj.Npad * A.proj.Kpad * A.proj.wes; a.pf_taps = 1; }
	a.out_f8 = A.proj.wes == 1;
	launch_attn_fwd(h->dt, a, s);
	gemm1(h, h->L->ao.p, C, A.proj, rows, x, C, 1, ACT_NONE, x, s, T);
}

// x = x + conv3(SiLU(GN(conv1(SiLU(GN(x)))) * (1 + scale) + shift))        diffusion.py:1363-1376
// x_in (with its precomputed GroupNorm statistics ms_in): the block reads its input there and writes x -- the first integrator block of a
// sampler step reads the staged code embedding directly instead of a per-step copy of it
// hf_pre / ms_hf_pre: in_layers(x_in) and its statistics computed beforehand (ttk_diff_begin: the timestep enters a ResBlock only behind them, as
// the scale / shift of the second GroupNorm) -- the block then starts at its second half
static void res_block(ttk_diff* h, const ResBlk& R, float* x, int nb, int T, const float* emb_all, int64_t emb_stride, hipStream_t s,
					  const float* x_in = nullptr, const float* ms_in = nullptr, const float* hf_pre = nullptr, const float* ms_hf_pre = nullptr) {
	const int C = h->cfg.model_channels, rows = nb * T;
	const float* src = x_in ? x_in : x;
	if (!hf_pre) {
		gn(h, src, nb, T, R.gn1_g, R.gn1_b, nullptr, nullptr, 0, ACT_SILU, h->L->a.p, 0, nullptr, T, s, &R.in, ms_in);
		gemm1(h, h->L->a.p, C, R.in, rows, h->L->hf.p, C, 1, ACT_NONE, nullptr, s, T);
	}
	const float* sc = emb_all + (int64_t)R.emb_slot * 2 * C;
	gn(h, hf_pre ? hf_pre : (const float*)h->L->hf.p, nb, T, R.gn2_g, R.gn2_b, sc, sc + C, emb_stride, ACT_SILU, h->L->a.p, 0, nullptr, T, s, &R.out3, ms_hf_pre);
	gemm_conv3(h, h->L->a.p, C, R.out3, rows, T, x, C, 1, src, 0, s, T);
}

static int reserve_lane(ttk_diff* h, Lane& L, int nb, int T) {
	const size_t rows = (size_t)nb * T, C = h->cfg.model_channels, es = h->es;
	TTK_TRY(L.hf.reserve(rows * C * 4)); TTK_TRY(L.a.reserve(rows * C * es)); TTK_TRY(L.qkv.reserve(rows * 3 * C * es)); TTK_TRY(L.ao.reserve(rows * C * es));
	TTK_TRY(L.ms.reserve((size_t)nb * 32 * gn_num_chunks(T, (int)C) * 3 * 4));
	return TTK_OK;
}
static int reserve_ws(ttk_diff* h, int nb, int T) {
	TTK_REQUIRE(gn_num_chunks(T, h->cfg.model_channels) <= 64, TTK_E_ARG, "%d frames exceed the GroupNorm chunk table (64 chunks)", T);
	const size_t rows = (size_t)nb * T, C = h->cfg.model_channels, es = h->es;
	h->L = &h->lane[0];
	TTK_TRY(reserve_lane(h, h->lane[0], nb, T));
	TTK_TRY(h->cs.reserve(rows * C * 4)); TTK_TRY(h->xs.reserve(rows * C * 4));
	TTK_TRY(h->h0.reserve(rows * C * es)); TTK_TRY(h->csT.reserve(rows * C * es)); TTK_TRY(h->xcl.reserve(rows * h->in_pad * es));
	TTK_TRY(h->outb.reserve(rows * h->cfg.out_channels * 4)); TTK_TRY(h->ecl.reserve(rows * C * 4));
	return TTK_OK;
}

// time_embed MLP + every ResBlock's emb_layers for n timestep rows -> emb_all f32 [n][n_emb * 2C]    diffusion.py:1549, :1365
static int time_path(ttk_diff* h, const int64_t* t_dev, const int64_t* t_host, int n, hipStream_t s) {
	const int C = h->cfg.model_channels;
	const size_t es = h->es;
	TTK_TRY(h->temb.reserve((size_t)n * C * es)); TTK_TRY(h->e1.reserve((size_t)n * C * es)); TTK_TRY(h->e2.reserve((size_t)n * C * 4));
	TTK_TRY(h->se.reserve((size_t)n * C * es)); TTK_TRY(h->emb_all.reserve((size_t)n * h->n_emb * 2 * C * 4));
	if (t_dev) launch_timestep_embedding(h->dt, t_dev, 0, n, C, h->time_freqs, h->temb.p, s);
	else for (int i = 0; i < n; ++i) launch_timestep_embedding(h->dt, nullptr, t_host[i], 1, C, h->time_freqs, (char*)h->temb.p + (size_t)i * C * es, s);
	gemm1(h, h->temb.p, C, h->time0, n, h->e1.p, C, 0, ACT_SILU, nullptr, s);
	gemm1(h, h->e1.p, C, h->time2, n, h->e2.p, C, 1, ACT_NONE, nullptr, s);
	launch_silu_cast(h->dt, (const float*)h->e2.p, h->se.p, (int64_t)n * C, s);
	gemm1(h, h->se.p, C, h->emb_cat, n, h->emb_all.p, (int64_t)h->n_emb * 2 * C, 1, ACT_NONE, nullptr, s);
	return TTK_OK;
}

// One evaluation = integrator (the three conditioning_timestep_integrator layers on the code-embedding stream `cs`; depends on the timestep
// only, not on x) + body (everything that sees x).  Inputs of the body: h->xcl (T-typed [nb*T][in_pad]) and cs (f32 [nb*T][C]); emb rows at
// emb_all + b * emb_stride.  Output: out f32 [nb][out_channels][T].     diffusion.py:1549-1564
static void integrator(ttk_diff* h, int nb, int T, const float* emb_all, int64_t emb_stride, float* cs, hipStream_t s,
					   const float* cs_in = nullptr, const float* ms_in = nullptr) {
	for (int i = 0; i < 3; ++i) {
		const bool pre = i == 0 && cs_in && h->hf0_valid;
		res_block(h, h->integrator[i].res, cs, nb, T, emb_all, emb_stride, s, i == 0 ? cs_in : nullptr, i == 0 ? ms_in : nullptr,
				  pre ? (const float*)h->hf0.p : nullptr, pre ? (const float*)h->ms_hf0.p : nullptr);
		attn_block(h, h->integrator[i].attn, cs, nb, T, s);
	}
}
static void body(ttk_diff* h, int nb, int T, const float* emb_all, int64_t emb_stride, const float* cs, float* out, hipStream_t s,
				 hipEvent_t cs_consumed = nullptr) {
	const int C = h->cfg.model_channels, rows = nb * T;
	float* x = (float*)h->xs.p;
	gemm_conv3(h, h->xcl.p, h->in_pad, h->inp_block, rows, T, h->h0.p, C, 0, nullptr, 0, s);
	launch_cast(h->dt, cs, h->csT.p, (int64_t)rows * C, s);
	if (cs_consumed) (void)hipEventRecord(cs_consumed, s);     // the last reader of `cs`: the side stream may overwrite it from here on
	{   // integrating_conv over cat([h0, code_emb], channels): two K segments of one [C][2C] matrix
		GemmParams g = {};
		g.nseg = 2;
		g.seg[0] = {h->h0.p, C, 0, 0};
		g.seg[1] = {h->csT.p, C, 0, C};
		g.W = h->integ.w; g.ldw = h->integ.Kpad; g.M = rows; g.N = C; g.K = C; g.bias = h->integ.bias;
		g.C = x; g.ldc = C; g.out_f32 = 1;
		want_stats(h, g, T);
		launch_gemm(h->dt, g, s);
	}
	for (size_t i = 0; i < h->layers.size(); ++i) {
		res_block(h, h->layers[i].res, x, nb, T, emb_all, emb_stride, s);
		attn_block(h, h->layers[i].attn, x, nb, T, s);
	}
	for (int i = 0; i < 3; ++i) res_block(h, h->tail[i], x, nb, T, emb_all, emb_stride, s);
	gn(h, x, nb, T, h->out_g, h->out_b, nullptr, nullptr, 0, ACT_SILU, h->L->a.p, 0, nullptr, T, s);
	gemm_conv3(h, h->L->a.p, C, h->out_conv, rows, T, out, 0, 1, nullptr, 1, s);
}
static void network(ttk_diff* h, int nb, int T, const float* emb_all, int64_t emb_stride, float* out, hipStream_t s,
					const float* cs_in = nullptr, const float* ms_in = nullptr) {
	integrator(h, nb, T, emb_all, emb_stride, (float*)h->cs.p, s, cs_in, ms_in);
	body(h, nb, T, emb_all, emb_stride, (const float*)h->cs.p, out, s);
}

static int load_attn(ttk_diff* h, const WeightMap& wm, const std::string& p, AttnBlk* A) {
	const int C = h->cfg.model_channels;
	TTK_TRY(upload_f32(h->arena, wm, p + "norm.weight", C, &A->gn_g));
	TTK_TRY(upload_f32(h->arena, wm, p + "norm.bias", C, &A->gn_b));
	// The q / k / v projection stays in the handle's 16-bit type in the fp8 modes, weights and activation operand alike (round 6): e4m3's three significand bits
	// on q and k move scores of +-100 by whole units -- the peaked-attention regime of a trained checkpoint lost 37-42 % of an evaluation to it, against
	// 12 % for bf16 (profiles/r05_stress_errors.json) -- while this GEMM is a seventh of a block's flops.
	TTK_TRY(upload_mat(h->arena, wm, h->dt, p + "qkv.weight", p + "qkv.bias", PK_NK, 3 * C, C, false, &A->qkv));
	TTK_TRY(upload_mat(h->arena, wm, h->wdt, p + "proj_out.weight", p + "proj_out.bias", PK_NK, C, C, false, &A->proj));
	TTK_TRY(upload_f32(h->arena, wm, p + "__relbias", (int64_t)h->cfg.num_heads * 129, &A->relbias));
	return TTK_OK;
}
static int load_res(ttk_diff* h, const WeightMap& wm, const std::string& p, ResBlk* R, int slot) {
	const int C = h->cfg.model_channels;
	TTK_TRY(upload_f32(h->arena, wm, p + "in_layers.0.weight", C, &R->gn1_g));
	TTK_TRY(upload_f32(h->arena, wm, p + "in_layers.0.bias", C, &R->gn1_b));
	TTK_TRY(upload_f32(h->arena, wm, p + "out_layers.0.weight", C, &R->gn2_g));
	TTK_TRY(upload_f32(h->arena, wm, p + "out_layers.0.bias", C, &R->gn2_b));
	TTK_TRY(upload_mat(h->arena, wm, h->wdt, p + "in_layers.2.weight", p + "in_layers.2.bias", PK_NK, C, C, false, &R->in));
	TTK_TRY(upload_mat(h->arena, wm, h->wdt, p + "out_layers.3.weight", p + "out_layers.3.bias", PK_CONV3, C, C, false, &R->out3));
	R->emb_slot = slot;
	return TTK_OK;
}

extern "C" {

int ttk_diff_create(ttk_diff** out, const ttk_diff_config* cfg, const ttk_weight_view* w, int n_w) {
	TTK_REQUIRE(out && cfg && w, TTK_E_ARG, "ttk_diff_create: null argument");
	gemm_roles_refresh();
	TTK_REQUIRE(cfg->model_channels % 64 == 0 && cfg->num_heads * 64 == cfg->model_channels, TTK_E_ARG,
				"ttk_diff_create: head_dim must be 64 (channels %d, heads %d)", cfg->model_channels, cfg->num_heads);
	TTK_REQUIRE(cfg->in_latent_channels % 64 == 0, TTK_E_ARG, "ttk_diff_create: in_latent_channels %% 64 != 0");
	TTK_REQUIRE(cfg->model_channels % 128 == 0 && cfg->model_channels <= 1024 && 1024 % cfg->model_channels == 0, TTK_E_ARG,
				"ttk_diff_create: model_channels %d unsupported (128, 256, 512 or 1024)", cfg->model_channels);
	TTK_REQUIRE(cfg->dtype == TTK_F32 || cfg->dtype == TTK_BF16 || cfg->dtype == TTK_F16 || cfg->dtype == TTK_FP8W || cfg->dtype == TTK_FP8, TTK_E_ARG, "ttk_diff_create: bad dtype %d", cfg->dtype);
	TTK_REQUIRE(cfg->dtype != TTK_FP8 || cfg->model_channels % 128 == 0, TTK_E_ARG, "ttk_diff_create: fp8 GEMMs need channels %% 128 == 0");
	ttk_diff* h = new ttk_diff();
	h->cfg = *cfg;
	h->wdt = cfg->dtype;              // ResBlock / AttentionBlock GEMM weights: rounded to fp8-e4m3 in DT_FP8W (held exactly in bf16)
	h->dt = kernel_dtype(cfg->dtype);
	h->a8 = cfg->dtype == TTK_FP8;
	h->es = dtype_size(h->dt);
	h->in_pad = round_up(cfg->in_channels, 64);
	h->fuse_stats = getenv("TTK_NO_FUSED_GN") ? 0 : 1;
	{ const char* e = getenv("TTK_DIFF_PREFETCH"); h->prefetch = e ? atoi(e) : 1; }
	{ const char* e = getenv("TTK_DIFF_PIPE"); h->pipe = e ? atoi(e) : 1; }
	const int C = cfg->model_channels;
	WeightMap wm(w, n_w);
	int rc = TTK_OK;
	auto fail = [&](int code) { h->arena.release(); delete h; return code; };
#define D_TRY(expr) do { rc = (expr); if (rc != TTK_OK) return fail(rc); } while (0)
	D_TRY(upload_f32(h->arena, wm, "unconditioned_embedding", C, &h->uncond));
	D_TRY(upload_f32(h->arena, wm, "__time_freqs", C / 2, &h->time_freqs));
	D_TRY(upload_mat(h->arena, wm, h->dt, "inp_block.weight", "inp_block.bias", PK_CONV3, C, cfg->in_channels, false, &h->inp_block));
	D_TRY(upload_mat(h->arena, wm, h->dt, "time_embed.0.weight", "time_embed.0.bias", PK_NK, C, C, false, &h->time0));
	D_TRY(upload_mat(h->arena, wm, h->dt, "time_embed.2.weight", "time_embed.2.bias", PK_NK, C, C, false, &h->time2));
	D_TRY(upload_f32(h->arena, wm, "code_norm.weight", C, &h->code_g));
	D_TRY(upload_f32(h->arena, wm, "code_norm.bias", C, &h->code_b));
	D_TRY(upload_mat(h->arena, wm, h->dt, "latent_conditioner.0.weight", "latent_conditioner.0.bias", PK_CONV3, C, cfg->in_latent_channels, false, &h->lat_conv));
	for (int i = 0; i < 4; ++i) D_TRY(load_attn(h, wm, "latent_conditioner." + std::to_string(i + 1) + ".", &h->lat_attn[i]));
	D_TRY(upload_mat(h->arena, wm, h->dt, "integrating_conv.weight", "integrating_conv.bias", PK_NK, C, 2 * C, false, &h->integ));
	D_TRY(upload_f32(h->arena, wm, "out.0.weight", C, &h->out_g));
	D_TRY(upload_f32(h->arena, wm, "out.0.bias", C, &h->out_b));
	D_TRY(upload_mat(h->arena, wm, h->dt, "out.2.weight", "out.2.bias", PK_CONV3, cfg->out_channels, C, false, &h->out_conv));
	int slot = 0;
	for (int i = 0; i < 3; ++i) {
		const std::string p = "conditioning_timestep_integrator." + std::to_string(i) + ".";
		D_TRY(load_res(h, wm, p + "resblk.", &h->integrator[i].res, slot++));
		D_TRY(load_attn(h, wm, p + "attn.", &h->integrator[i].attn));
	}
	h->layers.resize(cfg->num_layers);
	for (int i = 0; i < cfg->num_layers; ++i) {
		const std::string p = "layers." + std::to_string(i) + ".";
		D_TRY(load_res(h, wm, p + "resblk.", &h->layers[i].res, slot++));
		D_TRY(load_attn(h, wm, p + "attn.", &h->layers[i].attn));
	}
	for (int i = 0; i < 3; ++i) D_TRY(load_res(h, wm, "layers." + std::to_string(cfg->num_layers + i) + ".", &h->tail[i], slot++));
	h->n_emb = slot;
	D_TRY(h->arena.alloc((void**)&h->d_tlen, 128 * sizeof(int)));
	D_TRY(h->arena.alloc((void**)&h->d_need, 128 * sizeof(int)));
	// all emb_layers.1 linears stacked into one [n_emb * 2C][C] matrix ("__emb_cat.*", built by the Python packer)
	D_TRY(upload_mat(h->arena, wm, h->dt, "__emb_cat.weight", "__emb_cat.bias", PK_NK, h->n_emb * 2 * C, C, false, &h->emb_cat));
#undef D_TRY
	hipError_t e = hipDeviceSynchronize();
	if (e != hipSuccess) { set_error("ttk_diff_create: %s", hipGetErrorString(e)); return fail(TTK_E_HIP); }
	*out = h;
	return TTK_OK;
}

int ttk_diff_destroy(ttk_diff* h) {
	if (!h) return TTK_OK;
	(void)hipDeviceSynchronize();
	WsBuf* all[] = {&h->cs, &h->cs2, &h->xs, &h->h0, &h->csT, &h->xcl, &h->outb, &h->ecl, &h->temb, &h->e1, &h->e2, &h->se, &h->emb_all, &h->lat_T, &h->ms_ecl, &h->hf0, &h->ms_hf0};
	for (WsBuf* b : all) b->release();
	for (Lane& L : h->lane) { L.a.release(); L.hf.release(); L.qkv.release(); L.ao.release(); L.ms.release(); }
	if (h->side) (void)hipStreamDestroy(h->side);
	for (hipEvent_t e : {h->ev_fork, h->ev_int[0], h->ev_int[1], h->ev_free[0], h->ev_free[1]}) if (e) (void)hipEventDestroy(e);
	h->arena.release();
	delete h;
	return TTK_OK;
}

int ttk_diff_precompute(ttk_diff* h, const float* latents, const float* cond, const int32_t* interp_idx, int b, int M, int T,
						float* E_out, void* stream) {
	TTK_REQUIRE(h && latents && cond && interp_idx && E_out, TTK_E_ARG, "ttk_diff_precompute: null argument");
	TTK_REQUIRE(b >= 1 && M >= 1 && T >= 1, TTK_E_ARG, "ttk_diff_precompute: empty input (b=%d M=%d T=%d)", b, M, T);
	const int C = h->cfg.model_channels, Cl = h->cfg.in_latent_channels;
	hipStream_t s = (hipStream_t)stream;
	TTK_TRY(reserve_ws(h, b, M > T ? M : T));
	h->staged = 0;   // ecl is reused below
	TTK_TRY(h->lat_T.reserve((size_t)b * M * Cl * h->es));
	float* x = (float*)h->xs.p;
	launch_cast(h->dt, latents, h->lat_T.p, (int64_t)b * M * Cl, s);   // latents are already [b][M][Cl] = channels-last
	gemm_conv3(h, h->lat_T.p, Cl, h->lat_conv, b * M, M, x, C, 1, nullptr, 0, s, M);
	for (int i = 0; i < 4; ++i) attn_block(h, h->lat_attn[i], x, b, M, s);
	// code_norm(x) * (1 + scale) + shift, then nearest-neighbour expansion M -> T      diffusion.py:1492,1498,1507
	gn(h, x, b, M, h->code_g, h->code_b, cond, cond + C, 2 * C, ACT_NONE, h->ecl.p, 1, interp_idx, T, s);
	launch_cl_to_cf((const float*)h->ecl.p, b, C, T, E_out, s);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

int ttk_diff_forward(ttk_diff* h, const float* x, const int64_t* t, const float* E, int b, int T, float* out, void* stream) {
	TTK_REQUIRE(h && x && t && out, TTK_E_ARG, "ttk_diff_forward: null argument");
	TTK_REQUIRE(b >= 1 && T >= 1, TTK_E_ARG, "ttk_diff_forward: empty input");
	const int C = h->cfg.model_channels;
	hipStream_t s = (hipStream_t)stream;
	TTK_TRY(reserve_ws(h, b, T));
	h->staged = 0;
	if (E) launch_cf_to_cl(DT_F32, E, b, C, T, h->cs.p, C, 1, s);
	else launch_bcast_rows(DT_F32, h->uncond, b * T, C, h->cs.p, s);            // diffusion.py:1534
	launch_cf_to_cl(h->dt, x, b, h->cfg.in_channels, T, h->xcl.p, h->in_pad, 1, s);
	TTK_TRY(time_path(h, t, nullptr, b, s));
	network(h, b, T, (const float*)h->emb_all.p, (int64_t)h->n_emb * 2 * C, out, s);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

int ttk_diff_begin(ttk_diff* h, const float* E, int b, int T, void* stream) {
	TTK_REQUIRE(h && E, TTK_E_ARG, "ttk_diff_begin: null argument");
	TTK_REQUIRE(b >= 1 && T >= 1, TTK_E_ARG, "ttk_diff_begin: empty input");
	const int C = h->cfg.model_channels;
	hipStream_t s = (hipStream_t)stream;
	TTK_TRY(reserve_ws(h, 2 * b, T));
	float* ecl = (float*)h->ecl.p;
	launch_cf_to_cl(DT_F32, E, b, C, T, ecl, C, 1, s);
	launch_bcast_rows(DT_F32, h->uncond, b * T, C, ecl + (size_t)b * T * C, s);
	// the staged embedding is the same in every step: its GroupNorm statistics once, here
	TTK_TRY(h->ms_ecl.reserve((size_t)2 * b * 32 * gn_num_chunks(T, C) * 3 * 4));
	launch_gn_stats(ecl, 2 * b, T, C, (float*)h->ms_ecl.p, s, h->tlen);
	h->cur_b = b; h->cur_T = T; h->staged = 1;
	// ... and so is the first half of the first integrator ResBlock on it, conv1x1(SiLU(GN(ecl))), with the statistics its second GroupNorm needs: the
	// same launches a step would run (so the same bits), once per utterance instead of once per step
	static const int hoist = [] { const char* e = getenv("TTK_DIFF_HOIST"); return e ? atoi(e) : 1; }();
	h->hf0_valid = 0;
	if (hoist) {
		const int nb = 2 * b, nch = gn_num_chunks(T, C);
		const size_t ms_bytes = (size_t)nb * 32 * nch * 3 * 4;
		TTK_TRY(h->hf0.reserve((size_t)nb * T * C * 4)); TTK_TRY(h->ms_hf0.reserve(ms_bytes));
		const ResBlk& R = h->integrator[0].res;
		gn(h, ecl, nb, T, R.gn1_g, R.gn1_b, nullptr, nullptr, 0, ACT_SILU, h->L->a.p, 0, nullptr, T, s, &R.in, (const float*)h->ms_ecl.p);
		gemm1(h, h->L->a.p, C, R.in, nb * T, h->hf0.p, C, 1, ACT_NONE, nullptr, s, T);
		if (h->L->ms_owner == (const void*)h->hf0.p) {      // statistics left by the GEMM's epilogue (gn() would take them from there) ...
			TTK_HIP(hipMemcpyAsync(h->ms_hf0.p, h->L->ms.p, ms_bytes, hipMemcpyDeviceToDevice, s));
			if (h->need) launch_gn_stats((const float*)h->hf0.p, nb, T, C, (float*)h->ms_hf0.p, s, h->tlen, h->need);   // ... except for the ragged sequences of a batch
		} else {
			launch_gn_stats((const float*)h->hf0.p, nb, T, C, (float*)h->ms_hf0.p, s, h->tlen);
		}
		h->L->ms_owner = nullptr;
		h->hf0_valid = 1;
	}
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

// the part of a sampler step that sees x: layout change, network body on the integrated code stream `cs`, the DDIM / ancestral update
// staged: the previous step of this loop has already written this step's channels-last copy of x (its update launch did); next: the step that follows in this loop, or null --
// its copy is then written by THIS step's update launch (round 6: one layout launch per loop instead of one per step; not for ragged batches, whose padding frames must read zero)
static int step_body(ttk_diff* h, float* x, const ttk_step* st, const float* noise, const float* emb_row, const float* cs, hipEvent_t cs_consumed, hipStream_t s,
					 bool staged = false, const ttk_step* next = nullptr) {
	const int b = h->cur_b, T = h->cur_T;
	const bool cf = st->cfk >= 0.f;
	const int nb = cf ? 2 * b : b;
	if (!staged || h->tlen) launch_cf_to_cl(h->dt, x, b, h->cfg.in_channels, T, h->xcl.p, h->in_pad, cf ? 2 : 1, s, h->tlen);
	float* out = (float*)h->outb.p;
	body(h, nb, T, emb_row, 0, cs, out, s, cs_consumed);
	StepCoefs k = {};
	k.sqrt_recip_ac = st->sqrt_recip_ac; k.sqrt_recipm1_ac = st->sqrt_recipm1_ac; k.sqrt_ac_prev = st->sqrt_ac_prev;
	k.sqrt_1m_ac_prev = st->sqrt_1m_ac_prev; k.cfk = st->cfk; k.coef1 = st->coef1; k.coef2 = st->coef2;
	k.min_log = st->min_log; k.max_log = st->max_log; k.sampler = st->sampler; k.nonzero = st->nonzero;
	const int Cin = h->cfg.in_channels;
	if (next && !h->tlen && (next->cfk >= 0.f) == cf) launch_diffusion_step(      // (same batch layout in the next step: its padding columns are already zero)
		out, out + (size_t)b * h->cfg.out_channels * T, x, noise, b, Cin, T, k, s, h->xcl.p, h->in_pad, next->cfk >= 0.f ? 2 : 1, elem_kind(h->dt));
	else launch_diffusion_step(out, out + (size_t)b * h->cfg.out_channels * T, x, noise, b, Cin, T, k, s);
	return TTK_OK;
}
static int step_impl(ttk_diff* h, float* x, const ttk_step* st, const float* noise, const float* emb_row, hipStream_t s) {
	const int nb = st->cfk >= 0.f ? 2 * h->cur_b : h->cur_b;
	integrator(h, nb, h->cur_T, emb_row, 0, (float*)h->cs.p, s, (const float*)h->ecl.p, (const float*)h->ms_ecl.p);
	return step_body(h, x, st, noise, emb_row, (const float*)h->cs.p, nullptr, s);
}

int ttk_diff_step(ttk_diff* h, float* x, const ttk_step* st, const float* noise, void* stream) {
	TTK_REQUIRE(h && x && st, TTK_E_ARG, "ttk_diff_step: null argument");
	TTK_REQUIRE(h->staged, TTK_E_STATE, "ttk_diff_step: call ttk_diff_begin first");
	TTK_REQUIRE(st->sampler == 0 || noise, TTK_E_ARG, "ttk_diff_step: the p sampler needs noise");
	TTK_REQUIRE(h->cfg.out_channels == 2 * h->cfg.in_channels, TTK_E_ARG, "ttk_diff_step: learned-range output needs out = 2 * in channels");
	hipStream_t s = (hipStream_t)stream;
	TTK_TRY(time_path(h, nullptr, &st->t, 1, s));
	TTK_TRY(step_impl(h, x, st, noise, (const float*)h->emb_all.p, s));
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

// The whole sampler loop, both samplers: steps[n-1], ..., steps[0]; `noise` (ancestral sampler) holds one [b, in, T] draw per step in the order
// the loop consumes them (the j-th executed step reads block j), null for ddim.
static int sample_loop(ttk_diff* h, float* x, const float* E, int b, int T, const ttk_step* steps, int n_steps, const float* noise, int sampler, void* stream, const char* who) {
	TTK_REQUIRE(h && x && E && steps && n_steps >= 1, TTK_E_ARG, "%s: bad argument", who);
	TTK_REQUIRE(sampler == 0 || noise, TTK_E_ARG, "%s: the p sampler needs one noise block per step", who);
	TTK_REQUIRE(h->cfg.out_channels == 2 * h->cfg.in_channels, TTK_E_ARG, "%s: learned-range output needs out = 2 * in channels", who);
	const size_t nz = (size_t)b * h->cfg.in_channels * T;
	hipStream_t s = (hipStream_t)stream;
	TTK_TRY(ttk_diff_begin(h, E, b, T, stream));
	// the timestep-only work of ALL steps in one pass: [n_steps] rows through time_embed + every emb_layers (weights read once)
	std::vector<int64_t> ts(n_steps);
	for (int i = 0; i < n_steps; ++i) { ts[i] = steps[i].t; TTK_REQUIRE(steps[i].sampler == sampler, TTK_E_ARG, "%s: step %d has sampler %d", who, i, steps[i].sampler); }
	TTK_TRY(time_path(h, nullptr, ts.data(), n_steps, s));
	const int64_t stride = (int64_t)h->n_emb * 2 * h->cfg.model_channels;
	const float* emb_all = (const float*)h->emb_all.p;
	if (!h->pipe || n_steps < 2) {
		for (int i = n_steps - 1; i >= 0; --i) TTK_TRY(step_impl(h, x, &steps[i], noise ? noise + (size_t)(n_steps - 1 - i) * nz : nullptr, emb_all + i * stride, s));
		TTK_HIP(hipGetLastError());
		return TTK_OK;
	}
	// Pipelined over two streams.  The integrator of a step depends on its timestep and the staged embedding only, never on x: the one of
	// step j+1 runs on the side stream (own scratch lane, the other `cs` buffer) while the body of step j runs here.  Both are chains of
	// small dependent launches that leave most of the chip idle, so they overlap; one fork and one join edge per step.
	const int nb_max = 2 * b;
	if (!h->side) {
		TTK_HIP(hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
		TTK_HIP(hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming));
		for (int i = 0; i < 2; ++i) {
			TTK_HIP(hipEventCreateWithFlags(&h->ev_int[i], hipEventDisableTiming));
			TTK_HIP(hipEventCreateWithFlags(&h->ev_free[i], hipEventDisableTiming));
		}
	}
	TTK_TRY(reserve_lane(h, h->lane[1], nb_max, T));
	TTK_TRY(h->cs2.reserve((size_t)nb_max * T * h->cfg.model_channels * 4));
	float* csb[2] = {(float*)h->cs.p, (float*)h->cs2.p};
	TTK_HIP(hipEventRecord(h->ev_fork, s));
	TTK_HIP(hipStreamWaitEvent(h->side, h->ev_fork, 0));
	auto enqueue_integrator = [&](int j) {      // j-th step of the loop = schedule index n_steps - 1 - j
		const int i = n_steps - 1 - j;
		const int nb = steps[i].cfk >= 0.f ? 2 * b : b;
		h->L = &h->lane[1];
		integrator(h, nb, T, emb_all + i * stride, 0, csb[j & 1], h->side, (const float*)h->ecl.p, (const float*)h->ms_ecl.p);
		h->L = &h->lane[0];
		(void)hipEventRecord(h->ev_int[j & 1], h->side);
	};
	enqueue_integrator(0);
	for (int j = 0; j < n_steps; ++j) {
		if (j + 1 < n_steps) {
			if (j >= 1) TTK_HIP(hipStreamWaitEvent(h->side, h->ev_free[(j - 1) & 1], 0));   // body j-1 has read the buffer integrator j+1 writes
			enqueue_integrator(j + 1);
		}
		TTK_HIP(hipStreamWaitEvent(s, h->ev_int[j & 1], 0));
		const int i = n_steps - 1 - j;
		const bool staged = j > 0 && (steps[i + 1].cfk >= 0.f) == (steps[i].cfk >= 0.f);      // the previous step's update launch wrote this step's copy (see step_body)
		TTK_TRY(step_body(h, x, &steps[i], noise ? noise + (size_t)j * nz : nullptr, emb_all + i * stride, csb[j & 1], h->ev_free[j & 1], s, staged, i > 0 ? &steps[i - 1] : nullptr));
	}
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

int ttk_diff_sample_ddim(ttk_diff* h, float* x, const float* E, int b, int T, const ttk_step* steps, int n_steps, void* stream) {
	return sample_loop(h, x, E, b, T, steps, n_steps, nullptr, 0, stream, "ttk_diff_sample_ddim");
}

// Several utterances of DIFFERENT length as one batch (no reference counterpart: the reference diffuses one line at a time, inference.py:237-422;
// its network is batch-capable, diffusion.py:1517-1574, and its ramped conditioning-free guidance asserts b = 1 only because it indexes t[0], :391-393).
// Element e occupies a slot of Tp frames of which tlen[e] are real: x [b, in, Tp], E [b, C, Tp], padding frames ignored on input and left
// undefined on output.  Every kernel that looks across frames takes the element's own length -- attention masks keys beyond it and skips
// query blocks beyond it, GroupNorm statistics cover exactly its frames, chunked from its first frame, the k = 3 convolutions read zeros beyond
// its last frame -- and everything else is row-wise, so element e comes out BIT FOR BIT as ttk_diff_sample_ddim(b = 1, T = tlen[e]) gives it,
// while every GEMM of a step runs over the rows of all elements (2 b Tp instead of 2 T: more than one tile per CU).
int ttk_diff_sample_ddim_lines(ttk_diff* h, float* x, const float* E, int b, int Tp, const int* tlen, const ttk_step* steps, int n_steps, void* stream) {
	TTK_REQUIRE(h && x && E && tlen && steps, TTK_E_ARG, "ttk_diff_sample_ddim_lines: null argument");
	TTK_REQUIRE(b >= 1 && 2 * b <= 64, TTK_E_ARG, "ttk_diff_sample_ddim_lines: %d elements (1..32)", b);
	TTK_REQUIRE(Tp >= 64 && Tp % 64 == 0, TTK_E_ARG, "ttk_diff_sample_ddim_lines: the slot length %d must be a multiple of 64 frames", Tp);
	const int C = h->cfg.model_channels;
	for (int i = 0; i < n_steps; ++i) TTK_REQUIRE(steps[i].cfk >= 0.f, TTK_E_ARG, "ttk_diff_sample_ddim_lines: step %d has no conditioning-free evaluation (the batch is laid out as [cond | uncond])", i);
	int host_len[128], host_need[128], any_need = 0;
	for (int e = 0; e < b; ++e) {
		TTK_REQUIRE(tlen[e] >= 1 && tlen[e] <= Tp, TTK_E_ARG, "ttk_diff_sample_ddim_lines: element %d has %d frames, slot %d", e, tlen[e], Tp);
		// a batch of its own length would take its statistics from the GEMM epilogues iff gemm_fuses_gn_stats says so for that length
		const int nd = !(h->fuse_stats && gemm_fuses_gn_stats(2 * tlen[e], C, C, tlen[e]));
		host_len[e] = host_len[b + e] = tlen[e];
		host_need[e] = host_need[b + e] = nd;
		any_need |= nd;
	}
	hipStream_t s = (hipStream_t)stream;
	TTK_HIP(hipMemcpyAsync(h->d_tlen, host_len, (size_t)2 * b * sizeof(int), hipMemcpyHostToDevice, s));
	TTK_HIP(hipMemcpyAsync(h->d_need, host_need, (size_t)2 * b * sizeof(int), hipMemcpyHostToDevice, s));
	TTK_HIP(hipStreamSynchronize(s));      // the staging arrays live on this stack frame
	h->tlen = h->d_tlen;
	h->need = any_need ? h->d_need : nullptr;
	const int rc = sample_loop(h, x, E, b, Tp, steps, n_steps, nullptr, 0, stream, "ttk_diff_sample_ddim_lines");
	h->tlen = nullptr; h->need = nullptr;
	return rc;
}

int ttk_diff_sample_p(ttk_diff* h, float* x, const float* E, int b, int T, const ttk_step* steps, int n_steps, const float* noise, void* stream) {
	return sample_loop(h, x, E, b, T, steps, n_steps, noise, 1, stream, "ttk_diff_sample_p");
}

}  // extern "C"
