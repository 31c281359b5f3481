// ttk_ar: the autoregressive half of the hot path -- UnifiedVoice's GPT-2 stack behind the C ABI of include/ttk.h.
//   prefill / latent pass : token-major dense pipeline (LayerNorm -> MFMA GEMM -> causal flash attention ...)
//   decode step           : 5 fused weight-streaming launches per layer + head, all position-dependent state read
//                           from a device scalar so the step is HIP-graph capturable
// Reference: /root/reference/tortoise_tts/models/unified_voice.py:98-254 (GPT2InferenceModel), :334-668 (UnifiedVoice),
//            HF:models/gpt2/modeling_gpt2.py:144-310,514-634 (GPT-2 block math).
#include "ttk_common.h"
#include "ttk_host.h"

using namespace ttk;

namespace {

struct ARLayer {
	float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
	Mat attn, proj, fc, proj2;
};

// rows of the prefill input: [cond | start_text, text.., stop_text | start_mel]      (unified_voice.py:639-649, :203-211)
// rows of the prefill: [cond | start_text, text.., stop_text | start_mel | prompt tokens (optional: prompted continuation, unified_voice.py:651-656)];
// the mel rows sit at mel positions 0, 1, ..., n_in   (GPT2InferenceModel.forward's first branch, unified_voice.py:203-211)
__global__ void k_build_prefill_emb(const float* cond, int Bc, const int64_t* text, int Tt, int B, int d,
									const float* text_emb, const float* text_pos, const float* mel_emb, const float* mel_pos,
									int start_text, int stop_text, int start_mel, float* out, const int64_t* prompt = nullptr, int prompt_rows = 0, int n_in = 0) {
	const int S = Tt + 4 + n_in;
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int d4 = d / 4;
	if (idx >= (int64_t)B * S * d4) return;
	const int c = (int)(idx % d4) * 4;
	const int64_t row = idx / d4;
	const int b = (int)(row / S), s = (int)(row - (int64_t)b * S);
	float4 v;
	if (s == 0) {
		v = *(const float4*)(cond + (int64_t)(Bc == 1 ? 0 : b) * d + c);
	} else if (s >= Tt + 3) {
		const int j = s - (Tt + 3);   // mel position: 0 = start_mel, j >= 1 = prompt token j - 1
		const int64_t tok = j == 0 ? start_mel : prompt[(int64_t)(prompt_rows == 1 ? 0 : b) * n_in + j - 1];
		const float4 e = *(const float4*)(mel_emb + tok * d + c), w = *(const float4*)(mel_pos + (int64_t)j * d + c);
		v = make_float4(e.x + w.x, e.y + w.y, e.z + w.z, e.w + w.w);
	} else {
		const int j = s - 1;   // position in [start, text.., stop]
		const int64_t tok = j == 0 ? start_text : (j == Tt + 1 ? stop_text : text[j - 1]);
		const float4 e = *(const float4*)(text_emb + tok * d + c), w = *(const float4*)(text_pos + (int64_t)j * d + c);
		v = make_float4(e.x + w.x, e.y + w.y, e.z + w.z, e.w + w.w);
	}
	*(float4*)(out + row * d + c) = v;
}

// rows of the latent pass: [cond | start_text, text.., stop_text | start_mel, codes.., stop_mel]   (unified_voice.py:576-593)
__global__ void k_build_latent_emb(const float* cond, const int64_t* text, int Tt, const int64_t* codes, int M, int B, int d,
								   const float* text_emb, const float* text_pos, const float* mel_emb, const float* mel_pos,
								   int start_text, int stop_text, int start_mel, int stop_mel, float* out) {
	const int S = Tt + M + 5;
	const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
	const int d4 = d / 4;
	if (idx >= (int64_t)B * S * d4) return;
	const int c = (int)(idx % d4) * 4;
	const int64_t row = idx / d4;
	const int b = (int)(row / S), s = (int)(row - (int64_t)b * S);
	float4 v;
	if (s == 0) {
		v = *(const float4*)(cond + (int64_t)b * d + c);
	} else if (s < Tt + 3) {
		const int j = s - 1;
		const int64_t tok = j == 0 ? start_text : (j == Tt + 1 ? stop_text : text[(int64_t)b * Tt + j - 1]);
		const float4 e = *(const float4*)(text_emb + tok * d + c), w = *(const float4*)(text_pos + (int64_t)j * d + c);
		v = make_float4(e.x + w.x, e.y + w.y, e.z + w.z, e.w + w.w);
	} else {
		const int j = s - (Tt + 3);   // position in [start, codes.., stop]
		const int64_t tok = j == 0 ? start_mel : (j == M + 1 ? stop_mel : codes[(int64_t)b * M + j - 1]);
		const float4 e = *(const float4*)(mel_emb + tok * d + c), w = *(const float4*)(mel_pos + (int64_t)j * d + c);
		v = make_float4(e.x + w.x, e.y + w.y, e.z + w.z, e.w + w.w);
	}
	*(float4*)(out + row * d + c) = v;
}

}  // namespace

struct ttk_ar {
	ttk_ar_config cfg;
	int dt;                 // arithmetic type the kernels run in (DT_F32 / DT_BF16 / DT_F16)
	int wdt;                // storage type of the GPT-2 block matrices (== dt, or DT_FP8W)
	size_t es;
	Arena arena;
	std::vector<ARLayer> L;
	float *lnf_g, *lnf_b, *fn_g, *fn_b;
	Mat head;
	float *text_emb, *mel_emb, *mel_pos, *text_pos;
	void *kc, *vc;          // [layers][max_batch][H][max_ctx][64]
	size_t kv_layer_stride; // elements
	int* d_pos;             // device: [0] valid cache rows, [1] rows of the shared prefix -- in the attention's position line when pos_slot >= 0, else at the start of the block that holds row_info
	int pos_slot = -1;
	int2* d_rowinfo;        // per candidate {first cache row, first candidate of its line}: line batches (ttk_ar_prefill_lines), prefixes right-aligned
	int lines_mode = 0;     // the current generation was started by ttk_ar_prefill_lines: the attention launches read d_rowinfo
	float *x, *qbuf;        // decode residual stream / scaled queries [max_batch][d]
	void *attn_out, *hbuf;  // T [max_batch][d], [max_batch][4d]
	void* x_frag = nullptr; // T copy of x in A-fragment order [m_tile][d/32][64][8]: operand of the folded-LayerNorm launches
	int poison = 0;         // TTK_DEBUG_POISON=1: the dense passes' scratch (ws_*) is filled with 0xFF when a prefill / latent pass ends -- nothing may read it afterwards
	int frag_rows = 0;      // rows attn_out / hbuf / x_frag were allocated (and zeroed) for: 16 * decode_row_tiles(max_batch)
	int shared_rows = 0;    // leading cache rows that are identical for all candidates of the current generation (AttnDecodeParams.shared_rows)
	int share_prefix = 1;   // TTK_AR_SHARE_PREFIX=0: every candidate reads its own copy
	// multinomial noise drawn by the mel-head launch (ttk_ar_set_noise): device RngArgs, per-row draw counters, q rows of this handle's candidates
	const int64_t* rng_args = nullptr; const int64_t* rng_draws = nullptr; float* rng_q = nullptr;
	int* d_health = nullptr;  // GemvParams.health of the folded launches (ttk_ar_health reads and clears it)
	float* ring_base = nullptr; const int64_t* ring_idx = nullptr; int64_t ring_stride = 0;      // ttk_ar_set_hidden_ring
	int64_t* d_ring = nullptr;                                                                 // device word holding ring_base: what the (possibly captured) launch reads
	int head_split = 1;     // decode head as LayerNorm launch + plain GEMV (TTK_AR_HEAD_SPLIT=0: norms inside the GEMV)
	int lnfold = 1;         // ln_1 + c_attn and ln_2 + c_fc of the decode step with the LayerNorm folded into the matrix (TTK_AR_LNFOLD=0: LN prologue)
	int lean = 1;           // decode launches on the compile-time-specialised kernels of gemv.hip where one exists (TTK_AR_LEAN=0: k_skinny everywhere)
	float* slab; int* tickets;   // split-K scratch of the mlp.c_proj decode GEMV, one set per row group
	// Waves per workgroup of the decode GEMVs that read their rows in fragment order (TTK_AR_WV_PROJ / _PROJ2 / _LN).  A wave requests its
	// operands in batches of 8 k-steps (bf16; 4 in f32), so K / 32 / waves should be a multiple of that: with 8 waves on K = 1024 every wave
	// owned 4 k-steps and half of its 16 load instructions re-read the last fragment -- on a path that is bound by what a CU can take in.
	// More than 8 waves lose again (16 partial tiles to merge through LDS): 4 / 4 / 8 measured best in the 250-token loop (bf16, B = 16).
	int wv_proj = 4, wv_proj2 = 8;
	int wv_ln = 4;                    // ln_1 + c_attn and ln_2 + c_fc with the LayerNorm folded in; the LN-prologue form (TTK_AR_LNFOLD=0) keeps 8
	int wv_head = 4;                  // waves per workgroup of the ln_f + final_norm + mel_head launch (TTK_AR_WV_HEAD): 513 n-tiles; with 8-wave
	                                  // workgroups two fit a CU (512 slots), so tile 513 ran as a second round on an empty chip; 4-wave ones fit four
	int hfrag = 1;                // MLP activations of the decode step in MFMA-fragment order (TTK_AR_HFRAG=0: row-major)
	int narrow2 = 4;              // same for mlp.c_proj (TTK_AR_NARROW2)
	int narrow = 4;               // c_proj / mlp.c_proj decode GEMVs as 4-column workgroups without split-K (TTK_AR_NARROW=0: 16-column + split-K)
	int nsplit = 1;               // row groups decoded concurrently (TTK_AR_SPLIT; measured slower: 283 -> 340 ms at 2, see below)
	hipStream_t side[3] = {nullptr, nullptr, nullptr};
	hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
	WsBuf ws_x, ws_a, ws_qkv, ws_ao, ws_h;
	int B = 0, P = 0, k = 0, ready = 0;
	int Pmax = 0;           // longest prefix of the batch (capacity checks); == P for one line
};

// Debug aid: per-call scratch must not outlive its call.  A captured token step that kept reading one of these buffers would see
// NaN patterns from here on (tests/test_gpu_reuse.py runs the reuse scenarios under the flag).
static int poison_scratch(ttk_ar* h, hipStream_t s) {
	if (!h->poison) return TTK_OK;
	for (WsBuf* w : {&h->ws_x, &h->ws_a, &h->ws_qkv, &h->ws_ao, &h->ws_h})
		if (w->p) TTK_HIP(hipMemsetAsync(w->p, 0xFF, w->cap, s));
	return TTK_OK;
}

static int dense_forward(ttk_ar* h, float* x, int B, int S, bool write_kv, hipStream_t s, int kv_row0 = 0, int kv_t0 = 0) {
	const int d = h->cfg.model_dim, H = h->cfg.heads, dt = h->dt;
	const int rows = B * S;
	TTK_TRY(h->ws_a.reserve((size_t)rows * d * h->es));
	TTK_TRY(h->ws_qkv.reserve((size_t)rows * 3 * d * h->es));
	TTK_TRY(h->ws_ao.reserve((size_t)rows * d * h->es));
	TTK_TRY(h->ws_h.reserve((size_t)rows * 4 * d * h->es));
	for (int l = 0; l < h->cfg.layers; ++l) {
		const ARLayer& L = h->L[l];
		launch_layernorm(dt, x, d, rows, d, L.ln1_g, L.ln1_b, nullptr, nullptr, h->ws_a.p, d, 0, s);
		GemmParams g = {};
		g.nseg = 1; g.seg[0] = {h->ws_a.p, d, 0, 0};
		g.W = L.attn.w; g.ldw = L.attn.Kpad; g.M = rows; g.N = 3 * d; g.K = d; g.bias = L.attn.bias;
		g.C = h->ws_qkv.p; g.ldc = 3 * d;
		launch_gemm(dt, g, s);
		if (write_kv) {
			const size_t row0 = (size_t)kv_row0 * H * h->cfg.max_ctx * 64;      // first candidate slice written
			char* kc = (char*)h->kc + ((size_t)l * h->kv_layer_stride + row0) * h->es;
			char* vc = (char*)h->vc + ((size_t)l * h->kv_layer_stride + row0) * h->es;
			launch_kv_scatter(dt, h->ws_qkv.p, B, S, H, kc, vc, h->cfg.max_ctx, s, kv_t0);
		}
		AttnParams a = {};
		a.qkv = h->ws_qkv.p; a.ld = 3 * d; a.q_off = 0; a.k_off = d; a.v_off = 2 * d; a.head_stride = 64;
		a.out = h->ws_ao.p; a.ldo = d; a.nb = B; a.T = S; a.H = H; a.causal = 1; a.bias = nullptr; a.scale = 0.125f;
		launch_attn_fwd(dt, a, s);
		g = {};
		g.nseg = 1; g.seg[0] = {h->ws_ao.p, d, 0, 0};
		g.W = L.proj.w; g.ldw = L.proj.Kpad; g.M = rows; g.N = d; g.K = d; g.bias = L.proj.bias;
		g.residual = x; g.ldr = d; g.C = x; g.ldc = d; g.out_f32 = 1;
		launch_gemm(dt, g, s);
		launch_layernorm(dt, x, d, rows, d, L.ln2_g, L.ln2_b, nullptr, nullptr, h->ws_a.p, d, 0, s);
		g = {};
		g.nseg = 1; g.seg[0] = {h->ws_a.p, d, 0, 0};
		g.W = L.fc.w; g.ldw = L.fc.Kpad; g.M = rows; g.N = 4 * d; g.K = d; g.bias = L.fc.bias; g.act = ACT_GELU_NEW;
		g.C = h->ws_h.p; g.ldc = 4 * d;
		launch_gemm(dt, g, s);
		g = {};
		g.nseg = 1; g.seg[0] = {h->ws_h.p, 4 * d, 0, 0};
		g.W = L.proj2.w; g.ldw = L.proj2.Kpad; g.M = rows; g.N = d; g.K = 4 * d; g.bias = L.proj2.bias;
		g.residual = x; g.ldr = d; g.C = x; g.ldc = d; g.out_f32 = 1;
		launch_gemm(dt, g, s);
	}
	return TTK_OK;
}

// ln_f + final_norm + mel_head on the decode rows h->x [B][d]
static void head_launch(ttk_ar* h, int B, float* logits, float* hidden_out, hipStream_t s) {
	SkinnyParams p = {};
	p.Wp = h->head.wfrag; p.N = h->cfg.number_mel_codes; p.K = h->cfg.model_dim; p.M = B; p.bias = h->head.bias;
	p.ln_count = 2; p.x = h->x; p.ldx = h->cfg.model_dim;
	p.g1 = h->lnf_g; p.b1 = h->lnf_b; p.g2 = h->fn_g; p.b2 = h->fn_b; p.ln_out = hidden_out;
	p.mode = SK_STORE_F32; p.out_f32 = logits; p.ldc = h->cfg.number_mel_codes;
	if (h->rng_q) { p.qbuf = h->rng_q; p.slab = (float*)h->rng_args; p.tickets = (int*)h->rng_draws; }
	launch_skinny(h->dt, p, h->cfg.model_dim >= 1024 ? h->wv_head : 4, s);
}

// one row group [r0, r0 + nrows) of a decode step on stream s: 30 x {ln_1+c_attn+KV append, attention, c_proj+res, ln_2+c_fc+gelu,
// mlp.c_proj+res} + ln_f/final_norm/mel_head
static void decode_rows(ttk_ar* h, int r0, int nrows, int gi, float* logits_out, float* hidden_out, hipStream_t s, bool bump_pos) {
	const ttk_ar_config& c = h->cfg;
	const int d = c.model_dim, H = c.heads, dt = h->dt;
	const size_t es = h->es;
	float* x = h->x + (size_t)r0 * d;
	float* qbuf = h->qbuf + (size_t)r0 * d;
	char* attn_out = (char*)h->attn_out + (size_t)r0 * d * es;
	char* hbuf = (char*)h->hbuf + (size_t)r0 * 4 * d * es;
	const size_t kv_row = (size_t)H * c.max_ctx * 64 * es;
	const int wv_small = d >= 1024 ? h->wv_ln : 4;   // waves per workgroup that split K = d (folded LayerNorm: plain operand path)
	const int wv_prologue = d >= 1024 ? 8 : 4;       // LayerNorm prologue form: 8 waves x 2 rows normalise the 16 candidates in one pass
	const bool whole = r0 == 0 && nrows == h->B;      // fragment-order activations exist for the whole batch only
	void* xf = whole ? h->x_frag : nullptr;
	const bool lean = h->lean && whole && h->lnfold && h->hfrag;      // gemv.hip's specialised launches: the whole batch, fragment-order activations
	for (int l = 0; l < c.layers; ++l) {
		const ARLayer& L = h->L[l];
		char* kc = (char*)h->kc + (size_t)l * h->kv_layer_stride * es + r0 * kv_row;
		char* vc = (char*)h->vc + (size_t)l * h->kv_layer_stride * es + r0 * kv_row;
		SkinnyParams p = {};
		p.Wp = L.attn.wfrag; p.w8 = L.attn.w8; p.wscale = L.attn.wscale; p.N = 3 * d; p.K = d; p.M = nrows; p.bias = L.attn.bias;
		const bool fold_qkv = h->lnfold && whole && L.attn.wfrag_fold;
		if (fold_qkv) { p.Wp = L.attn.wfrag_fold; p.w8 = 0; p.bias = L.attn.bias_fold; p.g1 = L.attn.csum; p.a = xf; p.lda = d; p.a_frag = 1; }
		else { p.ln_count = 1; p.x = x; p.ldx = d; p.g1 = L.ln1_g; p.b1 = L.ln1_b; }
		p.mode = SK_QKV; p.qbuf = qbuf; p.kcache = kc; p.vcache = vc; p.d_pos = h->d_pos; p.max_ctx = c.max_ctx; p.H = H; p.q_scale = 0.125f;
		GemvParams gq = {};
		gq.Wp = p.Wp; gq.a = xf; gq.bias = p.bias; gq.csum = p.g1; gq.qbuf = qbuf; gq.kcache = kc; gq.vcache = vc; gq.d_pos = h->d_pos;
		gq.M = nrows; gq.N = 3 * d; gq.K = d; gq.max_ctx = c.max_ctx; gq.H = H; gq.q_scale = 0.125f; gq.health = h->d_health;
		if (!(lean && fold_qkv && launch_gemv(dt, GV_QKV, gq, s))) launch_skinny(dt, p, fold_qkv ? wv_small : wv_prologue, s);
		AttnDecodeParams a = {};
		a.qbuf = qbuf; a.kcache = kc; a.vcache = vc; a.d_pos = h->d_pos; a.pos_slot_p1 = h->pos_slot + 1; a.B = nrows; a.H = H; a.max_ctx = c.max_ctx; a.ctx_hint = h->P + 2 + h->k; a.out = attn_out; a.out_frag = h->hfrag && r0 == 0 && nrows == h->B; a.shared_rows = r0 == 0 && h->share_prefix && h->nsplit == 1;
		a.row_info = h->lines_mode && r0 == 0 ? h->d_rowinfo : nullptr;
		launch_attn_decode(dt, a, s);
		p = {};
		p.Wp = L.proj.wfrag; p.w8 = L.proj.w8; p.wscale = L.proj.wscale; p.N = d; p.K = d; p.M = nrows; p.bias = L.proj.bias; p.a = attn_out; p.lda = d; p.a_frag = h->hfrag && r0 == 0 && nrows == h->B;
		p.mode = SK_RESIDUAL; p.out_f32 = x; p.ldc = d; p.narrow = h->narrow; p.out_T = h->lnfold ? xf : nullptr;
		GemvParams gp = {};
		gp.Wp = p.Wp; gp.a = attn_out; gp.bias = p.bias; gp.out_f32 = x; gp.out_T = p.out_T; gp.M = nrows; gp.N = d; gp.K = d; gp.w8 = p.w8; gp.wscale = p.wscale;
		if (!(lean && p.a_frag && h->narrow == 4 && launch_gemv(dt, GV_PROJ, gp, s))) launch_skinny(dt, p, d >= 1024 ? h->wv_proj : 4, s);
		p = {};
		p.Wp = L.fc.wfrag; p.w8 = L.fc.w8; p.wscale = L.fc.wscale; p.N = 4 * d; p.K = d; p.M = nrows; p.bias = L.fc.bias;
		const bool fold_fc = h->lnfold && whole && L.fc.wfrag_fold;
		if (fold_fc) { p.Wp = L.fc.wfrag_fold; p.w8 = 0; p.bias = L.fc.bias_fold; p.g1 = L.fc.csum; p.a = xf; p.lda = d; p.a_frag = 1; }
		else { p.ln_count = 1; p.x = x; p.ldx = d; p.g1 = L.ln2_g; p.b1 = L.ln2_b; }
		p.mode = SK_ACT_T; p.act = ACT_GELU_NEW; p.out_T = hbuf; p.out_frag = h->hfrag && r0 == 0 && nrows == h->B;
		GemvParams gf = {};
		gf.Wp = p.Wp; gf.a = xf; gf.bias = p.bias; gf.csum = p.g1; gf.out_T = hbuf; gf.M = nrows; gf.N = 4 * d; gf.K = d; gf.health = h->d_health;
		if (!(lean && fold_fc && p.out_frag && launch_gemv(dt, GV_FC, gf, s))) launch_skinny(dt, p, fold_fc ? wv_small : wv_prologue, s);
		p = {};
		p.Wp = L.proj2.wfrag; p.w8 = L.proj2.w8; p.wscale = L.proj2.wscale; p.N = d; p.K = 4 * d; p.M = nrows; p.bias = L.proj2.bias; p.a = hbuf; p.lda = 4 * d; p.a_frag = h->hfrag && r0 == 0 && nrows == h->B;
		p.mode = SK_RESIDUAL; p.out_f32 = x; p.ldc = d; p.out_T = h->lnfold ? xf : nullptr;
		p.narrow = h->narrow2;
		if (d >= 1024 && !h->narrow2) {   // 64 n-tiles x 4 K-slices = 256 workgroups; each row group has its own slab and tickets
			p.ksplit = 4; p.slab = h->slab + (size_t)gi * (d / 16) * 4 * 4 * 256; p.tickets = h->tickets + (size_t)gi * (d / 16);
		}
		GemvParams g2 = {};
		g2.Wp = p.Wp; g2.a = hbuf; g2.bias = p.bias; g2.out_f32 = x; g2.out_T = p.out_T; g2.M = nrows; g2.N = d; g2.K = 4 * d; g2.w8 = p.w8; g2.wscale = p.wscale;
		if (!(lean && p.a_frag && h->narrow2 == 4 && launch_gemv(dt, GV_PROJ, g2, s))) launch_skinny(dt, p, d >= 1024 ? h->wv_proj2 : 4, s);
	}
	SkinnyParams p = {};
	p.Wp = h->head.wfrag; p.N = c.number_mel_codes; p.K = d; p.M = nrows; p.bias = h->head.bias;
	p.mode = SK_STORE_F32; p.out_f32 = logits_out + (size_t)r0 * c.number_mel_codes; p.ldc = c.number_mel_codes;
	// the last launch of the step advances the cache length: every reader of *d_pos (c_attn epilogues, attention) is behind it on the
	// stream, the next reader is the next step -- one 1-thread launch per token less (an otherwise unused field carries the pointer)
	if (bump_pos) p.d_pos = h->d_pos;
	if (h->rng_q) { p.qbuf = h->rng_q + (size_t)r0 * c.number_mel_codes; p.slab = (float*)h->rng_args; p.tickets = (int*)(h->rng_draws + r0); p.max_ctx = r0; }
	float* hid = hidden_out ? hidden_out + (size_t)r0 * d : nullptr;
	if (h->head_split && whole) {
		// ln_f + final_norm ONCE (4 workgroups), the mel head as a plain 513-tile GEMV over the normalised rows in fragment order: with the
		// norms inside the GEMV every one of the 513 workgroups normalised all 16 rows, two passes each -- 17 us for 16.8 MB of weights
		if (h->ring_base && !hid) launch_layernorm(dt, x, d, nrows, d, h->lnf_g, h->lnf_b, h->fn_g, h->fn_b, h->attn_out, d, 0, s, 1, nullptr, h->ring_idx, h->ring_stride, h->d_ring);
		else launch_layernorm(dt, x, d, nrows, d, h->lnf_g, h->lnf_b, h->fn_g, h->fn_b, h->attn_out, d, 0, s, 1, hid);
		p.a = h->attn_out; p.lda = d; p.a_frag = 1;
		GemvParams gh = {};
		gh.Wp = p.Wp; gh.a = h->attn_out; gh.bias = p.bias; gh.out_f32 = p.out_f32; gh.d_pos = p.d_pos; gh.noise = p.qbuf; gh.rng = p.slab; gh.draws = (const int64_t*)p.tickets;
		gh.M = nrows; gh.N = c.number_mel_codes; gh.K = d; gh.row0 = r0;
		if (!(lean && launch_gemv(dt, GV_HEAD, gh, s)))
			launch_skinny(dt, p, 4, s);      // 4-wave workgroups: five fit a CU, so the 513 tiles run as one round (two 8-wave ones fit: 512 slots)
	} else {
		p.ln_count = 2; p.x = x; p.ldx = d;
		p.g1 = h->lnf_g; p.b1 = h->lnf_b; p.g2 = h->fn_g; p.b2 = h->fn_b; p.ln_out = hid;
		launch_skinny(dt, p, d >= 1024 ? h->wv_head : 4, s);
	}
}

extern "C" {

int ttk_ar_create(ttk_ar** out, const ttk_ar_config* cfg, const ttk_weight_view* w, int n_w) {
	TTK_REQUIRE(out && cfg && w, TTK_E_ARG, "ttk_ar_create: null argument");
	TTK_REQUIRE(cfg->model_dim % 64 == 0 && cfg->heads * 64 == cfg->model_dim, TTK_E_ARG,
				"ttk_ar_create: head_dim must be 64 (model_dim %d, heads %d)", cfg->model_dim, cfg->heads);
	TTK_REQUIRE(cfg->model_dim <= 2048, TTK_E_ARG, "ttk_ar_create: model_dim %d > 2048 unsupported", cfg->model_dim);
	TTK_REQUIRE(cfg->dtype == TTK_F32 || cfg->dtype == TTK_BF16 || cfg->dtype == TTK_F16 || cfg->dtype == TTK_FP8W, TTK_E_ARG, "ttk_ar_create: bad dtype %d", cfg->dtype);
	TTK_REQUIRE(cfg->max_batch >= 1 && cfg->max_batch <= (cfg->dtype == TTK_F32 ? 32 : 64), TTK_E_ARG,
				"ttk_ar_create: max_batch %d out of range", cfg->max_batch);
	TTK_REQUIRE(cfg->max_ctx >= 8, TTK_E_ARG, "ttk_ar_create: max_ctx %d too small", cfg->max_ctx);
	ttk_ar* h = new ttk_ar();
	h->cfg = *cfg;
	h->wdt = cfg->dtype;
	h->dt = kernel_dtype(cfg->dtype);
	h->es = dtype_size(h->dt);
	WeightMap wm(w, n_w);
	const int d = cfg->model_dim;
	int rc = TTK_OK;
	auto fail = [&](int code) { attn_pos_slot_release(h->pos_slot); h->arena.release(); delete h; return code; };
#define AR_TRY(expr) do { rc = (expr); if (rc != TTK_OK) return fail(rc); } while (0)
	h->L.resize(cfg->layers);
	for (int l = 0; l < cfg->layers; ++l) {
		const std::string p = "gpt.h." + std::to_string(l) + ".";
		ARLayer& L = h->L[l];
		AR_TRY(upload_f32(h->arena, wm, p + "ln_1.weight", d, &L.ln1_g));
		AR_TRY(upload_f32(h->arena, wm, p + "ln_1.bias", d, &L.ln1_b));
		AR_TRY(upload_f32(h->arena, wm, p + "ln_2.weight", d, &L.ln2_g));
		AR_TRY(upload_f32(h->arena, wm, p + "ln_2.bias", d, &L.ln2_b));
		AR_TRY(upload_mat(h->arena, wm, h->wdt, p + "attn.c_attn.weight", p + "attn.c_attn.bias", PK_KN, 3 * d, d, true, &L.attn));
		AR_TRY(upload_mat(h->arena, wm, h->wdt, p + "attn.c_proj.weight", p + "attn.c_proj.bias", PK_KN, d, d, true, &L.proj));
		AR_TRY(upload_mat(h->arena, wm, h->wdt, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", PK_KN, 4 * d, d, true, &L.fc));
		AR_TRY(upload_mat(h->arena, wm, h->wdt, p + "mlp.c_proj.weight", p + "mlp.c_proj.bias", PK_KN, d, 4 * d, true, &L.proj2));
		{   // (fp8 weights: the fold is taken of the rounded matrices and kept in bf16 -- TTK_FP8W stays "bf16 on the rounded weights")
			AR_TRY(fold_layernorm(h->arena, wm, h->dt, p + "attn.c_attn.weight", p + "attn.c_attn.bias", p + "ln_1.weight", p + "ln_1.bias", &L.attn));
			AR_TRY(fold_layernorm(h->arena, wm, h->dt, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", p + "ln_2.weight", p + "ln_2.bias", &L.fc));
		}
	}
	AR_TRY(upload_f32(h->arena, wm, "gpt.ln_f.weight", d, &h->lnf_g));
	AR_TRY(upload_f32(h->arena, wm, "gpt.ln_f.bias", d, &h->lnf_b));
	AR_TRY(upload_f32(h->arena, wm, "final_norm.weight", d, &h->fn_g));
	AR_TRY(upload_f32(h->arena, wm, "final_norm.bias", d, &h->fn_b));
	AR_TRY(upload_mat(h->arena, wm, h->dt, "mel_head.weight", "mel_head.bias", PK_NK, cfg->number_mel_codes, d, true, &h->head));
	AR_TRY(upload_f32(h->arena, wm, "text_embedding.weight", (int64_t)cfg->number_text_tokens_p1 * d, &h->text_emb));
	AR_TRY(upload_f32(h->arena, wm, "mel_embedding.weight", (int64_t)cfg->number_mel_codes * d, &h->mel_emb));
	AR_TRY(upload_f32(h->arena, wm, "mel_pos_embedding.emb.weight", (int64_t)cfg->max_mel_seq_len * d, &h->mel_pos));
	AR_TRY(upload_f32(h->arena, wm, "text_pos_embedding.emb.weight", (int64_t)cfg->max_text_seq_len * d, &h->text_pos));
	h->kv_layer_stride = (size_t)cfg->max_batch * cfg->heads * cfg->max_ctx * 64;
	AR_TRY(h->arena.alloc(&h->kc, h->kv_layer_stride * cfg->layers * h->es));
	AR_TRY(h->arena.alloc(&h->vc, h->kv_layer_stride * cfg->layers * h->es));
	AR_TRY(h->arena.alloc((void**)&h->d_pos, (size_t)(4 + 2 * cfg->max_batch) * sizeof(int)));
	h->d_rowinfo = (int2*)(h->d_pos + 4);
	h->d_health = h->d_pos + 2;      // (words 2, 3 of the block are otherwise unused; zeroed with it below)
	if (hipMemset(h->d_pos, 0, (size_t)(4 + 2 * cfg->max_batch) * sizeof(int)) != hipSuccess) return fail(TTK_E_HIP);
	{   // the two position words move into the decode attention's position line when a slot is free (csrc/attn.hip); row_info and the health word stay in the block above
		int* words = nullptr;
		h->pos_slot = attn_pos_slot_acquire(&words);
		if (h->pos_slot >= 0) {
			h->d_pos = words;
			if (hipMemset(h->d_pos, 0, 2 * sizeof(int)) != hipSuccess) return fail(TTK_E_HIP);
		}
	}
	AR_TRY(h->arena.alloc((void**)&h->d_ring, sizeof(int64_t)));
	if (hipMemset(h->d_ring, 0, sizeof(int64_t)) != hipSuccess) return fail(TTK_E_HIP);
	AR_TRY(h->arena.alloc((void**)&h->x, (size_t)cfg->max_batch * d * sizeof(float)));
	AR_TRY(h->arena.alloc((void**)&h->qbuf, (size_t)cfg->max_batch * d * sizeof(float)));
	// Fragment-order operands of the decode launches, [m_tile][k-step][lane]: the kernels are instantiated for 1, 2 or 4 sixteen-row tiles (csrc/gemv.hip:
	// gemv_mt) and request EVERY tile of their instantiation, so 33..48 rows read a fourth tile: it must exist (each buffer is its own allocation; a
	// read past its end is a fault whenever the driver has not mapped anything behind it) and hold zeros, like the padding rows inside a tile.
	const size_t frag_rows = (size_t)16 * decode_row_tiles(cfg->max_batch);      // the SAME function the launchers pick their instantiation with
	h->frag_rows = (int)frag_rows;
	for (int B = 1; B <= cfg->max_batch; ++B)      // every batch this handle accepts: the tiles its launches request exist
		if (16 * decode_row_tiles(B) > h->frag_rows) { set_error("ttk_ar_create: %d rows would request %d fragment rows, %d allocated", B, 16 * decode_row_tiles(B), h->frag_rows); return fail(TTK_E_STATE); }
	AR_TRY(h->arena.alloc(&h->attn_out, frag_rows * d * h->es));
	if (hipMemset(h->attn_out, 0, frag_rows * d * h->es) != hipSuccess) return fail(TTK_E_HIP);
	AR_TRY(h->arena.alloc(&h->hbuf, frag_rows * 4 * d * h->es));
	if (hipMemset(h->hbuf, 0, frag_rows * 4 * d * h->es) != hipSuccess) return fail(TTK_E_HIP);
	AR_TRY(h->arena.alloc(&h->x_frag, frag_rows * d * h->es));
	if (hipMemset(h->x_frag, 0, frag_rows * d * h->es) != hipSuccess) return fail(TTK_E_HIP);
	AR_TRY(h->arena.alloc((void**)&h->slab, (size_t)4 * (d / 16) * 4 * 4 * 256 * sizeof(float)));
	AR_TRY(h->arena.alloc((void**)&h->tickets, (size_t)4 * (d / 16) * sizeof(int)));
	if (hipMemset(h->tickets, 0, (size_t)4 * (d / 16) * sizeof(int)) != hipSuccess) return fail(TTK_E_HIP);
	{
		const char* ep = getenv("TTK_DEBUG_POISON");
		h->poison = ep && atoi(ep) != 0;
		const char* en = getenv("TTK_AR_NARROW");
		h->narrow = en ? atoi(en) : 4;                    // 0 = 16-column workgroups (+ split-K for mlp.c_proj), 2 / 4 = workgroups per tile
		const char* ef = getenv("TTK_AR_HFRAG");
		h->hfrag = ef ? (atoi(ef) != 0) : 1;
		const char* en2 = getenv("TTK_AR_NARROW2");
		h->narrow2 = en2 ? atoi(en2) : h->narrow;
		const char* e1 = getenv("TTK_AR_WV_PROJ");
		const char* e2 = getenv("TTK_AR_WV_PROJ2");
		if (e1 && atoi(e1) >= 4 && atoi(e1) <= 16) h->wv_proj = atoi(e1);
		if (h->dt == DT_F32) { h->wv_proj = 8; h->wv_ln = 8; h->wv_proj2 = 8; }      // batches of 4 k-steps: 32 / 4 = 8 waves; K = 4096: 4 batches each
		if (e2 && atoi(e2) >= 4 && atoi(e2) <= 16) h->wv_proj2 = atoi(e2);
		const char* e4 = getenv("TTK_AR_WV_LN");
		if (e4 && atoi(e4) >= 4 && atoi(e4) <= 16) h->wv_ln = atoi(e4);
		const char* e3 = getenv("TTK_AR_WV_HEAD");
		if (e3 && (atoi(e3) == 4 || atoi(e3) == 8)) h->wv_head = atoi(e3);
		const char* es = getenv("TTK_AR_SHARE_PREFIX");
		h->share_prefix = es ? (atoi(es) != 0) : 1;
		const char* eh = getenv("TTK_AR_HEAD_SPLIT");
		h->head_split = eh ? (atoi(eh) != 0) : 1;
		const char* el = getenv("TTK_AR_LNFOLD");
		h->lnfold = el ? (atoi(el) != 0) : 1;
		const char* eg = getenv("TTK_AR_LEAN");
		h->lean = eg ? (atoi(eg) != 0) : 1;
		const char* e = getenv("TTK_AR_SPLIT");
		h->nsplit = e ? atoi(e) : 1;
		if (h->nsplit != 1 && h->nsplit != 2 && h->nsplit != 4) h->nsplit = 1;
		if (hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess) return fail(TTK_E_HIP);
		for (int i = 0; i < 3; ++i) {
			if (hipStreamCreateWithFlags(&h->side[i], hipStreamNonBlocking) != hipSuccess) return fail(TTK_E_HIP);
			if (hipEventCreateWithFlags(&h->ev_join[i], hipEventDisableTiming) != hipSuccess) return fail(TTK_E_HIP);
		}
	}
#undef AR_TRY
	hipError_t e = hipDeviceSynchronize();
	if (e != hipSuccess) { set_error("ttk_ar_create: %s", hipGetErrorString(e)); return fail(TTK_E_HIP); }
	*out = h;
	return TTK_OK;
}

int ttk_ar_destroy(ttk_ar* h) {
	if (!h) return TTK_OK;
	(void)hipDeviceSynchronize();
	for (int i = 0; i < 3; ++i) { if (h->side[i]) (void)hipStreamDestroy(h->side[i]); if (h->ev_join[i]) (void)hipEventDestroy(h->ev_join[i]); }
	if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
	h->ws_x.release(); h->ws_a.release(); h->ws_qkv.release(); h->ws_ao.release(); h->ws_h.release();
	attn_pos_slot_release(h->pos_slot);
	h->arena.release();
	delete h;
	return TTK_OK;
}

static int prefill_impl(ttk_ar* h, const float* cond_latent, int Bc, const int64_t* text, int Tt, const int64_t* prompt, int prompt_rows, int n_in, int B, float* logits_out,
						void* stream, const char* who) {
	TTK_REQUIRE(h && cond_latent && text && logits_out, TTK_E_ARG, "%s: null argument", who);
	const ttk_ar_config& c = h->cfg;
	TTK_REQUIRE(B >= 1 && B <= c.max_batch, TTK_E_ARG, "%s: B=%d exceeds max_batch=%d", who, B, c.max_batch);
	TTK_REQUIRE(Bc == 1 || Bc == B, TTK_E_ARG, "%s: cond batch %d must be 1 or B=%d", who, Bc, B);
	TTK_REQUIRE(Tt >= 1 && Tt + 2 <= c.max_text_seq_len, TTK_E_ARG, "%s: %d text tokens exceed the position table (%d)", who, Tt, c.max_text_seq_len - 2);
	TTK_REQUIRE(n_in == 0 || (prompt && (prompt_rows == 1 || prompt_rows == B)), TTK_E_ARG, "%s: %d prompt tokens need 1 or B=%d prompt rows (got %d)", who, n_in, B, prompt_rows);
	TTK_REQUIRE(n_in >= 0 && n_in + 2 < c.max_mel_seq_len, TTK_E_ARG, "%s: %d prompt tokens exceed the mel position table (%d)", who, n_in, c.max_mel_seq_len);
	const int S = Tt + 4 + n_in, d = c.model_dim;
	TTK_REQUIRE(S + 1 <= c.max_ctx, TTK_E_ARG, "%s: prefix of %d rows does not fit max_ctx=%d", who, S, c.max_ctx);
	hipStream_t s = (hipStream_t)stream;
	// One conditioning latent and one text line for all B candidates (what inference_speech passes): the B prefixes are the same rows, so
	// the prefix is run ONCE and its cache rows live in candidate 0's slice only -- the decode attention reads rows [0, S) there for every
	// candidate (AttnDecodeParams.shared_rows) -- and the last row / the logits are replicated.  (A prompt shared by all candidates is part of that
	// prefix; with one prompt per candidate every candidate's rows are run and cached on their own.)
	const bool shared = Bc == 1 && B > 1 && h->share_prefix && h->nsplit == 1 && (n_in == 0 || prompt_rows == 1);
	const int Bp = shared ? 1 : B;
	TTK_TRY(h->ws_x.reserve((size_t)Bp * S * d * sizeof(float)));
	float* x = (float*)h->ws_x.p;
	const int64_t total = (int64_t)Bp * S * (d / 4);
	hipLaunchKernelGGL(k_build_prefill_emb, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cond_latent, Bc, text, Tt, Bp, d,
					   h->text_emb, h->text_pos, h->mel_emb, h->mel_pos, c.start_text_token, c.stop_text_token, c.start_mel_token, x, prompt, prompt_rows, n_in);
	TTK_TRY(dense_forward(h, x, Bp, S, true, s));
	launch_copy_rows(x + (size_t)(S - 1) * d, shared ? 0 : (int64_t)S * d, h->x, d, B, d, s);      // source stride 0: one row to all candidates
	head_launch(h, B, logits_out, nullptr, s);
	launch_set_int(h->d_pos, S, s);
	launch_set_int(h->d_pos + 1, shared ? S : 0, s);      // rows of the shared prefix (AttnDecodeParams.shared_rows): device-resident, like the cache length
	TTK_TRY(poison_scratch(h, s));
	// k = mel tokens behind start_mel that are cached already: the decode step's capacity and position checks count from it (the position itself is taken
	// from the cache length on the device: cached rows - P, the reference's attention_mask.shape[1] - mel_len)
	h->B = B; h->P = h->Pmax = Tt + 3; h->k = n_in; h->ready = 1; h->lines_mode = 0;
	h->shared_rows = shared ? S : 0;
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

int ttk_ar_prefill(ttk_ar* h, const float* cond_latent, int Bc, const int64_t* text, int Tt, int B, float* logits_out, void* stream) {
	return prefill_impl(h, cond_latent, Bc, text, Tt, nullptr, 0, 0, B, logits_out, stream, "ttk_ar_prefill");
}

int ttk_ar_prefill_prompted(ttk_ar* h, const float* cond_latent, int Bc, const int64_t* text, int Tt, const int64_t* prompt, int prompt_rows, int n_prompt, int B,
							float* logits_out, void* stream) {
	TTK_REQUIRE(n_prompt >= 1 && prompt, TTK_E_ARG, "ttk_ar_prefill_prompted: needs at least one prompt token");
	return prefill_impl(h, cond_latent, Bc, text, Tt, prompt, prompt_rows, n_prompt, B, logits_out, stream, "ttk_ar_prefill_prompted");
}

// Several text lines as ONE decode batch (no reference counterpart: TTS.inference walks its lines one by one, inference.py:244-246, each with
// the weights streamed again for 16 rows): line g occupies candidates [g * rows_per_line, (g + 1) * rows_per_line), its prefix is run once into
// the slice of its first candidate, and the decode step gives every candidate its own cache length -- d_pos + row_off[b] -- so each row's keys
// are dealt to the waves exactly as in a batch of its own and its logits are bit for bit those of ttk_ar_prefill / ttk_ar_decode on that line alone.
int ttk_ar_prefill_lines(ttk_ar* h, const float* cond_latents, const int64_t* text, const int* text_len, int n_lines, int rows_per_line,
						 float* logits_out, void* stream) {
	TTK_REQUIRE(h && cond_latents && text && text_len && logits_out, TTK_E_ARG, "ttk_ar_prefill_lines: null argument");
	const ttk_ar_config& c = h->cfg;
	const int B = n_lines * rows_per_line, d = c.model_dim;
	TTK_REQUIRE(n_lines >= 1 && rows_per_line >= 1 && B <= c.max_batch, TTK_E_ARG, "ttk_ar_prefill_lines: %d lines x %d candidates exceed max_batch=%d", n_lines, rows_per_line, c.max_batch);
	TTK_REQUIRE(h->share_prefix && h->nsplit == 1, TTK_E_STATE, "ttk_ar_prefill_lines: needs the shared-prefix decode (TTK_AR_SHARE_PREFIX=1, TTK_AR_SPLIT=1)");
	int Smax = 0;
	for (int g = 0; g < n_lines; ++g) {
		const int Tt = text_len[g];
		TTK_REQUIRE(Tt >= 1 && Tt + 2 <= c.max_text_seq_len, TTK_E_ARG, "ttk_ar_prefill_lines: line %d: %d text tokens exceed the position table (%d)", g, Tt, c.max_text_seq_len - 2);
		TTK_REQUIRE(Tt + 5 <= c.max_ctx, TTK_E_ARG, "ttk_ar_prefill_lines: line %d: prefix of %d rows does not fit max_ctx=%d", g, Tt + 4, c.max_ctx);
		Smax = Tt + 4 > Smax ? Tt + 4 : Smax;
	}
	hipStream_t s = (hipStream_t)stream;
	TTK_TRY(h->ws_x.reserve((size_t)Smax * d * sizeof(float)));
	float* x = (float*)h->ws_x.p;
	int64_t toff = 0;
	for (int g = 0; g < n_lines; ++g) {
		// prefixes right-aligned at Smax: the cache length is then ONE number for all candidates (the c_attn epilogue appends every row at
		// *d_pos, as for one line) and only the attention needs to know where a candidate's rows begin
		const int Tt = text_len[g], S = Tt + 4, r0 = g * rows_per_line, start = Smax - S;
		const int64_t total = (int64_t)S * (d / 4);
		hipLaunchKernelGGL(k_build_prefill_emb, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cond_latents + (size_t)g * d, 1, text + toff, Tt, 1, d,
						   h->text_emb, h->text_pos, h->mel_emb, h->mel_pos, c.start_text_token, c.stop_text_token, c.start_mel_token, x);
		TTK_TRY(dense_forward(h, x, 1, S, true, s, r0, start));
		launch_copy_rows(x + (size_t)(S - 1) * d, 0, h->x + (size_t)r0 * d, d, rows_per_line, d, s);      // the line's last prefix row to all its candidates
		launch_fill_int2((int*)(h->d_rowinfo + r0), start, r0, rows_per_line, s);
		toff += Tt;
	}
	head_launch(h, B, logits_out, nullptr, s);
	launch_set_int(h->d_pos, Smax, s);
	launch_set_int(h->d_pos + 1, Smax, s);
	TTK_TRY(poison_scratch(h, s));
	h->B = B; h->P = h->Pmax = Smax - 1; h->k = 0; h->ready = 1; h->lines_mode = 1;
	h->shared_rows = Smax;
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

static int decode_impl(ttk_ar* h, const int64_t* tok, float* logits_out, float* hidden_out, void* stream, const char* who) {
	TTK_REQUIRE(h && logits_out, TTK_E_ARG, "%s: null argument", who);
	TTK_REQUIRE(h->ready, TTK_E_STATE, "%s: call ttk_ar_prefill first", who);
	const ttk_ar_config& c = h->cfg;
	const int B = h->B, d = c.model_dim;
	TTK_REQUIRE(h->Pmax + 1 + h->k + 1 <= c.max_ctx, TTK_E_STATE, "%s: KV cache full (max_ctx=%d)", who, c.max_ctx);
	TTK_REQUIRE(B <= c.max_batch && 16 * decode_row_tiles(B) <= h->frag_rows, TTK_E_STATE, "%s: %d rows request %d fragment rows, the handle holds %d (max_batch=%d)", who, B,
				16 * decode_row_tiles(B), h->frag_rows, c.max_batch);
	TTK_REQUIRE((h->lnfold && h->nsplit == 1) || B <= 32, TTK_E_STATE, "%s: the LayerNorm-prologue decode kernels (TTK_AR_LNFOLD=0 / TTK_AR_SPLIT) hold at most 32 candidates' rows in LDS; B=%d", who, B);
	TTK_REQUIRE(h->k + 2 < c.max_mel_seq_len, TTK_E_STATE, "%s: mel position table exhausted (%d rows)", who, c.max_mel_seq_len);
	TTK_REQUIRE(!h->ring_base || (h->head_split && h->nsplit == 1), TTK_E_STATE, "%s: the hidden ring needs the default decode form (TTK_AR_HEAD_SPLIT=1, TTK_AR_SPLIT=1)", who);
	hipStream_t s = (hipStream_t)stream;
	// x[b] = mel_embedding[tok] + mel_pos[k + 1]; *d_pos = P + k rows are cached  =>  offset 1 - P   (unified_voice.py:213-214)
	// (tok == null: ttk_ar_sample_next has written the rows already)
	if (tok) launch_decode_embed(h->mel_emb, tok, h->mel_pos, h->d_pos, 1 - h->P, c.max_mel_seq_len, h->x, B, d, s, h->lnfold ? h->x_frag : nullptr, elem_kind(h->dt));
	// Experiment kept behind TTK_AR_SPLIT (default 1 = off): cut the candidates into row groups whose launch chains run on forked
	// streams.  Rows are independent, so results are unchanged -- but on MI355X it LOSES (B=16, 250 tokens: 283 ms -> 340 ms with
	// 2 groups, 517 ms with 4): the LayerNorm kernels already hold one 8-wave workgroup per CU, so the second chain cannot
	// co-reside and only the doubled weight traffic remains.
	int nsplit = h->nsplit;
	while (nsplit > 1 && B / nsplit < 1) nsplit >>= 1;
	if (nsplit > 1) {
		TTK_HIP(hipEventRecord(h->ev_fork, s));
		for (int gi = 1; gi < nsplit; ++gi) TTK_HIP(hipStreamWaitEvent(h->side[gi - 1], h->ev_fork, 0));
	}
	for (int gi = 0; gi < nsplit; ++gi) {
		hipStream_t gs = gi == 0 ? s : h->side[gi - 1];
		const int r0 = (B * gi) / nsplit, r1 = (B * (gi + 1)) / nsplit;
		decode_rows(h, r0, r1 - r0, gi, logits_out, hidden_out, gs, nsplit == 1);
	}
	for (int gi = 1; gi < nsplit; ++gi) {
		TTK_HIP(hipEventRecord(h->ev_join[gi - 1], h->side[gi - 1]));
		TTK_HIP(hipStreamWaitEvent(s, h->ev_join[gi - 1], 0));
	}
	if (nsplit > 1) launch_add_int(h->d_pos, 1, s);
	h->k += 1;
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

int ttk_ar_decode(ttk_ar* h, const int64_t* tok, float* logits_out, float* hidden_out, void* stream) {
	TTK_REQUIRE(tok, TTK_E_ARG, "ttk_ar_decode: null argument");
	return decode_impl(h, tok, logits_out, hidden_out, stream, "ttk_ar_decode");
}

int ttk_ar_decode_next(ttk_ar* h, float* logits_out, float* hidden_out, void* stream) {
	return decode_impl(h, nullptr, logits_out, hidden_out, stream, "ttk_ar_decode_next");
}

int ttk_ar_sample_next(ttk_ar* h, const ttk_sample_args* a, void* stream) {
	TTK_REQUIRE(h && a, TTK_E_ARG, "ttk_ar_sample_next: null argument");
	TTK_REQUIRE(h->ready, TTK_E_STATE, "ttk_ar_sample_next: call ttk_ar_prefill first");
	TTK_REQUIRE(a->B == h->B && a->V == h->cfg.number_mel_codes, TTK_E_ARG, "ttk_ar_sample_next: shape (B %d, V %d) is not the prefilled one (B %d, V %d)",
				a->B, a->V, h->B, h->cfg.number_mel_codes);
	return launch_sample_step(a, h->mel_emb, h->mel_pos, h->x, h->cfg.model_dim, h->cfg.max_mel_seq_len, h->lnfold ? h->x_frag : nullptr, elem_kind(h->dt),
							  (hipStream_t)stream, "ttk_ar_sample_next");
}

int ttk_ar_set_noise(ttk_ar* h, const int64_t* rng_args, const int64_t* draws, float* q) {
	TTK_REQUIRE(h, TTK_E_ARG, "ttk_ar_set_noise: null handle");
	TTK_REQUIRE((rng_args && draws && q) || (!rng_args && !draws && !q), TTK_E_ARG, "ttk_ar_set_noise: pass all three pointers or none");
	h->rng_args = rng_args; h->rng_draws = draws; h->rng_q = q;
	return TTK_OK;
}

int ttk_ar_set_hidden_ring(ttk_ar* h, float* base, const int64_t* index, int64_t stride, void* stream) {
	TTK_REQUIRE(h, TTK_E_ARG, "ttk_ar_set_hidden_ring: null handle");
	TTK_REQUIRE(!base || (index && stride >= 0), TTK_E_ARG, "ttk_ar_set_hidden_ring: a ring needs its device index and a stride");
	h->ring_base = base; h->ring_idx = base ? index : nullptr; h->ring_stride = base ? stride : 0;
	if (base) {      // the launches read the base from device memory (a captured step is replayed by later generations with their own buffers): stream-ordered upload
		const int64_t v = (int64_t)(uintptr_t)base;
		TTK_HIP(hipMemcpyAsync(h->d_ring, &v, sizeof(v), hipMemcpyHostToDevice, (hipStream_t)stream));
	}
	return TTK_OK;
}

int ttk_ar_decode_geometry(int dtype, int max_batch, int rows, int32_t out[4]) {
	TTK_REQUIRE(out, TTK_E_ARG, "ttk_ar_decode_geometry: null argument");
	TTK_REQUIRE(dtype == TTK_F32 || dtype == TTK_BF16 || dtype == TTK_F16 || dtype == TTK_FP8W || dtype == TTK_FP8, TTK_E_ARG, "ttk_ar_decode_geometry: bad dtype %d", dtype);
	const int cap = dtype == TTK_F32 ? 32 : 64;
	TTK_REQUIRE(max_batch >= 1 && max_batch <= cap, TTK_E_ARG, "ttk_ar_decode_geometry: max_batch %d out of range (1..%d)", max_batch, cap);
	TTK_REQUIRE(rows >= 1 && rows <= max_batch, TTK_E_ARG, "ttk_ar_decode_geometry: rows %d outside 1..max_batch=%d", rows, max_batch);
	out[0] = 16 * decode_row_tiles(max_batch);      // rows ttk_ar_create allocates (and zeroes) per fragment-order operand
	out[1] = decode_row_tiles(rows);                // sixteen-row tiles of the instantiation a launch over `rows` rows runs
	out[2] = 16 * out[1];                           // rows that launch requests
	out[3] = max_batch;                             // candidate slices of the KV cache
	return TTK_OK;
}

int ttk_ar_health(ttk_ar* h, int* flags_out, void* stream) {
	TTK_REQUIRE(h && flags_out, TTK_E_ARG, "ttk_ar_health: null argument");
	hipStream_t s = (hipStream_t)stream;
	TTK_HIP(hipMemcpyAsync(flags_out, h->d_health, sizeof(int), hipMemcpyDeviceToHost, s));
	TTK_HIP(hipStreamSynchronize(s));
	if (*flags_out) launch_set_int(h->d_health, 0, s);
	return TTK_OK;
}

int ttk_ar_last_hidden(ttk_ar* h, float* hidden_out, void* stream) {
	TTK_REQUIRE(h && hidden_out, TTK_E_ARG, "ttk_ar_last_hidden: null argument");
	TTK_REQUIRE(h->ready, TTK_E_STATE, "ttk_ar_last_hidden: call ttk_ar_prefill first");
	const int d = h->cfg.model_dim;
	// h->x holds the residual stream of the newest row after the last block; enc = final_norm(ln_f(x))   (unified_voice.py:106, HF GPT2Model ln_f)
	launch_layernorm(h->dt, h->x, d, h->B, d, h->lnf_g, h->lnf_b, h->fn_g, h->fn_b, hidden_out, d, 1, (hipStream_t)stream);
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

int ttk_ar_latents(ttk_ar* h, const float* cond, const int64_t* text, int Tt, const int64_t* codes, int M, int B, float* latents_out, void* stream) {
	TTK_REQUIRE(h && cond && text && codes && latents_out, TTK_E_ARG, "ttk_ar_latents: null argument");
	const ttk_ar_config& c = h->cfg;
	TTK_REQUIRE(B >= 1 && Tt >= 1 && M >= 1, TTK_E_ARG, "ttk_ar_latents: empty input (B=%d Tt=%d M=%d)", B, Tt, M);
	TTK_REQUIRE(Tt + 2 <= c.max_text_seq_len, TTK_E_ARG, "ttk_ar_latents: %d text tokens exceed the position table", Tt);
	TTK_REQUIRE(M + 2 <= c.max_mel_seq_len, TTK_E_ARG, "ttk_ar_latents: %d mel codes exceed the position table (%d)", M, c.max_mel_seq_len - 2);
	const int S = Tt + M + 5, d = c.model_dim;
	hipStream_t s = (hipStream_t)stream;
	TTK_TRY(h->ws_x.reserve((size_t)B * S * d * sizeof(float)));
	float* x = (float*)h->ws_x.p;
	const int64_t total = (int64_t)B * S * (d / 4);
	hipLaunchKernelGGL(k_build_latent_emb, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cond, text, Tt, codes, M, B, d,
					   h->text_emb, h->text_pos, h->mel_emb, h->mel_pos, c.start_text_token, c.stop_text_token, c.start_mel_token, c.stop_mel_token, x);
	TTK_TRY(dense_forward(h, x, B, S, false, s));
	// enc = final_norm(ln_f(h)); mel rows start at 1 + (Tt + 2); keep the first M of the M + 2   (unified_voice.py:518-522,595)
	for (int b = 0; b < B; ++b)
		launch_layernorm(h->dt, x + ((size_t)b * S + Tt + 3) * d, d, M, d, h->lnf_g, h->lnf_b, h->fn_g, h->fn_b,
						 latents_out + (size_t)b * M * d, d, 1, s);
	TTK_TRY(poison_scratch(h, s));
	TTK_HIP(hipGetLastError());
	return TTK_OK;
}

}  // extern "C"
