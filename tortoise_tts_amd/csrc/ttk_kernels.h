// Host-side launchers of the libttk kernels (internal C++ interface between the handles in ar.hip / diff.hip
// and the kernel files).  All pointers are device pointers; every launcher only enqueues work on `stream`
// (no allocation, no synchronisation: graph-capture safe).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttk {

// DT_FP8W (handles only): bf16 activations and arithmetic, GEMM weights rounded to fp8-e4m3 with a power-of-two per-tensor scale; the
// decode GEMVs stream the weights as fp8 bytes.  Kernels are launched with DT_BF16 plus a per-matrix flag.
// DT_FP8 (diffusion handle only): as DT_FP8W, and the ResBlock / AttentionBlock GEMMs take their ACTIVATION operand in fp8-e4m3 as well
// (written by the GroupNorm-apply and attention kernels) and run on v_mfma_f32_16x16x32_fp8_fp8; launch_gemm accepts DT_FP8 for those.
// DT_F16: the bf16 design with fp16 operands (v_mfma_f32_16x16x32_f16); autoregressive and diffusion handles only.
enum DType { DT_F32 = 0, DT_BF16 = 1, DT_FP8W = 2, DT_FP8 = 3, DT_F16 = 4 };
inline int elem_kind(int dt) { return dt == DT_F32 ? 1 : (dt == DT_F16 ? 2 : 0); }      // ttk::ElemKind of a "T-typed" buffer
inline size_t dtype_size(int dt) { return dt == DT_F32 ? 4 : 2; }
inline int kernel_dtype(int dt) { return (dt == DT_FP8W || dt == DT_FP8) ? DT_BF16 : dt; }

// ---------------------------------------------------------------- per-kernel timing (ttk_host.hip)
// When enabled (ttk_prof_begin) every launcher brackets its launch with two HIP events on the launch stream and adds its
// ALGORITHMIC work (flop for MFMA-bound kernels, bytes for HBM-bound ones) to its kind's tally; bench.py turns the
// tallies into the `roofline` object.  Off by default: zero cost and graph-capture safe.
enum ProfKind { PROF_GEMM = 0, PROF_SKINNY = 1, PROF_ATTN_FWD = 2, PROF_ATTN_DECODE = 3, PROF_GN_STATS = 4, PROF_GN_APPLY = 5,
				PROF_LAYERNORM = 6, PROF_KINDS = 7 };
extern bool g_prof_on;
void prof_start(int kind, double work, hipStream_t s);
void prof_stop(hipStream_t s);
// For launchers that pass the pair to hipExtLaunchKernelGGL: the events then carry the kernel's own start / stop timestamps (what a
// kernel trace reports) instead of bracketing the launch on the stream, which for a 5 us kernel also times the launch gap.
void prof_pair(int kind, double work, hipEvent_t* start, hipEvent_t* stop);
struct ProfScope {
	hipStream_t s; bool on;
	ProfScope(int kind, double work, hipStream_t s_) : s(s_), on(g_prof_on) { if (on) prof_start(kind, work, s); }
	~ProfScope() { if (on) prof_stop(s); }
};

// ---------------------------------------------------------------- dense GEMM (gemm.hip)
// C[M,N] = epilogue( sum_seg  shift(A_seg)[M,K] * W_seg[N,K]^T )      ("NT": both operands K-contiguous)
// Segments express (a) the taps of a 'same' convolution over rows (row shift (tap - (k-1)/2) * dilation inside each batch element
// of `rows_per_batch` rows, zero outside), (b) channel concatenation (two A sources) and (c) the taps one output phase of a
// transposed convolution sees (output rows interleaved through ldc).
struct GemmSeg {
	const void* A;      // T [M][lda]
	int64_t lda;
	int shift;          // source row = m + shift (valid iff it stays inside m's batch element)
	int64_t w_off;      // element offset of this segment's [Npad][K] matrix inside W
};
struct GemmParams {
	// (the fields every launch reads sit in the first cache lines of the argument block; the segment table comes last)
	int nseg;
	int M, N, K;        // K (per segment) % 64 == 0
	const void* W;      // T, each segment matrix [Npad][ldw] row-major (first K columns used), Npad % 128 == 0
	int64_t ldw;
	int rows_per_batch; // >0 when any shift != 0 or transpose_out
	int act;            // ttk::Act
	const float* bias;  // [N] or null
	const float* residual;  // f32 [M][ldr] or null (may alias C when out_f32)
	int64_t ldr;
	void* C;
	int64_t ldc;
	float out_scale;    // 0 = none; else the accumulators are multiplied by it before the bias (the fp8 weights' power-of-two tensor scale)
	int out_f32;        // C is f32 (else T; bf16 when the operands are fp8)
	int transpose_out;  // C is f32 [M / rows_per_batch][N][rows_per_batch]
	// optional fused GroupNorm32 statistics of the f32 output (see gemm_fuses_gn_stats): part[b][32][gn_T / 64][3]
	int gn_T; float* gn_part;
	int seg_inner;      // set by launch_gemm: the segments are the taps -1 / 0 / +1 of a k = 3 convolution over one tensor -- tile order tap-inner (k-chunk 0 of every tap, k-chunk 1 ...), as the CONV role runs it
	int m_major;        // XCD-aware tile order: 0 = each XCD gets a few n-tiles x all m-tiles (its L2 keeps a weight slice), 1 = a few m-tiles x all n-tiles
	// filled by launch_gemm for the role-specialised instantiations (GemmRole): reciprocals that replace the kernel's run-time integer divisions
	// (x / d as an f32 product + one correction step, exact for x < 2^23) -- by the m-tile count, rows_per_batch and gn_T
	int tiles_m; float inv_tiles_m, inv_rpb, inv_gn_T;
	int mix_full, mix_fm;      // k_gemm_mixed: full tiles (= mix_fm tile rows x 16) in front of the half-height tiles of the remaining rows
#ifdef TTK_STAMPS
	unsigned long long* stamps;   // diagnostic build only
#endif
	GemmSeg seg[12];    // up to 11 taps of a dilated convolution (BigVGAN AMP blocks), 3 for the diffusion convs, 2 for a concat
};
// Roles of the DDIM loop's four GEMMs at model_channels = 1024, 16-bit operands.  A role fixes at COMPILE time what the generic kernel reads
// from its arguments and branches on: K = lda = ldw = 1024, N, the segment table (one segment, or the three row-shifted taps of a k = 3 convolution over one
// activation tensor), the epilogue form (bias always, no activation, no scale; f32 or T-typed output; residual; GroupNorm statistics) and the tile order.
// M and the frames per batch element stay run-time values.  launch_gemm picks the role from the parameters; anything else runs the generic kernel.
enum GemmRole { GR_NONE = 0, GR_IN1x1 = 1, GR_CONV3_RES = 2, GR_QKV = 3, GR_PROJ_RES = 4 };
void launch_gemm(int dt, const GemmParams& p, hipStream_t s);
void gemm_roles_refresh();      // re-reads TTK_GEMM_ROLE (handle creation)
// true when launch_gemm(M, N) picks a 128-row tile, i.e. each wave owns 64 rows x one or two whole 32-channel groups and can emit
// the (count, mean, M2) triple of its block in the epilogue (needs N % 64 == 0, 32 channels per group, T % 64 == 0)
bool gemm_fuses_gn_stats(int M, int N, int C, int T);

// ---------------------------------------------------------------- skinny GEMM for decode (skinny.hip)
// out[M<=16*MT, N] = epilogue( LN?(x)[M,K] * W[N,K]^T + bias ), weights streamed once from HBM in MFMA-fragment
// order: Wp[n_tile][k_step][lane][8].
enum SkinnyMode { SK_STORE_F32 = 0, SK_RESIDUAL = 1, SK_ACT_T = 2, SK_QKV = 3 };
struct SkinnyParams {
	const void* Wp;
	int N, K, M;
	const float* bias;
	// A source: LN mode (ln_count 1|2) reads f32 rows and normalises; plain mode reads T rows
	int ln_count;
	const float* x;  int64_t ldx;
	// LN mode: the affine parameters.  Plain mode with g1 != null = "folded LayerNorm": A holds the UN-normalised rows, Wp = gamma o W, g1 =
	// colsum of the T-typed Wp, bias = b + beta W; the epilogue computes (acc - mean * g1[n]) * rstd + bias from the rows' statistics,
	// which the waves gather from the A fragments they multiply anyway (needs a_frag, no narrow / ksplit)
	const float *g1, *b1, *g2, *b2;
	float* ln_out;              // optional f32 [M][K]: the normalised rows (block 0 writes), else null
	const void* a;   int64_t lda;
	int mode, act;
	float* out_f32;  int64_t ldc;   // SK_STORE_F32 / SK_RESIDUAL (in place +=)
	void* out_T;                    // SK_ACT_T, [M][N]; SK_RESIDUAL (optional): a T-typed copy of the updated rows in A-fragment order
	                                // [m_tile][N/32][lane][8], the operand of the folded-LayerNorm launch that follows
	// SK_QKV: n in [0,3d): q -> qbuf[m][n] f32 (pre-scaled), k/v -> cache[m][h][*pos][64]
	// SK_STORE_F32 with qbuf != null (mel head): qbuf[m][n] (row stride ldc) receives the Exp(1) noise torch.multinomial would draw for logit
	// (m, n) -- `slab` then points to a device RngArgs and `tickets` to the per-row int64 draw counters (ttk_rng.h)
	float* qbuf; void* kcache; void* vcache; const int* d_pos; int max_ctx, H; float q_scale;
	// optional split-K over workgroups: slab f32 [n_tiles][ksplit][MT][256], tickets int [n_tiles] (zero between launches)
	int ksplit; float* slab; int* tickets;
	// narrow mode (plain A only, excludes ksplit): N/4 workgroups of 4 columns each instead of N/16 of 16 -- for the N = d projections
	int narrow;
	// activations in MFMA-fragment order [m_tile][k_step][lane][8] (a wave reads 1 KiB contiguous instead of 16 row segments): a_frag = the
	// plain-mode A operand is stored that way, out_frag = SK_ACT_T writes its output that way (for the next launch's a_frag)
	int a_frag, out_frag;
	// fp8 weights: Wp holds one byte per element in the same fragment order; the accumulated product is multiplied by wscale (a power of two)
	int w8; float wscale;
#ifdef TTK_STAMPS
	unsigned long long* stamps;   // diagnostic build only
#endif
};
void launch_skinny(int dt, const SkinnyParams& p, int waves, hipStream_t s);

// ---------------------------------------------------------------- lean decode GEMV (gemv.hip)
// The launches of the KV-cached decode step at the benchmarked geometry (K = 1024 / 4096, whole batch, activations in fragment order, LayerNorm
// folded), one compile-time specialisation per role; bit-identical to k_skinny on the same operands.  launch_gemv returns false (and launches
// nothing) for a geometry it has no instantiation for: the caller then takes launch_skinny.
enum GemvRole { GV_QKV = 0, GV_PROJ = 1, GV_FC = 2, GV_HEAD = 3 };
// Sixteen-row tiles a decode launch is INSTANTIATED for when the batch has M rows (k_gemv / k_skinny: MT = 1, 2, 3 or 4; round 4 added 3: three lines of
// 16 candidates as one decode batch ran the 4-tile form, a quarter of its MFMA and operand traffic on zero rows).  A launch requests EVERY tile of
// its instantiation from the fragment-order operands, so the buffers behind them must hold 16 * decode_row_tiles(max_batch) rows: ttk_ar_create sizes
// them with this function and every decode entry re-checks it (round 3's fault was a create-time size derived from max_batch alone).
inline int decode_row_tiles(int M) { return M <= 16 ? 1 : (M <= 32 ? 2 : (M <= 48 ? 3 : 4)); }

struct GemvParams {
	const void* Wp;             // weights in fragment order [n_tile][K/32][lane][8] (GV_QKV / GV_FC: gamma o W)
	const void* a;              // rows in A-fragment order [m_tile][K/32][lane][8], T-typed, rows >= M zero (GV_QKV / GV_FC: un-normalised)
	const float* bias;          // [N] (GV_QKV / GV_FC: b + beta W)
	const float* csum;          // GV_QKV / GV_FC: column sums of the T-typed folded matrix
	float* out_f32;             // GV_PROJ: residual stream f32 [M][N], updated in place; GV_HEAD: logits [M][N]
	void* out_T;                // GV_PROJ (optional) / GV_FC: T-typed output in A-fragment order [m_tile][N/32][lane][8]
	float* qbuf; void* kcache; void* vcache;   // GV_QKV: q f32 [M][d] pre-scaled, caches T [M][H][max_ctx][64]
	const int* d_pos;           // GV_QKV: cache row to append at; GV_HEAD (optional): incremented by the launch
	float* noise; const void* rng; const int64_t* draws;   // GV_HEAD (optional): Exp(1) noise rows [M][N], device RngArgs, per-row draw counters
	int* health;                // GV_QKV / GV_FC (optional): device word that collects what the folded LayerNorm cannot represent well -- bit 0: a row
	                            // with |mean| > 8 std (the T-typed un-normalised operand then spends > 3 of its bits on the common offset that the
	                            // norm removes), bit 1: non-finite row statistics (an f16 operand above 65504)
	int M, N, max_ctx, H, row0; // row0: first row of this batch inside the noise tensor's row numbering
	float q_scale, wscale;
	int K, w8;                  // host side only (dispatch): K in {1024, 4096}; w8: Wp holds fp8 bytes (GV_PROJ, bf16 arithmetic)
#ifdef TTK_STAMPS
	unsigned long long* stamps; // diagnostic build only
#endif
};
bool launch_gemv(int dt, int role, const GemvParams& p, hipStream_t s);

// ---------------------------------------------------------------- norms (norm.hip)
// y = LN2?(LN1(x)) per row; out is T or f32; frag: out in the skinny GEMV's A-fragment order instead of row-major; out2 (optional): also f32 [rows][d]
void launch_layernorm(int dt, const float* x, int64_t ldx, int rows, int d, const float* g1, const float* b1,
					  const float* g2, const float* b2, void* out, int64_t ldo, int out_f32, hipStream_t s, int frag = 0, float* out2 = nullptr,
					  const int64_t* out2_idx = nullptr, int64_t out2_stride = 0, const int64_t* out2_base = nullptr);
// out2_idx: out2 += out2_idx[0] * out2_stride on the device; out2_base: out2 itself is read from that device word first
// GroupNorm32 over channels-last x f32 [nb][T][C], 32 groups: per-chunk statistics part[nb][32][nchunks][3] = (count, mean, M2)
int gn_num_chunks(int T, int C);
int gn_rows_per_chunk(int C);
// Ragged batches (sequences of different length sharing one row stride T; tlen = device int [nb], null = all T rows valid): the statistics
// of sequence b cover its first tlen[b] rows, chunked from its first row exactly as a batch of its own would be.  need (optional, device
// int [nb]): only sequences with need[b] != 0 are computed -- the others keep the triples a GEMM epilogue left for them.
void launch_gn_stats(const float* x, int nb, int T, int C, float* part, hipStream_t s, const int* tlen = nullptr, const int* need = nullptr);
// y[nb][Tout][C] (T-typed or f32) = act( gn(x)*gamma+beta [ *(1+scale[b][c]) + shift[b][c] ] ), optional nearest
// row gather (row_idx[Tout] into [0,T)) used by timestep_independent's F.interpolate.
struct GnApplyParams {
	const float* x; const float* ms; const float* gamma; const float* beta;
	const float* scale; const float* shift; int64_t ss_stride;   // per-batch stride of scale/shift rows (0 = shared)
	const int* row_idx; int nb, T, Tout, C; int nchunks; int act; void* out; int out_f32;
	const int* tlen; int chunk_rows;   // ragged batch (needs Tout == T, no row_idx): sequence b has tlen[b] valid rows -- its statistics are its first
	                                   // ceil(tlen[b] / chunk_rows) chunk triples, its rows [tlen[b], T) are written as ZEROS (the k = 3 convs' edge padding)
	int out_f8;        // out is fp8-e4m3 bytes (operand of an fp8 GEMM); overrides out_f32
	// optional: weights of the GEMM that consumes this output, touched so that they sit in L2 when it starts.  The matrix is `pf_taps`
	// blocks of `pf_bytes` each; block bytes are split into 8 equal slices, slice x = the n-range the GEMM's tile order gives XCD x, and
	// the workgroups on XCD x (blockIdx % 8) touch one 128-byte line per thread of their share of slice x.
	const void* pf; int64_t pf_bytes; int pf_taps;
#ifdef TTK_STAMPS
	unsigned long long* stamps;   // diagnostic build only
#endif
};
void launch_gn_apply(int dt, const GnApplyParams& p, hipStream_t s);

// ---------------------------------------------------------------- attention (attn.hip)
struct AttnParams {
	const void* qkv; int64_t ld;          // T [nb*T][ld]
	int q_off, k_off, v_off, head_stride; // column of (h, d) = off + h*head_stride + d ; head_dim is 64
	void* out; int64_t ldo;               // T [nb*T][ldo], column h*64 + d
	int out_f8;                           // out is fp8-e4m3 bytes instead of T (operand of an fp8 GEMM; bf16 kernel only)
	int nb, T, H, causal;
	const int* tlen;                      // optional, device int [nb]: sequence b has tlen[b] valid rows of the T its slot holds (ragged batch): keys beyond are
	                                      // masked, query blocks beyond are skipped -- every query sees exactly what a batch of its own length would
	const float* bias;                    // [H][129] relative-position bias (already scaled) or null
	float scale;                          // multiplies q.k
	const void* pf; int64_t pf_bytes; int pf_taps;   // optional L2 touch of the following GEMM's weights (see GnApplyParams)
#ifdef TTK_STAMPS
	unsigned long long* stamps;   // diagnostic build only
#endif
};
void launch_attn_fwd(int dt, const AttnParams& p, hipStream_t s);

struct AttnDecodeParams {
	const float* qbuf;        // f32 [B][H*64], already scaled
	const void* kcache; const void* vcache;   // T [B][H][max_ctx][64]
	const int* d_pos;         // keys valid = *d_pos + 1
	int B, H, max_ctx;
	int ctx_hint;             // host-side copy of the key count (profiling only; stale under graph replay)
	void* out;                // T [B][H*64], or (out_frag) MFMA-fragment order [m_tile][H*2][lane][8] for the projection that follows
	int out_frag;
	const int2* row_info;     // null, or [B] {start, first}: several text lines decoded as one batch (ttk_ar_prefill_lines).  The lines' prefixes are
	                          // right-aligned: candidate b's cache begins at row `start` (its line is that much shorter than the longest), and `first`
	                          // = first candidate of its line, the owner of its shared prefix rows.  Selects the kernel variant that reads it.
	int pos_slot_p1;          // 0: the two position words are read through d_pos.  s + 1: d_pos points at slot s of the position line (attn_pos_slot_acquire), a 64-byte
	                          // line at a LINK-TIME address that the kernel requests before its arguments have arrived (one round trip instead of two in front of the keys)
	int shared_rows;          // != 0: cache rows [0, d_pos[1]) are identical for every candidate (one conditioning latent + one text line: the
	                          // prefill computed the same prefix B times): read them from candidate 0's slice, which the 16 workgroups of a head
	                          // -- equal blockIdx.x, so one XCD -- then share in L2 instead of fetching B copies from HBM
#ifdef TTK_STAMPS
	unsigned long long* stamps;   // diagnostic build only
#endif
};
void launch_attn_decode(int dt, const AttnDecodeParams& p, hipStream_t s);
// the position line (csrc/attn.hip): up to 8 decode handles keep {valid cache rows, shared-prefix rows} in one 64-byte line of the code object.  acquire returns the slot
// (and the device address of its two words), or -1 when all are taken -- the handle then keeps the words in its own allocation and the kernel reads them through d_pos
int attn_pos_slot_acquire(int** words_out);
void attn_pos_slot_release(int slot);

// copy k/v of a dense qkv buffer [B*S][3d] (GPT-2 order q|k|v, head h at h*64) into the cache rows [0,S)
void launch_kv_scatter(int dt, const void* qkv, int B, int S, int H, void* kcache, void* vcache, int max_ctx, hipStream_t s, int t0 = 0);

// ---------------------------------------------------------------- elementwise (elementwise.hip)
void launch_set_int(int* p, int v, hipStream_t s);
void launch_fill_int(int* p, int v, int n, hipStream_t s);
void launch_fill_int2(int* p, int a, int b, int n, hipStream_t s);      // n pairs {a, b}
void launch_add_int(int* p, int v, hipStream_t s);
// out[r][:] = A[ia[r]][:] + Bt[ib[r]][:]   (f32 tables, f32 out); ia/ib int32 device arrays; Bt may be null
void launch_gather_add(const float* A, const int* ia, const float* Bt, const int* ib, float* out, int rows, int d, hipStream_t s);
// decode-time embedding: out[b] = mel_emb[tok[b]] + mel_pos[*d_pos - pos_bias]
void launch_decode_embed(const float* emb, const int64_t* tok, const float* pos, const int* d_pos, int pos_off, int pos_rows,
						 float* out, int B, int d, hipStream_t s, void* frag = nullptr, int frag_f32 = 0);
void launch_copy_rows(const float* src, int64_t lds_, float* dst, int64_t ldd, int rows, int d, hipStream_t s);
void launch_cast(int dt, const float* src, void* dst, int64_t n, hipStream_t s);
// [nb][C][T] f32  ->  [rep*nb*T][ldo] T, zero padded to ldo columns; the nb*T block is written `rep` times
// tlen (optional, device int [nb]): frames [tlen[b], T) of element b are padding and are written as zero rows (ragged batches)
void launch_cf_to_cl(int dt, const float* src, int nb, int C, int T, void* dst, int64_t ldo, int rep, hipStream_t s, const int* tlen = nullptr);
// [nb*T][C] f32 -> [nb][C][T] f32
void launch_cl_to_cf(const float* src, int nb, int C, int T, float* dst, hipStream_t s);
// rows of a [1][C] f32 vector broadcast to T-typed [rows][C]
void launch_bcast_rows(int dt, const float* vec, int rows, int C, void* dst, hipStream_t s);
// sinusoidal timestep embedding (diffusion.py:1277-1295): t int64 device [n] (or host value when t == null) -> T [n][C]
void launch_timestep_embedding(int dt, const int64_t* t, int64_t t_host, int n, int C, const float* freqs, void* out, hipStream_t s);
// y = T(silu(x)) elementwise (f32 in)
void launch_silu_cast(int dt, const float* x, void* y, int64_t n, hipStream_t s);
// fused DDIM/p epilogue on channel-first f32 buffers; see elementwise.hip
struct StepCoefs {
	float sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac_prev, sqrt_1m_ac_prev, cfk;   // cfk < 0: no cond-free mix
	float coef1, coef2, min_log, max_log; int sampler; int nonzero;            // sampler 0 ddim, 1 p
};
void launch_diffusion_step(const float* out_c, const float* out_u, float* x, const float* noise, int nb, int C, int T,
						   StepCoefs c, hipStream_t s, void* xcl = nullptr, int ldo = 0, int rep = 0, int ekind = 0);      // xcl: also the next step's channels-last operand copy of x (elementwise.hip)

// ---------------------------------------------------------------- packing (pack.hip)
enum PackLayout { PK_NK = 0, PK_KN = 1, PK_CONV3 = 2, PK_CONVK = 3, PK_CONVT = 4 };   // CONVK: src[n][k][ntap]; CONVT: src[k][n][ntap]
// fp8-e4m3 weights (pack.hip): |x| maximum of a device array; in-place x <- dequant(quant(x / s)) * s; fragment-order fp8 bytes of a
// bf16 [Npad][K] matrix whose values are already on the fp8 grid times s
int device_absmax(const float* x, int64_t n, float* out_host);
void launch_fp8_roundtrip(float* x, int64_t n, float scale, hipStream_t s);
void launch_pack_frag_fp8(const void* src_bf16, int Npad, int K, float scale, void* dst, hipStream_t s);
void launch_pack_nk_f8(const float* src, int layout, int N, int K, int Npad, int Kpad, float scale, void* dst, hipStream_t s, int ntap = 0);
float fp8_scale_for(float absmax);
// src f32 -> dst T [ntap][Npad][Kpad] (zero padded)
void launch_pack_nk(int dt, const float* src, int layout, int N, int K, int Npad, int Kpad, void* dst, hipStream_t s, int ntap = 0);
// T [Npad][K] -> fragment order [Npad/16][K/32][64][8]
void launch_pack_frag(int dt, const void* src, int Npad, int K, void* dst, hipStream_t s);
// LayerNorm folded into a decode GEMV (pack.hip): w[k][n] *= gamma[k]; csum[n] = sum_k T-typed w[n][k]; bias'[n] = b[n] + sum_k beta[k] w[k][n]
void launch_scale_kn(float* w, const float* gamma, int K, int N, hipStream_t s);
void launch_rowsum(int dt, const void* w, int64_t ld, int N, int K, float* out, hipStream_t s);
void launch_bias_fold(const float* w, const float* beta, const float* b, int K, int N, float* out, hipStream_t s);

}  // namespace ttk
