/* libttk -- C ABI of the MI355X-native TorToiSe inference hot path.
 *
 * The reference (e-c-k-e-r/tortoise-tts) is pure Python and has no FFI of its own; its plugin idiom is replacing the
 * module objects that `TTS.inference` calls (tortoise_tts/inference.py:185-202).  The entry points
 * below are what a binding for those call sites needs; each cites the reference interface it stands in for.  The
 * Python classes in tortoise_tts_amd/ wrap them under the reference's own method names.
 *
 * Conventions
 *  - return 0 on success, a negative TTK_E_* code on failure; ttk_last_error() gives the text; nothing throws or aborts.
 *  - every tensor argument is a raw DEVICE pointer to a contiguous row-major buffer in the reference's layout
 *    (b x C x T channels-first for the diffusion net, B x S x D for the AR net); the caller owns them.
 *  - `stream` is a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream); calls only enqueue work and never
 *    synchronise, allocate or free (except the create and destroy calls and the first call that grows a workspace), so a decode
 *    step or a diffusion step can be captured into a HIP graph by the caller.
 *  - a handle owns packed weights, KV cache and workspaces; one handle per (process, device); not re-entrant.  Different handles may be
 *    driven from different host threads on different streams (ttk_last_error is per thread).
 *  - weights are passed as f32 tensors (host or device memory) under the reference's state_dict key names and are copied.
 */
#ifndef TTK_H
#define TTK_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TTK_VERSION 1

enum { TTK_OK = 0, TTK_E_ARG = -1, TTK_E_HIP = -2, TTK_E_WEIGHT = -3, TTK_E_STATE = -4 };
/* arithmetic mode: storage/MFMA operand type (accumulation is always f32).  TTK_FP8W (BASELINE config 5) = TTK_BF16 arithmetic with
 * the GEMM weights of the GPT-2 blocks / ResBlocks / AttentionBlocks rounded to fp8-e4m3 (OCP, round to nearest even) times a
 * power-of-two per-tensor scale; the KV-cached decode streams them as fp8 bytes, the dense GEMMs hold the same values in bf16.
 * TTK_FP8 (diffusion handle only) = TTK_FP8W with the ACTIVATION operand of the ResBlock / AttentionBlock GEMMs rounded to fp8-e4m3 as well
 * (scale 1: they are GroupNorm outputs and attention outputs), those GEMMs running on the fp8 MFMA with f32 accumulation: the result is
 * the TTK_BF16 arithmetic applied to operands rounded that way, up to f32 summation order.  Diffusion handle, both fp8 modes (round 6): the
 * AttentionBlocks' q / k / v projection keeps its weights and its activation operand in bf16 -- e4m3 on q and k moves peaked attention scores
 * by whole units -- so "block GEMMs" there means in_layers.2, out_layers.3 and proj_out.
 * TTK_F16 (autoregressive and diffusion handles) = the TTK_BF16 design with IEEE half operands on v_mfma_f32_16x16x32_f16 -- the other dtype the
 * reference's `torch.autocast("cuda", dtype)` at inference.py:331 can be given (config.py:625-637); f32 accumulation, residual streams,
 * norms and softmax as in every mode, so unlike autocast nothing but an operand above 65504 can overflow. */
enum { TTK_F32 = 0, TTK_BF16 = 1, TTK_FP8W = 2, TTK_FP8 = 3, TTK_F16 = 4 };

typedef struct {
	const char* name;     /* reference state_dict key, e.g. "gpt.h.0.attn.c_attn.weight" */
	const float* data;    /* f32, contiguous, host or device */
	int ndim;
	int64_t shape[4];
} ttk_weight_view;

int ttk_version(void);
const char* ttk_last_error(void);   /* thread-local text of the last failure */

/* Per-kernel timing for the roofline report (no reference counterpart: the reference has no profiling of its own,
 * SURVEY.md section 5).  Between begin and end every libttk kernel launch is bracketed by HIP events on its own stream and
 * tallied per kernel kind together with its algorithmic work.  Do not capture graphs while enabled.
 * kinds: 0 gemm (flop), 1 skinny gemm (bytes), 2 attention fwd (flop), 3 decode attention (bytes), 4 groupnorm stats (bytes),
 *        5 groupnorm apply (bytes), 6 layernorm (bytes)                                                                 */
typedef struct { double ms; int64_t launches; double work; } ttk_prof_result;
int ttk_prof_begin(void);
int ttk_prof_end(ttk_prof_result* out, int n_kinds);   /* n_kinds >= 7; synchronises the device */

/* ------------------------------------------------------------------------------------------------------------------
 * Autoregressive model: UnifiedVoice + GPT2InferenceModel (tortoise_tts/models/unified_voice.py:98-254, 334-668).   */
typedef struct ttk_ar ttk_ar;

typedef struct {
	int layers, model_dim, heads;             /* unified_voice.py:337-339 ; head_dim must be 64 */
	int max_mel_seq_len, max_text_seq_len;    /* rows of the two learned position tables (:405-406) */
	int number_text_tokens_p1, number_mel_codes;
	int start_text_token, stop_text_token, start_mel_token, stop_mel_token;
	int dtype;                                /* TTK_F32 | TTK_BF16 | TTK_F16 | TTK_FP8W */
	int max_batch;                            /* candidates decoded together (<= 64; <= 32 in TTK_F32) */
	int max_ctx;                              /* KV-cache rows per sequence: prefix + generated tokens */
} ttk_ar_config;

/* Copies and packs the hot-path subset of UnifiedVoice.state_dict() (tortoise_tts_amd/weights.py: ar_shapes). */
int ttk_ar_create(ttk_ar** out, const ttk_ar_config* cfg, const ttk_weight_view* w, int n_w);
int ttk_ar_destroy(ttk_ar* h);

/* Prefill = first forward of `generate` (unified_voice.py:639-649 prefix build, :203-211 prefill branch, lm_head :239).
 *   cond_latent [Bc, D] f32 (Bc == 1 or B), text [Tt] int64 raw token ids (start/stop are added here, :639-640)
 *   logits_out  [B, number_mel_codes] f32 = logits of the last prefill row.  Resets the KV cache to P + 1 rows.   */
int ttk_ar_prefill(ttk_ar* h, const float* cond_latent, int Bc, const int64_t* text, int Tt, int B,
				   float* logits_out, void* stream);

/* Prefill of a PROMPTED continuation (`inference_speech(..., input_tokens=)`, unified_voice.py:651-656: the mel tokens of `input_tokens` follow start_mel in the
 * first forward, :203-211): prompt = device int64 [prompt_rows][n_prompt] mel codes, prompt_rows == 1 (one prompt for all candidates: it joins the shared
 * prefix) or B.  Mel positions 1 .. n_prompt; the cache then holds P + 1 + n_prompt rows, the logits are those of the last prompt row, and the first
 * decode step feeds its token at mel position n_prompt + 2 (the reference's attention_mask.shape[1] - mel_len).                                          */
int ttk_ar_prefill_prompted(ttk_ar* h, const float* cond_latent, int Bc, const int64_t* text, int Tt, const int64_t* prompt, int prompt_rows, int n_prompt, int B,
							float* logits_out, void* stream);

/* One KV-cached decode step (unified_voice.py:212-214 + HF GPT2Model + lm_head): feeds back tok [B] int64, the k-th
 * generated token (k = 1, 2, ... counted by the handle), with mel position k + 1 (the reference's indexing).
 *   logits_out [B, number_mel_codes] f32;  hidden_out (optional, may be NULL) [B, D] f32 = final_norm(ln_f(h)) of the
 *   new row, what `sample_stream` yields (stream_generator.py:1172).                                               */
int ttk_ar_decode(ttk_ar* h, const int64_t* tok, float* logits_out, float* hidden_out, void* stream);

/* final_norm(ln_f(h)) [B, D] f32 of the newest row: after ttk_ar_prefill the prefix's last row, after a decode step that step's row
 * (valid until the next ttk_ar_sample_next / decode call reuses the row buffer).  `sample_stream` yields this next to the token sampled
 * from the same forward's logits (stream_generator.py:1172) -- for the first token that is the prefill's row.                     */
int ttk_ar_last_hidden(ttk_ar* h, float* hidden_out, void* stream);

/* Host-only (no GPU call; usable before any handle exists): the geometry the decode launches of a handle created with (dtype, max_batch) run with when
 * `rows` candidates are decoded -- out[0] = rows ttk_ar_create allocates and zeroes for each fragment-order operand (attention output, MLP
 * activations, the T-typed residual copy), out[1] = sixteen-row tiles of the kernel instantiation the launch selects, out[2] = rows that
 * instantiation requests (every tile of it, whatever `rows` is), out[3] = candidate slices of the KV cache.  out[2] <= out[0] for every
 * rows <= max_batch is what ttk_ar_create asserts and every decode entry re-checks; tests/test_host_logic.py enumerates it.  No reference
 * counterpart (the reference's tensors are sized by the call that makes them).                                                              */
int ttk_ar_decode_geometry(int dtype, int max_batch, int rows, int32_t out[4]);

/* What the decode step's folded LayerNorm (ln_1 + c_attn, ln_2 + c_fc as one matrix over the UN-normalised rows, T-typed) could not represent
 * well since the last call: bit 0 = a row of the residual stream had |mean| > 8 std (more than 3 of the operand's significand bits went to the
 * common offset the norm removes: results drift from the reference's LayerNorm-then-matmul; TTK_AR_LNFOLD=0 selects the form that normalises
 * in f32 first), bit 1 = non-finite row statistics (an f16 operand above 65504).  Synchronises `stream`, clears the word.  No reference
 * counterpart: the fold is this library's, so is the duty to say when it does not apply.                                                   */
int ttk_ar_health(ttk_ar* h, int* flags_out, void* stream);

/* The same rows for EVERY decode step of a captured token loop, without a per-step pointer: while set, each ttk_ar_decode / ttk_ar_decode_next
 * also writes final_norm(ln_f(h)) [B, D] f32 of its new rows to base + index[0] * stride (elements), `index` a device int64 the sampling launch
 * advances (the `col` of ttk_sample_args: tokens sampled so far), so step n's rows land in slot n of a [slots, B, D] buffer although the
 * captured launch arguments never change -- what the streaming generator (unified_voice.py:670-679, stream_generator.py:1172) yields next to
 * token n.  The base is uploaded to a device word on `stream` and read from there by the launches, so a token step captured during one
 * generation writes into the buffer of whichever generation replays it.  Pass base = NULL to switch it off.  Needs the default decode form
 * (TTK_AR_HEAD_SPLIT=1, TTK_AR_SPLIT=1).                                                                                                 */
int ttk_ar_set_hidden_ring(ttk_ar* h, float* base, const int64_t* index, int64_t stride, void* stream);

/* One sampled token per candidate, the body of HF `_sample` that stream_generator.py drives (warpers :56-101; HF:generation/
 * utils.py:2894-2937): probs = softmax(scores / temperature); next = multinomial(probs, 1) = argmax(probs / q) with q the caller's
 * Exp(1) noise [B, V] (torch `exponential_`, so the generator stream is the reference's); finished rows get `stop_token`;
 * tok[b] = next; ids[b, col[b]] = next (skipped when col[b] >= ids_cols); history[b, hist_off + col[b]] = next (optional);
 * col[b] += 1; unfinished[b] &= next != stop_token.  suppress (optional) is a [V] byte mask of ids forced to -inf before the
 * temperature (SuppressTokensLogitsProcessor); any other warper is applied by the caller, who then passes temperature 1.
 * live_rows / all_done (optional): *live_rows (device int, caller sets it to the number of unfinished rows) is decremented when a
 * row finishes, and the row that brings it to 0 stores the number of tokens sampled so far (col + 1 >= 1: the generation ended WITH that
 * token) to *all_done (device or pinned host int, zeroed by the caller) -- HF's
 * `unfinished_sequences.max() == 0` stopping test without a host round trip per token.  Pointers are device memory unless said
 * otherwise; only enqueues one kernel, so it may be captured in a HIP graph.                                               */
int ttk_sample_step(const float* scores, int64_t ld, int B, int V, const float* q, int64_t ldq,
					const unsigned char* suppress, float temperature, int64_t stop_token, int64_t* unfinished, int64_t* tok, int64_t* ids, int64_t ids_ld, int64_t ids_cols,
					int64_t* col, int64_t* history, int64_t hist_ld, int64_t hist_off, int* live_rows, int* all_done,
					void* stream);

/* ttk_sample_step with the reference's other processors and warpers inside the same kernel, in HF's order (stream_generator.py:56-101 builds
 * Temperature -> TopK -> TopP; HF `_get_logits_processor` puts RepetitionPenaltyLogitsProcessor and SuppressTokensLogitsProcessor in front):
 *   repetition_penalty (HF:generation/logits_process.py RepetitionPenaltyLogitsProcessor): every id that occurs in history[b, 0 : hist_off + col[b]]
 *     (= input_ids: the caller writes the prefix ids into the first hist_off columns, the kernel appends the sampled ones) has its score
 *     multiplied (< 0) or divided (>= 0) by the penalty; 1 or 0 = off;
 *   suppress, temperature: as ttk_sample_step;
 *   top_k (TopKLogitsWarper): scores below the k-th largest become -inf; 0 = off;
 *   top_p (TopPLogitsWarper): ascending cumulative softmax <= 1 - top_p becomes -inf, the largest score always stays; >= 1 or 0 = off.
 * The CLI's defaults (__main__.py:17-21: top-k 16) therefore stay on this one launch: no torch.topk / torch.sort per token, no host
 * round trip, capturable.  top-k / top-p / typical sampling / the penalty need V <= 9216 (the row is held in registers).
 *   typical_mass (TypicalLogitsWarper, unified_voice.py:47-75; what `inference_speech(typical_sampling=True)` adds as a custom processor): tokens in
 *     ascending |-log p - H| are kept until their probability mass reaches typical_mass, the others become -inf; runs after the penalty and the
 *     suppression, before the temperature; 0 or >= 1 = off.                                                                              */
typedef struct {
	const float* scores; int64_t ld; int B, V;
	const float* q; int64_t ldq;              /* Exp(1) noise of torch.multinomial, drawn by the caller */
	const unsigned char* suppress;            /* [V] byte mask or NULL */
	float temperature;                        /* > 0 */
	int top_k; float top_p; float repetition_penalty;
	int64_t stop_token;
	int64_t *unfinished, *tok, *ids; int64_t ids_ld, ids_cols; int64_t* col;
	int64_t* history; int64_t hist_ld, hist_off;   /* optional unless repetition_penalty is on */
	int *live_rows, *all_done;                /* optional, see ttk_sample_step */
	float typical_mass;                       /* TypicalLogitsWarper (unified_voice.py:47-75), applied after the processors and before the temperature,
	                                           * where the reference's custom logits_processor runs; 0 or >= 1 = off */
} ttk_sample_args;
int ttk_sample_step_warped(const ttk_sample_args* a, void* stream);

/* The same launch, additionally writing the input row of the decode step that follows (the first lines of GPT2InferenceModel.forward's
 * cached branch, unified_voice.py:212-214): x[b] = mel_embedding[tok[b]] + mel_pos_embedding[col[b] + 1] with col[b] already counting
 * the new token (the reference's k + 1 indexing).  ttk_ar_decode_next then runs the step from that row: the pair replaces
 * ttk_ar_decode's embedding-gather launch.  Both may be mixed freely with ttk_ar_decode on one handle.                        */
int ttk_ar_sample_next(ttk_ar* h, const ttk_sample_args* a, void* stream);
int ttk_ar_decode_next(ttk_ar* h, float* logits_out, float* hidden_out, void* stream);

/* The Exp(1) noise of torch.multinomial without torch in the token step.  `q.exponential_(1)` on the GPU is a pure function of (generator seed,
 * generator offset, element index, launch geometry) -- ATen's distribution_nullary_kernel over a Philox4_32_10 state (csrc/ttk_rng.h restates
 * the indexing; the state and the uniform conversion are rocRAND's own header code, as in torch's build).  ttk_ar_set_noise makes the mel-head
 * launch of ttk_ar_prefill / ttk_ar_decode[_next] write q[m][n] for its logits: rng_args = device int64[6] {seed, offset before the first draw,
 * threads of torch's launch for the full [C, V] tensor, offset step per draw, first row of this handle's candidates in that tensor, candidates per text line or 0 (ttk_ar_prefill_lines)}, draws =
 * device int64[B] draws made so far per row (the `col` of ttk_sample_args), q = f32 rows [B][V].  The caller advances the torch generator by
 * step x draws afterwards, so everything drawn later is the reference's stream too.  Pass NULLs to switch it off.
 * ttk_exponential_like_torch fills out[li] for li < numel with the same function (draw-th draw): the bitwise comparison against
 * torch.Tensor.exponential_ that a caller runs before relying on it (tortoise_tts_amd/autoregressive.py does).                          */
int ttk_ar_set_noise(ttk_ar* h, const int64_t* rng_args, const int64_t* draws, float* q);
int ttk_exponential_like_torch(float* out, int64_t numel, int64_t seed, int64_t offset0, int64_t threads, int64_t step, int64_t draw, void* stream);

/* Several text lines prefilled as ONE decode batch (no reference counterpart: TTS.inference walks the lines of a text one by one,
 * inference.py:244-246, streaming the weights again for each line's 16 candidates).  Line g: conditioning latent cond_latents[g] (f32 [n_lines][model_dim]),
 * text ids text[sum(text_len[:g]) ...] (device int64, concatenated), text_len[g] of them (HOST int array); it occupies candidates
 * [g * rows_per_line, (g + 1) * rows_per_line).  logits_out f32 [n_lines * rows_per_line][number_mel_codes].  The prefixes are right-aligned in the cache (one cache length for the batch); the decode entries
 * then step all candidates together, the attention reading every candidate's keys from where its line begins, and every row's logits are bit for bit those of ttk_ar_prefill / ttk_ar_decode
 * on its line alone.  With ttk_ar_set_noise, rng_args[5] = rows_per_line makes every line draw the same [rows_per_line, V] noise, as the reference's
 * reseeding to 0 per line does.                                                                                                   */
int ttk_ar_prefill_lines(ttk_ar* h, const float* cond_latents, const int64_t* text, const int* text_len, int n_lines, int rows_per_line,
						 float* logits_out, void* stream);

/* hipGraphLaunch of an instantiated graph the caller captured around libttk launches (torch.cuda.CUDAGraph.raw_cuda_graph_exec()): the
 * token step replayed without torch.cuda.CUDAGraph.replay()'s prologue, which fills the generator's seed / offset tensors -- two launches per
 * token -- whether or not the captured work draws random numbers.  With ttk_ar_set_noise it does not.                                      */
int ttk_graph_launch(void* graph_exec, void* stream);

/* The weight rounding of TTK_FP8W applied in place to a device f32 array: x <- fp8_e4m3(x / s) * s with s = the smallest power of
 * two >= max|x| / 448, returned in *scale_out (host).  This is exactly what ttk_*_create does to a TTK_FP8W matrix, exposed so that a
 * caller (and the parity tests) can build the equivalent TTK_BF16 model.                                                     */
int ttk_fp8_round_weights(float* x, int64_t n, float* scale_out, void* stream);

/* UnifiedVoice.forward(..., return_latent=True, clip_inputs=False) (unified_voice.py:544-599, get_logits :508-522):
 *   cond [B, D] f32, text [B, Tt] int64, codes [B, M] int64  ->  latents_out [B, M, D] f32 (= mel_logits[:, :-2]).  */
int ttk_ar_latents(ttk_ar* h, const float* cond, const int64_t* text, int Tt, const int64_t* codes, int M, int B,
				   float* latents_out, void* stream);

/* ------------------------------------------------------------------------------------------------------------------
 * Diffusion decoder: DiffusionTTS + the DDIM / ancestral step (tortoise_tts/models/diffusion.py:1389-1574, 325-431,
 * 646-694, 510-554) and AttentionBlock / GroupNorm32 / RelativePositionBias (arch_utils.py:24-190, xtransformers.py:148). */
typedef struct ttk_diff ttk_diff;

typedef struct {
	int model_channels, num_layers, in_channels, in_latent_channels, out_channels, num_heads;   /* diffusion.py:1392-1400 */
	int dtype;                                /* TTK_F32 | TTK_BF16 | TTK_F16 | TTK_FP8W | TTK_FP8 */
} ttk_diff_config;

/* Besides the hot-path subset of DiffusionTTS.state_dict() (weights.py: diffusion_shapes) the caller passes two derived
 * host tables: "__time_freqs" [C/2] (diffusion.py:1288-1290) and, per attention block, "<block>.__relbias" [H, 129] =
 * scale * emb[bucket(d)] for d = -64..64 (xtransformers.py:157-188) -- built by tortoise_tts_amd/diffusion.py.       */
int ttk_diff_create(ttk_diff** out, const ttk_diff_config* cfg, const ttk_weight_view* w, int n_w);
int ttk_diff_destroy(ttk_diff* h);

/* DiffusionTTS.timestep_independent(latents, cond, T, False) (diffusion.py:1487-1510):
 *   latents [b, M, Cl] f32, cond [b, 2C] f32, interp_idx [T] int32 device (nearest-neighbour source row of each output
 *   frame, F.interpolate) -> E_out [b, C, T] f32.                                                                    */
int ttk_diff_precompute(ttk_diff* h, const float* latents, const float* cond, const int32_t* interp_idx, int b, int M,
						int T, float* E_out, void* stream);

/* DiffusionTTS.forward(x, t, precomputed_aligned_embeddings=E[, conditioning_free=True]) (diffusion.py:1517-1574):
 *   x [b, in, T] f32, t [b] int64 device, E [b, C, T] f32 or NULL (=> conditioning_free) -> out [b, out_channels, T].  */
int ttk_diff_forward(ttk_diff* h, const float* x, const int64_t* t, const float* E, int b, int T, float* out, void* stream);

typedef struct {
	int64_t t;                 /* original-schedule timestep fed to the network (_WrappedModel map, diffusion.py:1232-1237) */
	float sqrt_recip_ac, sqrt_recipm1_ac, sqrt_ac_prev, sqrt_1m_ac_prev;   /* float64 tables -> f32 (:1264) */
	float coef1, coef2, min_log, max_log;                                   /* ancestral sampler only */
	float cfk;                 /* conditioning-free weight of this step (:391-393); < 0 disables the second evaluation */
	int sampler;               /* 0 = ddim (eta 0), 1 = p */
	int nonzero;               /* p sampler: add noise (i != 0) */
} ttk_step;

/* Stage E (as returned by ttk_diff_precompute, [b, C, T] f32) for a run of ttk_diff_step calls. */
int ttk_diff_begin(ttk_diff* h, const float* E, int b, int T, void* stream);
/* One sampler step: both network evaluations batched (cond + cond-free share every weight read), then the fused
 * epilogue; x [b, in, T] f32 updated in place; noise [b, in, T] f32 for the p sampler (NULL for ddim).
 * (GaussianDiffusion.p_mean_variance / ddim_sample / p_sample, diffusion.py:325-431, 646-694, 510-554)               */
int ttk_diff_step(ttk_diff* h, float* x, const ttk_step* st, const float* noise, void* stream);
/* Whole DDIM loop: steps[n-1], ..., steps[0] applied in that order (ddim_sample_loop_progressive :794-810).  The conditioning integrator of a
 * step does not depend on x, so the one of the next step runs on an internal side stream beside the current step's body; the side stream is
 * forked from and joined back to `stream` inside the call (results are ordered on `stream`; identical to the one-stream loop bit for bit). */
int ttk_diff_sample_ddim(ttk_diff* h, float* x, const float* E, int b, int T, const ttk_step* steps, int n_steps, void* stream);

/* The DDIM loop for several utterances of DIFFERENT length as one batch (no reference counterpart: the reference diffuses a text's lines one at
 * a time, inference.py:237-422; its network is batch-capable, diffusion.py:1517-1574, and its ramped conditioning-free guidance asserts b = 1
 * only because it reads t[0], :391-393 -- every element of this batch is at the same step).  Element e occupies a slot of Tp frames (Tp % 64 == 0)
 * of which tlen[e] (HOST array, 1..Tp) are real: x [b, in, Tp] f32 in place, E [b, C, Tp] f32; padding frames are ignored on input and
 * undefined on output.  Element e comes out bit for bit as ttk_diff_sample_ddim(b = 1, T = tlen[e]) gives it.  b <= 32.                      */
int ttk_diff_sample_ddim_lines(ttk_diff* h, float* x, const float* E, int b, int Tp, const int* tlen, const ttk_step* steps, int n_steps, void* stream);

/* Whole ancestral-sampler loop (p_sample_loop_progressive, diffusion.py:556-644 -- what the second caller of the path, train.py:178, runs with
 * 30 steps): the same two-stream loop with p_sample's update (:510-554).  noise [n_steps, b, in, T] f32: the `th.randn_like(x)` draws of the
 * steps in the order the loop makes them (block j belongs to the j-th executed step, steps[n-1-j]); the caller draws them so the generator
 * stream stays the reference's.  Equal to n_steps ttk_diff_step calls bit for bit.                                                       */
int ttk_diff_sample_p(ttk_diff* h, float* x, const float* E, int b, int T, const ttk_step* steps, int n_steps, const float* noise, void* stream);

/* ------------------------------------------------------------------ BigVGAN vocoder
 * The generator of models/bigvgan.py (BigVGAN.__init__ :419-486) on the hot path's kernels.  Weights: the generator's state_dict
 * with weight norm folded into plain `weight` tensors (tortoise_tts_amd/vocoder.py does that), plus "__aa_filter" [12], the Kaiser
 * low-pass every Activation1d builds (kaiser_sinc_filter1d(0.25, 0.3, 12), :40-69).                                        */
typedef struct ttk_voc ttk_voc;
typedef struct {
	int num_mels;                             /* 100 */
	int n_ups;                                /* <= 8 */
	int up_rate[8], up_kernel[8];             /* kernel % rate == 0, (kernel - rate) even */
	int ch0;                                  /* upsample_initial_channel; halves per stage, every stage a multiple of 8 */
	int n_kernels;                            /* AMP blocks per stage, 2..4 */
	int rb_kernel[4];                         /* odd, <= 11 */
	int rb_dil[4][3];
	int snake_logscale;
	int dtype;                                /* TTK_F32 | TTK_BF16 */
} ttk_voc_config;
int ttk_voc_create(ttk_voc** out, const ttk_voc_config* cfg, const ttk_weight_view* weights, int n_weights);
int ttk_voc_destroy(ttk_voc* h);
/* BigVGAN.inference(c) :522-534: mel [B, num_mels, T] f32 (denormalised log-mel) -> audio [B, 1, T * hop] f32 in [-1, 1]; the 10
 * padding frames of -11.5129 and the trimming of their 10 hops happen inside.                                             */
int ttk_voc_inference(ttk_voc* h, const float* mel, int B, int T, float* audio, void* stream);

/* ------------------------------------------------------------------ CLVP candidate scoring
 * models/clvp.py:21-136 (x-transformers branch): weights = CLVP.state_dict() with each attention's to_q / to_k / to_v stacked into
 * "<attn>.__qkv.weight" [3 * dim, dim] and "__rotary_inv_freq" [16] (RotaryEmbedding(32).inv_freq), as tortoise_tts_amd/clvp.py packs. */
typedef struct ttk_clvp ttk_clvp;
typedef struct {
	int dim;                                  /* 768; heads * 64 == dim */
	int heads;                                /* 12 */
	int depth;                                /* 20 attention + 20 feed-forward layers per encoder */
	int inner;                                /* dim * ff_mult = 1536 */
	int num_text_tokens, num_speech_tokens;   /* 256, 8192 */
	int dtype;                                /* TTK_F32 | TTK_BF16 */
} ttk_clvp_config;
int ttk_clvp_create(ttk_clvp** out, const ttk_clvp_config* cfg, const ttk_weight_view* weights, int n_weights);
int ttk_clvp_destroy(ttk_clvp* h);
/* CLVP.forward(text, speech_tokens, return_loss=False) :100-131: text [Bt, Tt] int64 with Bt == 1 (one line scored against every
 * candidate, what `text_tokens.repeat(B, 1)` at inference.py:394 amounts to) or Bt == B; codes [B, M] int64 -> scores [B] f32.  */
int ttk_clvp_score(ttk_clvp* h, const int64_t* text, int Bt, int Tt, const int64_t* codes, int B, int M, float* scores, void* stream);

/* ------------------------------------------------------------------ conditioning-latent encoders
 * One handle type for both: UnifiedVoice.conditioning_encoder (models/unified_voice.py:269-293: 1x1 conv 80 -> 1024, six AttentionBlocks of
 * 16 heads, output = position 0) and DiffusionTTS.contextual_embedder (models/diffusion.py:1441-1447: two k=3 stride-2 convs 100 -> 1024 ->
 * 2048, five AttentionBlocks of 16 heads x 128 with relative position bias; `get_conditioning` :1477-1485 takes the mean over positions).
 * Weights under module-relative names, as tortoise_tts_amd/conditioning.py packs them from the parents' state_dict():
 * "stem.{0,1}.weight" reshaped to [out, in * taps] and ".bias"; "blocks.{i}.norm|qkv|proj_out.weight|bias"; with relpos
 * "blocks.{i}.__relbias" [heads, 129] (the bias of clamp(k - q, -64, 64), already scaled by sqrt(head width), models/xtransformers.py:179-188). */
typedef struct ttk_cond ttk_cond;
enum { TTK_COND_STEM_CONV1 = 0, TTK_COND_STEM_DOWN4 = 1 };
typedef struct {
	int in_channels;      /* mel bands: 80 (AR) / 100 (diffusion) */
	int channels;         /* block width: 1024 / 2048; channels / num_heads must be 64 or 128 */
	int num_heads;
	int num_blocks;       /* 6 / 5 */
	int stem;             /* TTK_COND_STEM_CONV1: one 1x1 conv; TTK_COND_STEM_DOWN4: k=3 stride-2 padding-1 convs in -> channels/2 -> channels */
	int relpos;           /* blocks carry a relative position bias */
	int pool;             /* 0: position 0 (ConditioningEncoder mean=False); 1: mean over positions */
	int dtype;            /* TTK_F32 | TTK_BF16 */
} ttk_cond_config;
int ttk_cond_create(ttk_cond** out, const ttk_cond_config* cfg, const ttk_weight_view* weights, int n_weights);
int ttk_cond_destroy(ttk_cond* h);
/* one clip per batch row: mel f32 [b, in_channels, T] (the reference's channels-first layout) -> out f32 [b, channels] */
int ttk_cond_encode(ttk_cond* h, const float* mel, int b, int T, float* out, void* stream);

/* ------------------------------------------------------------------ the dense GEMM kernel on caller-provided operands
 * C f32 [M, N] = out_scale * A [M, K] . W [N, K]^T + bias, operands in `dtype`: f32, bf16, or TTK_FP8 = fp8-e4m3 bytes on the fp8 MFMA (out_scale
 * then carries the tensor scale; 0 = none).  N % 128 == 0, K % 32 / 64 / 128 == 0.  No reference counterpart: exposed so the kernel behind every
 * conv / linear of both networks can be tested against a plain matmul of the same operands. */
int ttk_gemm_nt(int dtype, const void* A, const void* W, int M, int N, int K, float out_scale, const float* bias, float* C, void* stream);

/* ------------------------------------------------------------------ mel front-ends of the conditioning path
 * TorchMelSpectrogram (models/arch_utils.py:361-395) and TacotronSTFT (:662-700 over STFT :560-623) as one handle type: reflect-padded
 * frames x "basis" [2 * (n_fft/2 + 1), n_fft] (windowed DFT, Re rows then Im rows) -> |.|^power -> x "mel_basis" [n_mels, n_fft/2 + 1] ->
 * log(max(., 1e-5)) [/ "mel_norms" [n_mels]].  tortoise_tts_amd/mel.py builds the matrices; arithmetic is f32 throughout. */
typedef struct ttk_mel ttk_mel;
typedef struct {
	int n_fft;            /* 1024; multiple of 64; window length == n_fft */
	int hop;              /* 256 */
	int n_mels;           /* 80 (AR side, 22.05 kHz) / 100 (diffusion side, 24 kHz) */
	int power;            /* 2: power spectrogram (TorchMelSpectrogram); 1: magnitude (TacotronSTFT) */
	int clip;             /* clamp samples to [-1, 1] first (TacotronSTFT.mel_spectrogram :694) */
	int has_norms;        /* divide band m of the log-mel by mel_norms[m] (:392-394) */
} ttk_mel_config;
int ttk_mel_create(ttk_mel** out, const ttk_mel_config* cfg, const ttk_weight_view* weights, int n_weights);
int ttk_mel_destroy(ttk_mel* h);
/* wav f32 [b, n] (n > n_fft / 2) -> mel f32 [b, n_mels, n / hop + 1] */
int ttk_mel_forward(ttk_mel* h, const float* wav, int b, int n, float* mel, void* stream);
/* torchaudio.functional.resample's polyphase FIR (emb/mel.py:67,86 resample clips to 22.05 kHz and 22.05 -> 24 kHz): rates reduced by their
 * gcd to gorig -> gnew; kernels f32 [gnew, 2 * width + gorig] on the device (tortoise_tts_amd/mel.py builds the windowed-sinc table);
 * wav f32 [b, n] -> out f32 [b, n_out], n_out = ceil(gnew * n / gorig).  Stateless. */
int ttk_resample_fir(const float* wav, int b, int n, const float* kernels, int gorig, int gnew, int width, float* out, int n_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif
