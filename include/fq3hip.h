/*
 * fq3hip.h -- C ABI of libfq3hip.so: the MI355X (gfx950) fast decode path for Qwen3-TTS.
 *
 * This is the drop-in boundary.  Every entry point is extern "C", takes plain device pointers,
 * sizes and a hipStream_t passed as void*; there are no torch types in any signature.  Python binds
 * it with ctypes (faster-qwen3-tts_amd/fq3hip/_lib.py); INTEGRATION.md shows the stub a reference
 * maintainer would add.  The precedent for a ctypes-bound native runtime in the reference is the
 * libqwen adapter, /root/reference/faster_qwen3_tts/ggml_backend.py:31-39.
 *
 * Each function cites the reference interface it replaces (paths relative to /root/reference).
 *
 * Conventions
 *   - return 0 on success, a negative FQ3_E* code on failure; fq3_last_error() gives the message.
 *     Nothing throws across the ABI.
 *   - all work is enqueued on the stream argument and is asynchronous to the host unless stated.
 *   - a context is bound to one device and is NOT re-entrant (static buffers), exactly like the
 *     reference's graph objects (examples/openai_server.py:71 serialises callers with a lock).
 *   - "T" below is the context dtype: bf16 (FQ3_BF16) or fp32 (FQ3_F32).  Weights, activations, KV
 *     cache, logits and sampling noise are all T; accumulation is fp32.
 *   - weight pointers are BORROWED: the caller keeps the allocations alive for the context lifetime.
 */
#ifndef FQ3HIP_H
#define FQ3HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FQ3_ABI_VERSION 5

enum { FQ3_BF16 = 0, FQ3_F32 = 1,
       FQ3_BF16X2 = 2 };   /* codec decoder only: activations kept as a bf16 high part + a bf16 residual (16 mantissa bits), products of the
                            * two halves with the bf16 weights on v_mfma_f32_16x16x32_bf16; see fq3_codec_config.dtype */
enum { FQ3_OK = 0, FQ3_EINVAL = -1, FQ3_EHIP = -2, FQ3_ESTATE = -3, FQ3_ETOOLONG = -4, FQ3_EUNSUPPORTED = -5, FQ3_ENOMEM = -6 };

typedef struct fq3_stack_dims {
    int32_t hidden, inter, n_layers, n_heads, n_kv_heads, head_dim, vocab;
    float rms_eps;
} fq3_stack_dims;

typedef struct fq3_config {
    int32_t dtype;              /* FQ3_BF16 | FQ3_F32 */
    fq3_stack_dims talker;      /* talker_graph.py:36-37 */
    fq3_stack_dims predictor;   /* predictor_graph.py:42-46 */
    int32_t num_code_groups;    /* 16: generate.py:42 */
    int32_t max_seq_len;        /* static talker KV length: model.py:113, talker_graph.py:43 */
    int32_t codec_eos_token_id; /* generate.py:41 */
    int32_t has_projection;     /* small_to_mtp_projection present: predictor_graph.py:54 */
    int32_t max_frames;         /* capacity of the on-device code buffer (>= max_new_tokens) */
} fq3_config;

/* One pre-norm GQA decoder layer.  qkv = rows [q | k | v] of the three projections concatenated,
 * gate_up = rows [gate | up].  All row-major [out, in], no biases (Appendix D of SURVEY.md). */
typedef struct fq3_layer_weights {
    const void* input_norm;  /* [H] */
    const void* qkv;         /* [q_dim + 2*kv_dim, H] */
    const void* q_norm;      /* [head_dim] */
    const void* k_norm;      /* [head_dim] */
    const void* o;           /* [H, q_dim] */
    const void* post_norm;   /* [H] */
    const void* gate_up;     /* [2*I, H] */
    const void* down;        /* [H, I] */
} fq3_layer_weights;

typedef struct fq3_weight_table {
    const fq3_layer_weights* talker_layers;     /* talker.n_layers entries (host array) */
    const void* talker_final_norm;              /* [H] */
    const void* codec_embedding;                /* [V, H]   talker.get_input_embeddings(), generate.py:100 */
    const void* codec_head;                     /* [V, H]   generate.py:101 */
    const fq3_layer_weights* predictor_layers;  /* predictor.n_layers entries */
    const void* predictor_final_norm;           /* [Hp] */
    const void* proj_w;                         /* [Hp, H] or NULL (identity)  predictor_graph.py:54 */
    const void* proj_b;                         /* [Hp] or NULL */
    const void* const* predictor_embeddings;    /* 15 x [Vp, H]   predictor_graph.py:57 */
    const void* const* lm_heads;                /* 15 x [Vp, Hp]  predictor_graph.py:56 */
    /* RoPE tables, fp32 values already rounded through T, [n_pos, head_dim/2] each.  Built by the
     * host with the same torch ops as the upstream rotary module so they are bit-identical to it. */
    const float* talker_cos;  const float* talker_sin;  int32_t talker_rope_len;
    const float* pred_cos;    const float* pred_sin;    int32_t pred_rope_len;
} fq3_weight_table;

/* Sampling policy (sampling.py:32-41 keyword arguments + generate.py:26-30). */
typedef struct fq3_sampling {
    float temperature;
    int32_t top_k;
    float top_p;
    int32_t do_sample;
    float repetition_penalty;
} fq3_sampling;

typedef struct fq3_ctx fq3_ctx;

const char* fq3_last_error(void);
int fq3_abi_version(void);

/* Replaces TalkerGraph.__init__ + PredictorGraph.__init__ (talker_graph.py:27-58,
 * predictor_graph.py:34-76): allocates the KV cache, I/O buffers and scratch on the current device.  The talker's KV cache is
 * PAGED: blocks of 64 keys (one attention tile) addressed through a per-context block table; this entry point gives the
 * context a private pool of ceil(max_seq_len / 64) blocks, all taken at creation -- the static cache of the reference
 * (talker_graph.py:43,153-170). */
int fq3_ctx_create(const fq3_config* cfg, fq3_ctx** out);
int fq3_ctx_destroy(fq3_ctx* ctx);

/* ---- paged KV: a block pool shared by the contexts of one scheduler (no reference equivalent: the reference reserves
 * max_seq_len slots per graph object, talker_graph.py:43) ------------------------------------------------------------------
 * A pool holds n_blocks blocks of 64 keys for every talker layer (K and V).  A pooled context owns nothing when idle; prefill
 * (fq3_prefill / fq3_prefill_batch / fq3_kv_import) takes the blocks of the prompt, fq3_decode_begin those of
 * prefill_len + max_new_tokens, fq3_talker_step that of its position; FQ3_ENOMEM (and nothing taken) when the pool is short.
 * fq3_kv_release hands blocks back; the caller makes sure that the context's queued work has finished (the same condition
 * under which a static cache may be overwritten).  Not thread-safe per context; the pool itself is. */
typedef struct fq3_kv_pool fq3_kv_pool;
int fq3_kv_pool_create(const fq3_config* cfg, int n_blocks, fq3_kv_pool** out);
int fq3_kv_pool_destroy(fq3_kv_pool* pool);            /* FQ3_ESTATE while contexts are attached */
/* blocks in total / free now / most ever in use at once; bytes of one block over all layers (K + V) */
int fq3_kv_pool_stats(const fq3_kv_pool* pool, int* n_blocks, int* n_free, int* high_water, int64_t* bytes_per_block);
int fq3_ctx_create_pooled(const fq3_config* cfg, fq3_kv_pool* pool, fq3_ctx** out);
/* take the blocks of key slots [0, n_positions) (capped at max_seq_len) now -- what a scheduler does BEFORE it prefills a request
 * into a spare context (prompt + max_new_tokens + 1), so that a short pool shows up before any work is queued; FQ3_ENOMEM and
 * nothing taken otherwise.  New table entries are written on `stream`. */
int fq3_kv_reserve(fq3_ctx* ctx, int n_positions, void* stream);
/* keep the blocks of key slots [0, keep_positions), return the others to the pool (0: all of them) */
int fq3_kv_release(fq3_ctx* ctx, int keep_positions);
/* blocks this context owns now */
int fq3_kv_blocks(const fq3_ctx* ctx);

/* Kernel-variant switches, all parity-tested both ways (no reference equivalent; the defaults are the measured-fastest):
 *   "weight_nt" 0|1|2 (non-temporal weight loads: none | talker | all), "pred_m2" 0|1 (predictor two-token prefill as one
 *   M = 2 pass), "pred_attn" 0|1 (one-wave predictor attention), "rows_per_wave_max" 1|2, "prefill_mode" 0|1,
 *   "flash_prefill" 0|1 (bf16 prefill attention as a flash-style matrix-core kernel), "skinny_gemm" 0|1 (prompts of <= 416 rows:
 *   weight-stationary GEMMs with the SwiGLU fused into the [gate | up] launch; 0 = the tiled / split-K kernels of longer prompts),
 *   "flash_small" 0|1 (round 5; prompts of <= 256 rows: every key tile of a query block resident in LDS, and the prompts of a packed
 *   fq3_prefill_batch over ONE pool share two attention launches per layer; 0 = the streamed-tile kernel, per prompt; bit-identical),
 *   "packed_weights" 0|1 (round 6; the weight-stationary GEMMs read the fragment-major copies of the layer matrices; bit-identical),
 *   "swiglu_tile" 0|1 (round 6; prefills of more than 416 rows run gate | up on the ring tile over a 16-row-interleaved copy of the
 *   weight with SwiGLU in the epilogue instead of GEMM + elementwise pass; bit-identical).
 * Resets a captured graph. */
int fq3_set_option(fq3_ctx* ctx, const char* key, int value);

/* Replaces the module references the graph objects keep (predictor_graph.py:52-58, talker_graph.py:40).
 * Round 6: for a bf16 context the call also takes a reference on a FRAGMENT-MAJOR copy of every layer matrix and head whose shape the
 * weight-stationary GEMM serves (K in {1024, 2048, 3072, 6144}, whole 16-row blocks): [row block][K / 32][64 lanes][8] -- the kilobyte one
 * wave's matrix-core operand load reads is contiguous (from the row-major matrix it is 16 rows x 64 B: 40 GB/s per CU against 125-135
 * out of the L2).  The first context of a weight replica builds the copies (+ one replica's layer matrices of HBM, gate | up in two blockings: 1.5 GB
 * at 0.6B, 3.9 GB at 1.7B), the others share them; fq3_ctx_destroy / a re-bind returns the references.  The table's matrices must not change
 * while bound (re-bind after an in-place update). */
int fq3_bind_weights(fq3_ctx* ctx, const fq3_weight_table* table);

/* ---- talker -------------------------------------------------------------------------------- */

/* TalkerGraph.prefill_kv (talker_graph.py:153-170): copy one layer of an externally computed prefill
 * cache, k and v laid out [n_kv_heads, L, head_dim] (the HF [1, kvh, L, d] tensor), into static slots
 * [0, L).  FQ3_ETOOLONG if L > max_seq_len (the reference raises RuntimeError, :163-167). */
int fq3_kv_import(fq3_ctx* ctx, int layer, const void* k, const void* v, int L, void* stream);
/* Test hook: copy static KV slots [0, L) of one layer back out in the same layout. */
int fq3_kv_export(fq3_ctx* ctx, int layer, void* k, void* v, int L, void* stream);
/* `dst` takes over the KV rows [0, L) of every talker layer from `src`.  Contexts of ONE pool: a block-table hand-over -- dst
 * returns its own blocks, receives src's block ids (one tiny launch rewrites its device table) and src is left empty; no KV row
 * moves.  Contexts of different pools (or private ones): the rows are copied block by block in one launch and src keeps its
 * blocks.  No reference equivalent: it lets a server prefill the next request into a spare context while the lock-step batch
 * keeps decoding, and hand the result to whichever lane frees up (fq3hip/batching.py).  FQ3_ETOOLONG if L > dst's max_seq_len. */
int fq3_kv_adopt(fq3_ctx* dst, fq3_ctx* src, int L, void* stream);

/* TalkerGraph.set_generation_state (talker_graph.py:172-196): left-pad count of the prompt mask and
 * the rope delta; replaces the 2048-row additive mask table with two integers. */
int fq3_set_generation_state(fq3_ctx* ctx, int n_pad, int rope_delta);

/* TalkerGraph.run (talker_graph.py:198-214): one token through all talker layers + final norm.
 * embeds T[H]; position = static cache slot; out_hidden T[H] (post-norm). */
int fq3_talker_step(fq3_ctx* ctx, const void* embeds, int position, void* out_hidden, void* stream);

/* Prefill (generate.py:107-122): L prompt embeddings T[L,H] through the talker, KV written to slots
 * [0,L); outputs last-position logits T[V] and post-norm hidden T[H].  n_pad = left padding. */
int fq3_prefill(fq3_ctx* ctx, const void* embeds, int L, int n_pad, void* out_logits, void* out_hidden,
                void* stream);
/* The same prefill for n prompts with ONE pass over the weights: row-wise work (norms, GEMMs, SwiGLU) on the packed rows of
 * all prompts, q/k norm + RoPE + KV write and causal attention per prompt into ctxs[q]'s own cache.  ctxs must share one
 * weight table; workspaces are ctxs[0]'s (sum of L <= its max_seq_len) -- otherwise this is n fq3_prefill calls.  No
 * reference equivalent (its prefill is one upstream forward per request, generate.py:107-118); used when a lock-step batch
 * admits several requests at once.  out_logits may be null, or hold nulls. */
int fq3_prefill_batch(fq3_ctx* const* ctxs, int n, const void* const* embeds, const int* L, const int* n_pad,
                      void* const* out_logits, void* const* out_hidden, void* stream);

/* Allocate the matrix-core prefill's activation workspace of this context now (max_seq_len rows; otherwise it is allocated by
 * the first fq3_prefill / fq3_prefill_batch that leads with this context).  A scheduler that prefills into spare contexts on a
 * side stream while lock-step lanes decode calls this when it is built, so that no device allocation happens under the decode. */
int fq3_prefill_reserve(fq3_ctx* ctx);

/* Test hook: 0 = matrix-core prefill (default), 1 = walk the prompt token by token through the decode kernels. */
int fq3_set_prefill_mode(fq3_ctx* ctx, int mode);

/* talker.codec_head (generate.py:182): logits T[V] = codec_head(hidden T[H]) (hidden is post-norm). */
int fq3_codec_head(fq3_ctx* ctx, const void* hidden, void* out_logits, void* stream);

/* ---- predictor ----------------------------------------------------------------------------- */

/* Construction-time predictor sampling policy (model.py:209-218, predictor_graph.py:47-50). */
int fq3_set_predictor_sampling(fq3_ctx* ctx, const fq3_sampling* s);

/* PredictorGraph.run (predictor_graph.py:204-214 = _full_loop :115-167): pred_input T[2,H_talker];
 * noise T[15, Vp] of Exp(1) variates or NULL when greedy; out_ids int64[15]; optional out_logits
 * T[15, Vp] (may be NULL) for parity tests. */
int fq3_predictor_loop(fq3_ctx* ctx, const void* pred_input, const void* noise, int64_t* out_ids,
                       void* out_logits, void* stream);

/* ---- sampler ------------------------------------------------------------------------------- */

/* sample_logits + apply_repetition_penalty (sampling.py:10-66): logits T[V] (not modified);
 * history int64[n_hist] of previous first-codebook ids (may be NULL); suppress range [sup_lo, sup_hi)
 * except keep_id (generate.py:46-50), plus eos when suppress_eos (generate.py:125,188);
 * noise T[V] Exp(1) (NULL when !do_sample); out_token int64[1]. */
int fq3_sample(fq3_ctx* ctx, const void* logits, int V, const fq3_sampling* s, const int64_t* history,
               int n_hist, int sup_lo, int sup_hi, int keep_id, int suppress_eos, const void* noise,
               int64_t* out_token, void* stream);

/* apply_repetition_penalty (sampling.py:10-29) alone, in place on logits T[V]: ids in history get x/p (x > 0) or x*p. */
int fq3_apply_repetition_penalty(fq3_ctx* ctx, void* logits, int V, const int64_t* history, int n_hist, float penalty,
                                 void* stream);

/* ---- fused on-device decode loop (generate.py:149-199 / streaming.py:106-154) -------------- */

typedef struct fq3_decode_params {
    fq3_sampling talker;         /* per-call sampling arguments */
    int32_t min_new_tokens;
    int32_t max_new_tokens;
    int32_t prefill_len;         /* talker_graph.prefill_kv return value */
    int32_t gen_step;            /* out.generation_step of the prefill (generate.py:122) */
    int32_t first_token;         /* token sampled from the prefill logits (generate.py:124-134) */
    const void* past_hidden;     /* T[H]: out.past_hidden (generate.py:121) */
    const void* trailing_text;   /* T[trailing_len, H] (generate.py:168-169) */
    int32_t trailing_len;
    const void* tts_pad_embed;   /* T[H] (generate.py:171) */
    const void* talker_noise;    /* T[noise_frames, V] or NULL */
    const void* pred_noise;      /* T[noise_frames, 15, Vp] or NULL */
    int32_t noise_frames;        /* rows in the noise rings; frame f reads row f % noise_frames */
} fq3_decode_params;

/* Arms the on-device loop state (token, position, history bitmap, counters).  Asynchronous: `p` is consumed before
 * the call returns, the buffers it points to must stay alive until the loop has finished. */
int fq3_decode_begin(fq3_ctx* ctx, const fq3_decode_params* p, void* stream);
/* Stops the loop of this context at the next frame boundary (the device state is marked done: further frames are no-ops, in a
 * lock-step batch the lane idles).  For a scheduler that abandons an utterance half way (a streaming consumer that went away):
 * without it the lane would keep decoding to its own EOS / max_new_tokens.  Asynchronous. */
int fq3_decode_cancel(fq3_ctx* ctx, void* stream);
/* Parity-test hook (teacher forcing; the shape of the reference's own relation tests, tests/test_e2e_parity.py:414-427,
 * applied decision by decision): after fq3_decode_begin, every sampler of the loop records ITS OWN id in
 * decisions[f][j] and continues with forced_codes[f][j] instead (both device int32[n_frames + 1][16]: [f][0] = the
 * first-codebook id of frame f, [f][1 + cb] = predictor codebook cb).  NULL, NULL switches it off. */
int fq3_decode_set_forced(fq3_ctx* ctx, const int32_t* forced_codes, int32_t* decisions, void* stream);
/* Enqueue n_frames iterations of the loop body; each is one hipGraph replay once
 * fq3_graph_capture() has run (talker_graph.py:109-147, predictor_graph.py:169-202), otherwise the
 * same kernels are launched directly.  Frames after EOS / limits are no-ops on device. */
int fq3_decode_frames(fq3_ctx* ctx, int n_frames, void* stream);
/* Synchronises the stream and reports: frames emitted so far, done flag; copies codes
 * int64[n, 16] for frames [from, n_frames_total) into out_codes (host or device memory visible to host). */
int fq3_decode_poll(fq3_ctx* ctx, int* n_frames_total, int* done, void* stream);
int fq3_decode_codes(fq3_ctx* ctx, int from, int count, int64_t* out_codes_dev, void* stream);
/* Capture the loop body into a hipGraph (no-op if already captured). */
int fq3_graph_capture(fq3_ctx* ctx, void* stream);
int fq3_graph_reset(fq3_ctx* ctx);

/* ---- prompt builder (model.py:583-805 _build_talker_inputs_local and upstream generate_icl_prompt) -------------------
 * Every row of the talker prompt is  text_projection(text_embedding(id))  +  a codec-stream row  (either may be absent):
 * the text rows come from one batched MLP on the matrix cores, the codec rows are embedding gathers / the speaker
 * embedding / the 16-way embedding sum of one reference frame, and ONE kernel assembles all rows from a small row program
 * the host derives from the token ids (which segment goes where is control flow, not arithmetic). */
typedef struct fq3_prompt_weights {
    const void* text_embedding;   /* [text_vocab, text_hidden]   talker.get_text_embeddings(), model.py:605 */
    const void* fc1_w;            /* [text_hidden, text_hidden]  talker.text_projection.linear_fc1 */
    const void* fc1_b;            /* [text_hidden] */
    const void* fc2_w;            /* [H, text_hidden]            talker.text_projection.linear_fc2 */
    const void* fc2_b;            /* [H] */
    int32_t text_vocab, text_hidden;
} fq3_prompt_weights;
int fq3_bind_prompt_weights(fq3_ctx* ctx, const fq3_prompt_weights* w);
/* out T[n, H] = text_projection(text_embedding(ids int64[n]))  (fc1 -> SiLU -> fc2, one rounding per module output). */
int fq3_text_project(fq3_ctx* ctx, const int64_t* ids, int n, void* out, void* stream);
/* Assemble n_rows prompt rows.  prog int32[n_rows][3] = {text_row, kind, arg}: text_row indexes text_rows T[n_text, H]
 * (-1: no text part); kind 0 no codec part, 1 codec_embedding[arg], 2 the speaker embedding spk_embed T[H],
 * 3 the sum over the 16 codebook embeddings of reference frame arg (ref_codes int64[n_ref, 16], model.py:699-712).
 * out[r] = text part + codec part (one rounding), or the single part that is present. */
int fq3_prompt_rows(fq3_ctx* ctx, const void* text_rows, int n_text, const int32_t* prog, int n_rows,
                    const int64_t* ref_codes, int n_ref, const void* spk_embed, void* out, void* stream);

/* ---- batched decode: B utterances in lock-step over one weight stream ------------------------------
 * No reference equivalent (the reference fixes batch = 1: talker_graph.py:46, predictor_graph.py:70; SURVEY.md
 * section 8f rank 3).  A batch borrows n_lanes (1..128) ordinary contexts that share ONE weight table, ONE config and
 * ONE max_seq_len.  Each lane is prepared with the single-stream entry points (fq3_prefill, fq3_set_generation_state,
 * fq3_decode_begin) and read back with fq3_decode_poll / fq3_decode_codes on ITS context; fq3_batch_frames replaces
 * fq3_decode_frames for all lanes at once.  Lanes that are done (EOS, limits, or never begun) idle on device and can be
 * re-armed with fq3_decode_begin between calls (continuous batching).  Every lane samples with its own policy (the one
 * its fq3_decode_begin / fq3_set_predictor_sampling set): top_p < 1 (sampling.py:57-65) is honoured per lane inside the
 * same launch.  With the VALU GEMVs ("mfma" 0, the fp32 default) a lane's ids are bit-identical to the same utterance
 * decoded alone with the same noise; the bf16 default multiplies one 16-column token tile per v_mfma_f32_16x16x32_bf16,
 * so 9..16 lanes read the weights once and issue the MFMAs of 1..8; lanes 17..32, 33..48, ... 113..128 are further token tiles
 * of the same launch over the register-resident weight fragments. */
typedef struct fq3_batch fq3_batch;
int fq3_batch_create(fq3_ctx* const* lanes, int n_lanes, fq3_batch** out);
int fq3_batch_destroy(fq3_batch* b);
int fq3_batch_size(const fq3_batch* b);
/* Enqueue n_frames lock-step iterations (hipGraph replays once fq3_batch_graph_capture() has run). */
int fq3_batch_frames(fq3_batch* b, int n_frames, void* stream);
int fq3_batch_graph_capture(fq3_batch* b, void* stream);
int fq3_batch_graph_reset(fq3_batch* b);
/* Every lane's fq3_decode_poll in one go, split in two so that a scheduler can keep the GPU fed: fq3_batch_poll_async enqueues ONE
 * gather launch + ONE small copy on `stream` (in stream order: it sees the frames queued before it) into slot 0..3;
 * fq3_batch_poll_wait blocks until that slot's copy has landed and fills n_frames_total[fq3_batch_size] / done[fq3_batch_size]
 * (either may be NULL).  Frames queued AFTER the poll run while the host waits for and digests it.  FQ3_ESTATE: nothing was queued
 * in the slot. */
int fq3_batch_poll_async(fq3_batch* b, int slot, void* stream);
int fq3_batch_poll_wait(fq3_batch* b, int slot, int* n_frames_total, int* done);
/* "mfma" 0|1: batch GEMVs on the matrix cores (bf16 contexts; default 1: ids checked against the oracle by teacher
 * forcing) or on the VALU kernels (0: every lane bit-identical to the same utterance decoded alone).
 * "skinny" 0|1 (with "mfma" 1, more than 16 lanes): o_proj / down through the weight-stationary kernel of the short-prompt prefill
 * (default 1) or through the one-row-block-per-workgroup batch GEMV (0); 2 takes the weight-stationary kernel at every lane count
 * (measurement switch).
 * "norm_skinny" 0|1 (more than 32 lanes): the normalising GEMVs (qkv, gate | up, heads) as ONE normalisation launch + the
 * weight-stationary GEMM kernel (default 1) or as the per-workgroup-panel kernels of the lower lane counts (0);
 * "norm_skinny_above" n moves that lane count (measurement switch).
 * Round 5: "attn_lane" 0|1|2 -- the talker attention as ONE workgroup per (kv head, lane) that writes the final head outputs (no
 * partial slots, no merge launch): never | from "attn_lane_from" lanes on (default 64; bf16 matrix-core path) | always; "attn_lane_keys"
 * 8|16 (keys per load step); "pred_pair" 0|1 -- the predictor's two-token prefill as one pass over 2 B token rows where that is
 * bit-identical (above 32 lanes; default 1); "norm_fused" 0|1 -- the RMSNorm of qkv / gate | up / lm heads inside the
 * weight-stationary GEMM, from sum-of-squares partials the residual GEMM's epilogue leaves (measured SLOWER on MI355X: default 0).
 * Round 6: "pred_attn_group" 0|1 -- the code predictor's attention as one wave per (kv group, lane) that serves the group's q heads from
 * one read of the LIVE K / V rows (default 1; bit-identical) or as one wave per (q head, lane) over all 16 slots (0).
 * "packed_weights" 0|1 -- the weight-stationary GEMMs read the fragment-major copies fq3_bind_weights keeps of the bf16 layer matrices
 * (default 1; bit-identical) or the row-major matrices (0).
 * "groups" 0..4: LANE GROUPS (a measurement switch).  The lanes split into that many independent lock-step chains of whole 16-lane
 * tiles, each with its own frame graph, advanced concurrently on streams the library probes for a hardware queue of their own (they
 * fork from / join `stream` inside fq3_batch_frames, so the caller sees one stream as before); 0 (default) = automatic = ONE chain:
 * measured on MI355X, two concurrent chains of 32 lanes take 7.15 ms per frame where one chain of 64 takes 5.57 ms.  A lane's values
 * do not depend on the grouping.  When no stream with its own queue is found, or `stream` is being captured, the chains run one
 * after another on `stream`. */
int fq3_batch_set_option(fq3_batch* b, const char* key, int value);
/* The caller's own side streams for lane groups 1..n (n <= 3), e.g. streams it has probed against every other stream it keeps
 * busy (a vocoder stream, a prefill stream); they are borrowed, not owned.  n = 0 returns to the library's own probing. */
int fq3_batch_set_group_streams(fq3_batch* b, void* const* streams, int n);

/* ---- 12 Hz codec decoder (speech_tokenizer.decode, model.py:924) ---------------------------- */
typedef struct fq3_codec fq3_codec;
typedef struct fq3_codec_config {
    int32_t dtype;              /* FQ3_BF16: the checkpoint dtype, what the reference's Torch path runs (PCM ~8e-3 RMS from an fp32 evaluation on a
                                 * trained-vocoder-like network).  FQ3_F32: fp32 weights (a bf16 checkpoint widened exactly), activations and
                                 * v_mfma_f32_16x16x4_f32 products: ~1e-6 RMS, ~4x the time.  FQ3_BF16X2: the high-precision mode for serving --
                                 * weights stay bf16 (bound with every K column DUPLICATED: [N][taps][2 Cin]), every activation and
                                 * per-channel vector is a 32-bit word (bf16 hi | bf16 lo << 16; bind those as such), a GEMM reads the
                                 * activation tensor as a [rows][2 Cin] bf16 matrix and issues two bf16 MFMAs per K pair: <= 1e-3 PCM RMS
                                 * (north star) at about twice the bf16 time */
    int32_t codebook_size, codebook_dim, rvq_dim, num_quantizers, num_semantic;
    int32_t latent_dim, hidden, inter, n_layers, n_heads, head_dim, sliding_window;
    float rms_eps;
    int32_t n_upsample;  int32_t upsampling_ratios[4];
    int32_t n_rates;     int32_t upsample_rates[8];
    int32_t decoder_dim;
    int32_t max_frames;
} fq3_codec_config;
int fq3_codec_create(const fq3_codec_config* cfg, fq3_codec** out);
int fq3_codec_destroy(fq3_codec* c);
/* name -> device pointer binding; names are the checkpoint tensor names under "decoder." */
int fq3_codec_bind(fq3_codec* c, const char* name, const void* ptr, int64_t numel);
/* Kernel-variant switch of the codec decoder (parity tests / measurements): "fuse_units" 1 = the residual units of the
 * 96-channel decoder block (SnakeBeta -> k7 conv -> SnakeBeta -> 1x1 conv -> + skip) run as ONE launch each with the middle
 * tensor kept in LDS (the default: 7.45 vs 7.92 ms per 370-frame decode), 2 = the 192-channel block's too (measured slower:
 * 7.93 ms), 0 = two GEMM launches per unit.  All settings give bit-identical waveforms.  FQ3_BF16X2 (round 5): the 96-channel block's
 * units fuse too (the middle tensor parked in LDS as hi | lo words); measured a wash (14.46 -> 14.31 ms per 370-frame decode).
 * "glds_cap8" n (round 6, measurement hook): FQ3_BF16X2 GEMMs take the eight-wave LDS-DMA tiles up to n tiles of 128 x 64 (0 = the
 * launcher's default: whatever the grid); bit-identical.
 * The four activation workspaces follow the CALL: a batched decode needs B x elems(T) elements and grows them when short
 * (FQ3_ENOMEM when that fails: the Python host then decodes utterance by utterance). */
int fq3_codec_set_option(fq3_codec* codec, const char* key, int value);
int fq3_codec_finalize(fq3_codec* c, void* stream);
/* number of PCM samples produced for T frames (the causal transposed convs trim, so < 1920*T) */
int64_t fq3_codec_num_samples(const fq3_codec* c, int T);
/* codes int64[T,16] (device) -> pcm float32[num_samples] (device), clamped to [-1,1]. */
int fq3_codec_decode(fq3_codec* c, const int64_t* codes, int T, float* pcm, void* stream);
/* The same decode, but only PCM samples [first_sample, num_samples(T)) are produced, into pcm[0 ..): what the streaming
 * call sites keep of each re-decode (model.py:1095-1100 `audio[cut:][prev_len:]`, :1128-1133 `audio[ctx_samples:]`).
 * The decoder is causal with a bounded receptive field after its transformer, so only the rows those samples depend on are
 * recomputed; the values are bit-identical to the corresponding tail of fq3_codec_decode's output. */
int fq3_codec_decode_tail(fq3_codec* c, const int64_t* codes, int T, int64_t first_sample, float* pcm, void* stream);
/* The batched form of the vocoder interface (speech_tokenizer.decode takes audio_codes [B, T, 16], model.py:924; payload shape pinned
 * by the reference's tests/test_sample_rate.py:53-75): codes int64[B][T][16] (device), B utterances of T frames each, decoded by ONE
 * set of launches -- every tensor is [utterance][rows][C], every launch carries the utterance in a grid dimension, so the
 * latency-bound frame-level transformer and the skinny convs of short inputs fill the chip B times better.  pcm
 * float32[B][num_samples(T) - first_sample]: per utterance exactly (bit for bit) what fq3_codec_decode_tail(codes[b], T,
 * first_sample) produces.  Utterances of different lengths: pad the shorter ones with any valid ids -- the decoder is causal, a
 * prefix of the codes gives the same prefix of the waveform -- and keep num_samples(T_b) samples of each.  The workspace grows to
 * B utterances on first use (a device synchronisation; not inside a graph capture). */
int fq3_codec_decode_batch(fq3_codec* c, const int64_t* codes, int B, int T, int64_t first_sample, float* pcm, void* stream);
/* Reference-prefix state (round 6).  The ICL call sites decode `ref_codes + generated codes` and cut the reference part off
 * (model.py:919-937; every phase-1 chunk of the streaming path again, model.py:1085-1115): the frame-level front end -- RVQ sums,
 * pre_conv, the sliding-window transformer -- re-runs over the same reference frames for every chunk and every utterance of a voice.
 * fq3_codec_prefix_create runs it ONCE over ref_codes int64[ref_len][16] (device) and keeps what later rows read from the prefix: the
 * pre_conv's two context rows, every layer's post-RoPE K / V of the last sliding_window - 1 rows, the last 48 output rows (the conv
 * stack's halo).  fq3_codec_decode_batch_prefix is fq3_codec_decode_batch for codes whose first fq3_codec_prefix_frames(prefix) frames
 * of utterance b ARE that prefix's codes (prefixes: HOST array of B states of one length, B <= 128; the same state may repeat): the
 * front end computes rows [ref_len, T) only.  The waveform is bit-identical to the full decode's (the front end is causal and a row's
 * arithmetic does not depend on the row count); a first_sample that reaches further back than the cached rows silently takes the full
 * path.  A state belongs to the codec that made it and must be destroyed before it. */
typedef struct fq3_codec_prefix fq3_codec_prefix;
int fq3_codec_prefix_create(fq3_codec* c, const int64_t* ref_codes, int ref_len, fq3_codec_prefix** out, void* stream);
int fq3_codec_prefix_destroy(fq3_codec_prefix* prefix);
int fq3_codec_prefix_frames(const fq3_codec_prefix* prefix);
int fq3_codec_decode_batch_prefix(fq3_codec* c, const fq3_codec_prefix* const* prefixes, const int64_t* codes, int B, int T,
                                  int64_t first_sample, float* pcm, void* stream);

/* ---- reference-audio analysis (create_voice_clone_prompt, model.py:415-463 -> upstream qwen_tts) -----------------------
 * What the reference runs once per new (ref_audio, ref_text) pair and then caches (model.py:424-463):
 *   speech_tokenizer.encode(ref_wav)  -> ref_code  int64 [T, 16]   (the 12.5 Hz tokenizer's ENCODER: SEANet conv stack,
 *       8-layer causal transformer, stride-2 conv, split residual VQ -- the Mimi architecture, transformers
 *       models/mimi/modeling_mimi.py:450-492, :782-928, :1030-1138, :1231-1268)
 *   extract_speaker_embedding(wav)    -> ref_spk_embedding [enc_dim] (log-mel front end + ECAPA-TDNN, transformers
 *       models/qwen2_5_omni/modeling_qwen2_5_omni.py:2412-2707 is the readable sibling)
 * fp32 only; either half may be left unbound.  Weight names are the checkpoint names under "encoder." / "speaker_encoder."
 * in the layouts fq3hip/refenc.py packs (documented in csrc/fq3_refenc.hip). */
typedef struct fq3_refenc fq3_refenc;
typedef struct fq3_refenc_config {
    /* speech-tokenizer encoder */
    int32_t num_filters;                     /* 64 */
    int32_t n_ratios; int32_t ratios[8];     /* strides in encoder order (4, 5, 6, 8) */
    int32_t kernel_size, last_kernel_size, residual_kernel_size, n_residual_layers, dilation_growth_rate, compress;   /* 7 3 3 1 2 2 */
    int32_t hidden, n_layers, n_heads, head_dim, inter, sliding_window;     /* 512 8 8 64 2048 250 */
    float   norm_eps;                        /* 1e-5 */
    int32_t num_quantizers, num_semantic, codebook_size, codebook_dim;      /* 16 1 2048 256 */
    int32_t max_positions;                   /* rows of the "encoder.rope.cos/sin" tables (25 Hz frames) */
    /* speaker encoder */
    int32_t mel_dim, n_fft, hop, n_bins_padded;   /* 128 1024 256 544 (n_fft/2 + 1 rounded up to a multiple of 32) */
    int32_t n_enc; int32_t enc_channels[8], enc_kernel_sizes[8], enc_dilations[8];   /* 5: 512 512 512 512 1536 / 5 3 3 3 1 / 1 2 3 4 1 */
    int32_t attn_channels, res2net_scale, se_channels, enc_dim;             /* 128 8 128 1024|2048 */
} fq3_refenc_config;
int fq3_refenc_create(const fq3_refenc_config* cfg, fq3_refenc** out);
int fq3_refenc_destroy(fq3_refenc* r);
int fq3_refenc_bind(fq3_refenc* r, const char* name, const void* ptr, int64_t numel);
int fq3_refenc_finalize(fq3_refenc* r, void* stream);
/* number of 12.5 Hz frames the encoder produces for n_samples of 24 kHz audio (MimiModel.get_encoded_length, :1270-1284) */
int64_t fq3_refenc_num_frames(const fq3_refenc* r, int64_t n_samples);
/* pcm float32[n] (device, 24 kHz mono) -> codes int64[num_frames][num_quantizers] (device) */
int fq3_refenc_encode(fq3_refenc* r, const float* pcm, int64_t n, int64_t* codes, void* stream);
/* pcm float32[n] (device, 24 kHz mono) -> embed float32[enc_dim] (device); mel (optional, may be null) receives the
 * log-mel frames float32[n / hop][mel_dim] the encoder consumed */
int fq3_refenc_speaker(fq3_refenc* r, const float* pcm, int64_t n, float* embed, float* mel, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* FQ3HIP_H */
