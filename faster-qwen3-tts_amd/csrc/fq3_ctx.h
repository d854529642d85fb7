// Internal: context layout shared by the decode TU (fq3_api.hip) and the prefill TU (fq3_prefill.hip).
#pragma once
#include "../../include/fq3hip.h"
#include "decode_kernels.cuh"
#include <hip/hip_runtime.h>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

// Talker KV cache: PAGED.  A pool holds, per layer, n_blocks blocks of kKeysPerTile = 64 keys ([n_kv][64][128] elements of K, the
// same of V); a context owns a list of block ids -- its block table, mirrored on the device for the kernels -- and grows / returns
// it through kv_ensure_ / fq3_kv_release.  fq3_ctx_create gives a context a private pool of ceil(max_seq_len / 64) blocks, all of
// them taken at creation (the static cache of the reference, talker_graph.py:43); fq3_ctx_create_pooled draws from a pool shared by
// the contexts of one scheduler, so that an idle context holds nothing, a prefilled one its prompt's blocks, and a lane takes a
// staged prompt over by exchanging block ids (fq3_kv_adopt) instead of copying rows.
struct fq3_kv_pool {
    int dtype = 0, esz = 2, n_layers = 0, n_kv = 0, n_blocks = 0;
    size_t blk_elems = 0;             // elements per block per layer and per K | V: n_kv * 64 * 128
    std::vector<void*> k, v;          // per layer [n_blocks][n_kv][64][128]
    std::vector<int> free_list;       // LIFO of free block ids
    std::mutex mu;
    int in_use_high = 0;              // high-water mark of blocks handed out
    int users = 0;                    // contexts attached
    bool is_private = false;          // owned by exactly one context (destroyed with it)
};

struct StackBufs {
    std::vector<void*> k, v;          // per layer: predictor [n_kv][max_seq][128] (contiguous); talker: the pool's layer arrays
    int max_seq = 0, workers = 1;
    // talker only
    fq3_kv_pool* pool = nullptr;
    int* d_table = nullptr;           // device int32[max_blocks]: block id of key tile i (unowned entries: 0, never attended)
    std::vector<int> blocks;          // host mirror: the block ids this context owns, in tile order
    int max_blocks = 0;
};

struct fq3_ctx {
    fq3_config cfg{};
    int esz = 2;
    std::vector<fq3_layer_weights> tl, pl;
    fq3_weight_table wt{};
    std::vector<const void*> pemb, lmh;
    const void** d_pemb = nullptr;    // device array of the 15 predictor embedding tables
    bool bound = false;
    StackBufs tk, pk;
    // scratch (device)
    void *h = nullptr, *xin = nullptr, *qkv = nullptr, *act = nullptr, *logits = nullptr, *past_hidden = nullptr;
    void *pred_in = nullptr, *pred_x = nullptr, *pred_next = nullptr, *plogits = nullptr, *tmp_hidden = nullptr;
    void *attn_out = nullptr, *attn_out2 = nullptr;   // final attention output T[q_dim] (code predictor's one-wave-per-head kernel)
    void *h2 = nullptr, *qkv2 = nullptr, *act2 = nullptr, *pred_x2 = nullptr;      // second token of the M = 2 predictor prefill
    size_t part_stride = 0;
    float* part = nullptr;
    float* rope_now = nullptr;
    unsigned char* seen_api = nullptr;
    fq3::DecodeState* st = nullptr;
    fq3::TeacherForcing* tf = nullptr;     // allocated by the first fq3_decode_set_forced (parity tests); null for a product context
    unsigned char* seen = nullptr;
    int* codes = nullptr;
    int64_t* ids64 = nullptr;
    int n_pad = 0, rope_delta = 0;
    int opt_nt = 1;               // weight-load cache policy (see run_stack)
    int opt_m2 = 1;               // code predictor: the two-token prefill is ONE M = 2 pass over the weights (FQ3_M2=0: two
                                  // single-token passes); parity-tested both ways; 2.06 vs 2.11 ms/frame with the v4 kernels
    int opt_pred_attn = 1;        // code predictor: one-wave-per-head register-only attention writing the final head output (FQ3_PRED_ATTN=0: generic split-KV kernel + merge)
    int opt_rmax = 2;             // GEMV rows-per-wave cap
    int opt_flash_prefill = 1;    // bf16 prefill attention on the matrix cores (0: the per-row wave kernel)
    int opt_flash_small = 1;      // prompts of <= 256 rows: every key tile resident in LDS, the sequences of a packed prefill in one launch (0: the streamed-tile kernel, per prompt)
    int opt_swiglu_tile = 1;      // many-row prefills: gate | up on the ring tile over the 16-row-interleaved copy with SwiGLU in the epilogue (round 6; bit-identical); 0: GEMM + silu_mul
    int opt_packed = 1;           // weight-stationary GEMMs read the fragment-major copies of the layer matrices (round 6; bit-identical); 0: the row-major matrices
    std::vector<std::pair<const void*, int>> packed_refs;      // (matrix, kind) references this context holds in the registry
    int opt_no_skinny = 0;        // 1: short-prompt prefill GEMMs on the tiled / split-K kernels instead of the weight-stationary one (measurement)
    int prefill_mode = 0;         // 0 auto (MFMA), 1 token walk
    bool talker_wave = true;      // talker sampler variant baked into the captured graph
    fq3_sampling pred_sampling{0.9f, 50, 1.0f, 1, 1.0f};
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr;
    std::vector<void*> allocs;
    // prompt builder (fq3_prompt.hip): text embedding table + text_projection MLP, lazily grown workspaces
    fq3_prompt_weights pw{};
    bool pw_bound = false;
    void *pw_x = nullptr, *pw_h = nullptr;
    int pw_cap = 0;
    // MFMA prefill workspace (lazily allocated, sized for max_seq_len rows)
    void *pf_x = nullptr, *pf_xn = nullptr, *pf_qkv = nullptr, *pf_att = nullptr, *pf_gu = nullptr, *pf_act = nullptr, *pf_ws = nullptr;
};


// Fragment-major copies of the bf16 layer matrices (skinny_gemm.cuh, SkinnyArgs::Wp): a process-wide, reference-counted registry keyed by
// the row-major matrix's address.  fq3_bind_weights takes a reference per matrix for the binding context (the first one packs: a few
// hundred microseconds per matrix, + N K 2 bytes of HBM; every context of the same replica shares the copy), fq3_ctx_destroy / a
// re-bind returns them; the last reference frees the copy.  kind 0: 16-row blocks; 1: [gate | up] pairs (8 + 8 rows per block).
// fq3_packed_find_ returns the copy for the weight-stationary GEMM's launchers, or null (not packed: the row-major path).
const void* fq3_packed_find_(const void* W, int kind);
int fq3_packed_acquire_(const void* W, int N, int K, int kind);
void fq3_packed_release_(const void* W, int kind);

int fq3_fail_(int code, const std::string& m);                 // sets the thread-local error string
// make sure the context owns the blocks of key slots [0, n_pos) (capped at max_seq_len); new table entries are written on `s`.
// FQ3_ENOMEM when the pool has too few free blocks (nothing is taken then).
int fq3_kv_ensure_(fq3_ctx* c, int n_pos, hipStream_t s);
int fq3_prefill_reserve_(fq3_ctx* c);
int fq3_prefill_mfma_(fq3_ctx* c, const void* embeds, int L, int n_pad, void* out_logits, void* out_hidden, hipStream_t s);
int fq3_prefill_batch_mfma_(fq3_ctx* const* cs, int n, const void* const* embeds, const int* L, const int* n_pad, void* const* out_logits,
                            void* const* out_hidden, hipStream_t s);
int fq3_codec_head_launch_(fq3_ctx* c, const void* hidden, void* out_logits, hipStream_t s);
int fq3_dmalloc_(fq3_ctx* c, void** p, size_t bytes);
