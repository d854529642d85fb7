// libfq3hip.so: context, weight binding, launch orchestration and hipGraph capture for the decode path.
// C ABI declared in include/fq3hip.h (which cites the reference interface each entry replaces).
#define FQ3_SKINNY_EXTERN           // skinny_gemm.cuh: the kernels (and skinny_pack) are instantiated in fq3_prefill.hip only
#include "fq3_ctx.h"
#include "sampler.cuh"
#include "sampler_wave.cuh"
#include "skinny_gemm.cuh"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

using namespace fq3;

static thread_local std::string g_err;
static int fail(int code, const std::string& m) { g_err = m; return code; }
int fq3_fail_(int code, const std::string& m) { return fail(code, m); }
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(FQ3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

extern "C" const char* fq3_last_error(void) { return g_err.c_str(); }
extern "C" void fq3_set_error_(const char* msg) { g_err = msg ? msg : ""; }   // used by the codec TU
extern "C" int fq3_abi_version(void) { return FQ3_ABI_VERSION; }

static int dmalloc(fq3_ctx* c, void** p, size_t bytes, bool zero = true) {
    HIPCHK(hipMalloc(p, bytes));
    // (a legacy-stream memset is illegal while ANOTHER host thread captures a graph on a blocking stream: workspaces that
    //  are allocated lazily -- possibly next to another context's capture -- are therefore never zeroed; they are fully
    //  written before they are read)
    if (zero) HIPCHK(hipMemset(*p, 0, bytes));
    c->allocs.push_back(*p);
    return 0;
}
int fq3_dmalloc_(fq3_ctx* c, void** p, size_t bytes) { return dmalloc(c, p, bytes, false); }

static bool dims_ok(const fq3_stack_dims& d) {
    return d.head_dim == kHeadDim && d.hidden % 8 == 0 && d.inter % 8 == 0 && d.n_heads % d.n_kv_heads == 0 &&
           (d.n_heads / d.n_kv_heads == 1 || d.n_heads / d.n_kv_heads == 2 || d.n_heads / d.n_kv_heads == 4) &&
           d.vocab <= kMaxVocab && d.vocab % 8 == 0 && d.hidden <= 2048 && d.inter <= 6144 && d.n_heads * kHeadDim <= 6144 && d.n_layers >= 1;
}

// -------------------------------------------------------------------------------------------------
// Paged KV: block pool, block tables
// -------------------------------------------------------------------------------------------------
static int cfg_check(const fq3_config* cfg) {
    if (cfg->dtype != FQ3_BF16 && cfg->dtype != FQ3_F32) return fail(FQ3_EINVAL, "dtype must be FQ3_BF16 or FQ3_F32");
    if (!dims_ok(cfg->talker) || !dims_ok(cfg->predictor))
        return fail(FQ3_EUNSUPPORTED, "unsupported dims (need head_dim 128, GQA ratio 1/2/4, vocab <= 4096 and a multiple of 8, "
                                      "hidden <= 2048 and intermediate <= 6144, both multiples of 8)");
    if (cfg->num_code_groups < 2 || cfg->num_code_groups > 64) return fail(FQ3_EINVAL, "num_code_groups");
    if (cfg->max_seq_len < 8) return fail(FQ3_EINVAL, "max_seq_len");
    return 0;
}

static void pool_free_(fq3_kv_pool* p) {
    for (void* q : p->k) if (q) (void)hipFree(q);
    for (void* q : p->v) if (q) (void)hipFree(q);
    delete p;
}

static int pool_create_(const fq3_config* cfg, int n_blocks, fq3_kv_pool** out) {
    if (n_blocks < 1) return fail(FQ3_EINVAL, "fq3_kv_pool_create: n_blocks must be positive");
    fq3_kv_pool* p = new fq3_kv_pool();
    p->dtype = cfg->dtype; p->esz = cfg->dtype == FQ3_BF16 ? 2 : 4;
    p->n_layers = cfg->talker.n_layers; p->n_kv = cfg->talker.n_kv_heads; p->n_blocks = n_blocks;
    p->blk_elems = (size_t)p->n_kv * kKeysPerTile * kHeadDim;
    p->k.assign(p->n_layers, nullptr); p->v.assign(p->n_layers, nullptr);
    const size_t bytes = (size_t)n_blocks * p->blk_elems * p->esz;
    for (int i = 0; i < p->n_layers; ++i) {
        // zero-filled: a table entry of a tile the context does not own points at block 0, whose rows are loaded (and masked) only
        if (hipMalloc(&p->k[i], bytes) != hipSuccess || hipMalloc(&p->v[i], bytes) != hipSuccess ||
            hipMemset(p->k[i], 0, bytes) != hipSuccess || hipMemset(p->v[i], 0, bytes) != hipSuccess) {
            pool_free_(p);
            return fail(FQ3_EHIP, "fq3_kv_pool_create: allocation of " + std::to_string(2 * bytes >> 20) + " MiB per layer failed");
        }
    }
    p->free_list.resize(n_blocks);
    for (int i = 0; i < n_blocks; ++i) p->free_list[i] = n_blocks - 1 - i;      // block 0 is handed out first
    *out = p;
    return FQ3_OK;
}

extern "C" int fq3_kv_pool_create(const fq3_config* cfg, int n_blocks, fq3_kv_pool** out) {
    if (!cfg || !out) return fail(FQ3_EINVAL, "null argument");
    if (int r = cfg_check(cfg)) return r;
    return pool_create_(cfg, n_blocks, out);
}

extern "C" int fq3_kv_pool_destroy(fq3_kv_pool* p) {
    if (!p) return FQ3_OK;
    if (p->users > 0) return fail(FQ3_ESTATE, "fq3_kv_pool_destroy: contexts are still attached");
    (void)hipDeviceSynchronize();
    pool_free_(p);
    return FQ3_OK;
}

extern "C" int fq3_kv_pool_stats(const fq3_kv_pool* p, int* n_blocks, int* n_free, int* high_water, int64_t* bytes_per_block) {
    if (!p) return fail(FQ3_EINVAL, "null pool");
    fq3_kv_pool* q = const_cast<fq3_kv_pool*>(p);
    std::lock_guard<std::mutex> lk(q->mu);
    if (n_blocks) *n_blocks = p->n_blocks;
    if (n_free) *n_free = (int)p->free_list.size();
    if (high_water) *high_water = p->in_use_high;
    if (bytes_per_block) *bytes_per_block = (int64_t)2 * p->n_layers * p->blk_elems * p->esz;
    return FQ3_OK;
}

struct KvTableVals { int v[128]; };
__global__ void kv_table_write_kernel(int* table, KvTableVals vals, int first, int count) {
    const int i = threadIdx.x;
    if (i < count) table[first + i] = vals.v[i];
}
// entries [first, first + count) of the device table <- ids (by value in the launch: nothing on the host is read later)
static void table_write(fq3_ctx* c, const int* ids, int first, int count, hipStream_t s) {
    for (int o = 0; o < count; o += 128) {
        KvTableVals t{};
        const int n = std::min(128, count - o);
        for (int i = 0; i < n; ++i) t.v[i] = ids[o + i];
        hipLaunchKernelGGL(kv_table_write_kernel, dim3(1), dim3(128), 0, s, c->tk.d_table, t, first + o, n);
    }
}

static int kv_ensure(fq3_ctx* c, int n_pos, hipStream_t s, bool sync_copy);
int fq3_kv_ensure_(fq3_ctx* c, int n_pos, hipStream_t s) { return kv_ensure(c, n_pos, s, false); }
static int kv_ensure(fq3_ctx* c, int n_pos, hipStream_t s, bool sync_copy) {
    StackBufs& b = c->tk;
    int need = (std::min(n_pos, b.max_seq) + kKeysPerTile - 1) / kKeysPerTile;
    need = std::min(need, b.max_blocks);
    const int have = (int)b.blocks.size();
    if (need <= have) return 0;
    fq3_kv_pool* p = b.pool;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if ((int)p->free_list.size() < need - have) {
            char m[200];
            snprintf(m, sizeof m, "KV pool exhausted: %d more block(s) of 64 keys needed, %d free of %d", need - have, (int)p->free_list.size(), p->n_blocks);
            return fail(FQ3_ENOMEM, m);
        }
        for (int i = have; i < need; ++i) { b.blocks.push_back(p->free_list.back()); p->free_list.pop_back(); }
        p->in_use_high = std::max(p->in_use_high, p->n_blocks - (int)p->free_list.size());
    }
    if (sync_copy) HIPCHK(hipMemcpy(b.d_table + have, b.blocks.data() + have, (size_t)(need - have) * sizeof(int), hipMemcpyHostToDevice));
    else table_write(c, b.blocks.data() + have, have, need - have, s);
    return 0;
}

extern "C" int fq3_kv_release(fq3_ctx* c, int keep_positions) {
    if (!c) return fail(FQ3_EINVAL, "null ctx");
    StackBufs& b = c->tk;
    const int keep = std::max(0, std::min((int)b.blocks.size(), (keep_positions + kKeysPerTile - 1) / kKeysPerTile));
    if (keep >= (int)b.blocks.size()) return FQ3_OK;
    std::lock_guard<std::mutex> lk(b.pool->mu);
    while ((int)b.blocks.size() > keep) { b.pool->free_list.push_back(b.blocks.back()); b.blocks.pop_back(); }
    return FQ3_OK;                      // (the stale device entries are never attended: keys >= the next sequence's position)
}

extern "C" int fq3_kv_blocks(const fq3_ctx* c) { return c ? (int)c->tk.blocks.size() : 0; }

extern "C" int fq3_kv_reserve(fq3_ctx* c, int n_positions, void* stream) {
    if (!c) return fail(FQ3_EINVAL, "null ctx");
    if (n_positions < 0) return fail(FQ3_EINVAL, "fq3_kv_reserve: negative position count");
    return fq3_kv_ensure_(c, n_positions, (hipStream_t)stream);
}

static int ctx_build(fq3_ctx* c, const fq3_config* cfg, fq3_kv_pool* pool);
static int ctx_create_(const fq3_config* cfg, fq3_kv_pool* pool, fq3_ctx** out) {
    if (!cfg || !out) return fail(FQ3_EINVAL, "null argument");
    if (int r = cfg_check(cfg)) return r;
    if (pool && (pool->dtype != cfg->dtype || pool->n_layers != cfg->talker.n_layers || pool->n_kv != cfg->talker.n_kv_heads))
        return fail(FQ3_EINVAL, "fq3_ctx_create_pooled: the pool was built for another dtype / layer count / kv head count");
    fq3_ctx* c = new fq3_ctx();
    if (int r = ctx_build(c, cfg, pool)) { const std::string m = g_err; fq3_ctx_destroy(c); g_err = m; return r; }   // no half-built context leaks
    *out = c;
    return FQ3_OK;
}
extern "C" int fq3_ctx_create(const fq3_config* cfg, fq3_ctx** out) { return ctx_create_(cfg, nullptr, out); }
extern "C" int fq3_ctx_create_pooled(const fq3_config* cfg, fq3_kv_pool* pool, fq3_ctx** out) {
    if (!pool) return fail(FQ3_EINVAL, "null pool");
    return ctx_create_(cfg, pool, out);
}

static int ctx_build(fq3_ctx* c, const fq3_config* cfg, fq3_kv_pool* pool) {
    c->cfg = *cfg;
    c->esz = cfg->dtype == FQ3_BF16 ? 2 : 4;
    const fq3_stack_dims &t = cfg->talker, &p = cfg->predictor;
    const int pred_seq = cfg->num_code_groups + 1;          // 2 + 15 (predictor_graph.py:46)
    auto alloc_stack = [&](StackBufs& b, const fq3_stack_dims& d, int max_seq) -> int {
        b.max_seq = max_seq;
        b.workers = std::min(kMaxWorkers, (max_seq + kKeysPerTile - 1) / kKeysPerTile);
        b.k.resize(d.n_layers); b.v.resize(d.n_layers);
        const size_t bytes = (size_t)d.n_kv_heads * max_seq * kHeadDim * c->esz;
        for (int i = 0; i < d.n_layers; ++i) {
            if (int r = dmalloc(c, &b.k[i], bytes)) return r;
            if (int r = dmalloc(c, &b.v[i], bytes)) return r;
        }
        return 0;
    };
    int r;
    {   // talker: paged.  Private pool (fq3_ctx_create): every block taken now = a static cache of max_seq_len slots
        StackBufs& b = c->tk;
        b.max_seq = cfg->max_seq_len;
        b.max_blocks = (cfg->max_seq_len + kKeysPerTile - 1) / kKeysPerTile;
        b.workers = std::min(kMaxWorkers, b.max_blocks);
        if (!pool) {
            if ((r = pool_create_(cfg, b.max_blocks, &pool))) return r;
            pool->is_private = true;
        }
        b.pool = pool;
        { std::lock_guard<std::mutex> lk(pool->mu); ++pool->users; }
        b.k = pool->k; b.v = pool->v;
        if ((r = dmalloc(c, (void**)&b.d_table, (size_t)b.max_blocks * sizeof(int)))) return r;
        if (pool->is_private && (r = kv_ensure(c, cfg->max_seq_len, nullptr, true))) return r;
    }
    if ((r = alloc_stack(c->pk, p, pred_seq))) return r;
    const int Hm = std::max(t.hidden, p.hidden), Im = std::max(t.inter, p.inter);
    const int qkvm = std::max(t.n_heads + 2 * t.n_kv_heads, p.n_heads + 2 * p.n_kv_heads) * kHeadDim;
    const int Vm = std::max(t.vocab, p.vocab);
    const int G = cfg->num_code_groups;
    const int frames = cfg->max_frames > 0 ? cfg->max_frames : 4096;
    c->cfg.max_frames = frames;
    if ((r = dmalloc(c, &c->h, (size_t)Hm * c->esz))) return r;
    if ((r = dmalloc(c, &c->h2, (size_t)Hm * c->esz))) return r;
    if ((r = dmalloc(c, &c->qkv2, (size_t)qkvm * c->esz))) return r;
    if ((r = dmalloc(c, &c->attn_out, (size_t)qkvm * c->esz))) return r;
    if ((r = dmalloc(c, &c->attn_out2, (size_t)qkvm * c->esz))) return r;
    if ((r = dmalloc(c, &c->act2, (size_t)Im * c->esz))) return r;
    if ((r = dmalloc(c, &c->pred_x2, (size_t)p.hidden * c->esz))) return r;
    if ((r = dmalloc(c, &c->xin, (size_t)Hm * c->esz))) return r;
    if ((r = dmalloc(c, &c->tmp_hidden, (size_t)Hm * c->esz))) return r;
    if ((r = dmalloc(c, &c->qkv, (size_t)qkvm * c->esz))) return r;
    if ((r = dmalloc(c, &c->act, (size_t)Im * c->esz))) return r;
    if ((r = dmalloc(c, &c->logits, (size_t)Vm * c->esz))) return r;
    if ((r = dmalloc(c, &c->past_hidden, (size_t)t.hidden * c->esz))) return r;
    if ((r = dmalloc(c, &c->pred_in, (size_t)2 * t.hidden * c->esz))) return r;
    if ((r = dmalloc(c, &c->pred_next, (size_t)t.hidden * c->esz))) return r;
    if ((r = dmalloc(c, &c->pred_x, (size_t)p.hidden * c->esz))) return r;
    if ((r = dmalloc(c, &c->plogits, (size_t)(G - 1) * p.vocab * c->esz))) return r;
    const int maxkv = std::max(t.n_kv_heads, p.n_kv_heads);
    c->part_stride = (size_t)maxkv * kMaxWorkers * 4 * kPartStride;
    if ((r = dmalloc(c, (void**)&c->part, 2 * c->part_stride * sizeof(float)))) return r;
    if ((r = dmalloc(c, (void**)&c->rope_now, 2 * 64 * sizeof(float)))) return r;
    if ((r = dmalloc(c, (void**)&c->seen_api, kMaxVocab))) return r;
    if ((r = dmalloc(c, (void**)&c->st, sizeof(DecodeState)))) return r;
    if ((r = dmalloc(c, (void**)&c->seen, kMaxVocab))) return r;
    if ((r = dmalloc(c, (void**)&c->codes, (size_t)(frames + 1) * G * sizeof(int)))) return r;
    if ((r = dmalloc(c, (void**)&c->ids64, (size_t)64 * sizeof(int64_t)))) return r;
    if ((r = dmalloc(c, (void**)&c->d_pemb, (size_t)64 * sizeof(void*)))) return r;
    HIPCHK(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    return FQ3_OK;
}

// Kernel-variant switches (parity tests exercise every variant; the defaults are the measured-fastest ones).
extern "C" int fq3_set_option(fq3_ctx* c, const char* key, int value) {
    if (!c || !key) return fail(FQ3_EINVAL, "null argument");
    const std::string k(key);
    if (k == "weight_nt") c->opt_nt = value;                   // 0 none, 1 talker (default), 2 all: non-temporal weight loads
    else if (k == "pred_m2") c->opt_m2 = value;                // predictor two-token prefill as one M = 2 pass (default 1)
    else if (k == "pred_attn") c->opt_pred_attn = value;       // one-wave predictor attention (default 1)
    else if (k == "rows_per_wave_max") c->opt_rmax = value;    // GEMV rows per wave cap (default 2)
    else if (k == "prefill_mode") c->prefill_mode = value;     // 0 matrix-core prefill, 1 token walk
    else if (k == "flash_prefill") c->opt_flash_prefill = value;   // bf16 prefill attention on MFMA (default 1)
    else if (k == "flash_small") c->opt_flash_small = value;   // <= 256-row prompts: resident key tiles + packed sequences in one launch (default 1; bit-identical to 0)
    else if (k == "swiglu_tile") c->opt_swiglu_tile = value;   // many-row prefills: SwiGLU in the ring tile's epilogue over the interleaved gate | up copy (default 1; bit-identical)
    else if (k == "packed_weights") c->opt_packed = value;     // weight-stationary GEMMs on the fragment-major weight copies (default 1; bit-identical)
    else if (k == "skinny_gemm") c->opt_no_skinny = !value;    // weight-stationary short-prompt prefill GEMMs (default 1)
    else return fail(FQ3_EINVAL, "unknown option: " + k);
    fq3_graph_reset(c);
    return FQ3_OK;
}

extern "C" int fq3_graph_reset(fq3_ctx* c) {
    if (!c) return fail(FQ3_EINVAL, "null ctx");
    if (c->exec) { (void)hipGraphExecDestroy(c->exec); c->exec = nullptr; }
    if (c->graph) { (void)hipGraphDestroy(c->graph); c->graph = nullptr; }
    return FQ3_OK;
}

// ---- fragment-major copies of the layer matrices (fq3_ctx.h) ----
namespace {
struct PackedEntry { void* p; int refs; int N, K; };
std::mutex g_packed_mu;
std::map<std::pair<const void*, int>, PackedEntry> g_packed;
}
const void* fq3_packed_find_(const void* W, int kind) {
    std::lock_guard<std::mutex> lk(g_packed_mu);
    auto it = g_packed.find({W, kind});
    return it == g_packed.end() ? nullptr : it->second.p;
}
int fq3_packed_acquire_(const void* W, int N, int K, int kind) {
    if (!W || !skinny_pack_ok(N, K) || (kind == 1 && (N / 2) % 8) || (kind == 2 && N % 512)) return FQ3_EUNSUPPORTED;
    std::lock_guard<std::mutex> lk(g_packed_mu);
    auto it = g_packed.find({W, kind});
    if (it != g_packed.end()) {
        if (it->second.N != N || it->second.K != K) return fail(FQ3_EINVAL, "a matrix is bound with two different shapes");
        ++it->second.refs;
        return FQ3_OK;
    }
    void* p = nullptr;
    if (hipMalloc(&p, (size_t)N * K * 2) != hipSuccess) { (void)hipGetLastError(); return fail(FQ3_ENOMEM, "no device memory for the fragment-major weight copy"); }
    // kind 0: fragment-major 16-row blocks; 1: fragment-major 8 gate + 8 up rows; 2: ROW-major, [gate | up] interleaved in 16-row blocks
    skinny_pack(reinterpret_cast<const bf16_t*>(W), reinterpret_cast<bf16_t*>(p), N, K, kind == 1 ? N / 2 : (kind == 2 ? -(N / 2) : 0), nullptr);
    if (hipStreamSynchronize(nullptr) != hipSuccess) { (void)hipFree(p); return fail(FQ3_EHIP, "packing a weight matrix failed"); }
    g_packed[{W, kind}] = PackedEntry{p, 1, N, K};
    return FQ3_OK;
}
void fq3_packed_release_(const void* W, int kind) {
    std::lock_guard<std::mutex> lk(g_packed_mu);
    auto it = g_packed.find({W, kind});
    if (it == g_packed.end()) return;
    if (--it->second.refs <= 0) { (void)hipFree(it->second.p); g_packed.erase(it); }
}
static void packed_release_all(fq3_ctx* c) {
    for (auto& r : c->packed_refs) fq3_packed_release_(r.first, r.second);
    c->packed_refs.clear();
}
// references for every layer matrix and head of a bf16 context whose shapes the weight-stationary kernels serve (others stay row-major)
static int packed_acquire_all(fq3_ctx* c) {
    if (c->cfg.dtype != FQ3_BF16) return FQ3_OK;
    auto take = [&](const void* W, int N, int K, int kind) -> int {
        const int r = fq3_packed_acquire_(W, N, K, kind);
        if (r == FQ3_OK) c->packed_refs.emplace_back(W, kind);
        return r == FQ3_EUNSUPPORTED ? FQ3_OK : r;
    };
    auto stack = [&](const std::vector<fq3_layer_weights>& L, const fq3_stack_dims& d) -> int {
        const int qd = d.n_heads * kHeadDim, kvd = d.n_kv_heads * kHeadDim;
        for (const fq3_layer_weights& l : L) {
            if (int r = take(l.qkv, qd + 2 * kvd, d.hidden, 0)) return r;
            if (int r = take(l.o, d.hidden, qd, 0)) return r;
            if (int r = take(l.gate_up, 2 * d.inter, d.hidden, 1)) return r;       // the 8 + 8 pairing of the weight-stationary SwiGLU GEMM (> 32 lanes, prefill)
            if (int r = take(l.gate_up, 2 * d.inter, d.hidden, 0)) return r;       // 16-row blocks for the panel kernels of <= 32 lanes
            if (int r = take(l.down, d.hidden, d.inter, 0)) return r;
        }
        return FQ3_OK;
    };
    if (int r = stack(c->tl, c->cfg.talker)) return r;
    if (int r = stack(c->pl, c->cfg.predictor)) return r;
    // the talker's gate | up once more, row-major with the halves interleaved: the many-row prefill's ring tile with SwiGLU in its epilogue
    for (const fq3_layer_weights& l : c->tl) if (int r = take(l.gate_up, 2 * c->cfg.talker.inter, c->cfg.talker.hidden, 2)) return r;
    if (int r = take(c->wt.codec_head, c->cfg.talker.vocab, c->cfg.talker.hidden, 0)) return r;
    for (const void* h : c->lmh) if (int r = take(h, c->cfg.predictor.vocab, c->cfg.predictor.hidden, 0)) return r;
    return FQ3_OK;
}

extern "C" int fq3_ctx_destroy(fq3_ctx* c) {
    if (!c) return FQ3_OK;
    (void)hipDeviceSynchronize();
    fq3_graph_reset(c);
    packed_release_all(c);
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    if (fq3_kv_pool* p = c->tk.pool) {
        (void)fq3_kv_release(c, 0);
        bool last;
        { std::lock_guard<std::mutex> lk(p->mu); last = --p->users == 0; }
        if (p->is_private && last) pool_free_(p);
    }
    for (void* p : c->allocs) (void)hipFree(p);
    delete c;
    return FQ3_OK;
}

extern "C" int fq3_bind_weights(fq3_ctx* c, const fq3_weight_table* w) {
    if (!c || !w) return fail(FQ3_EINVAL, "null argument");
    const int G = c->cfg.num_code_groups;
    if (!w->talker_layers || !w->predictor_layers || !w->codec_embedding || !w->codec_head || !w->talker_final_norm ||
        !w->predictor_final_norm || !w->predictor_embeddings || !w->lm_heads || !w->talker_cos || !w->talker_sin ||
        !w->pred_cos || !w->pred_sin)
        return fail(FQ3_EINVAL, "weight table has null entries");
    if (c->cfg.has_projection && !w->proj_w) return fail(FQ3_EINVAL, "has_projection set but proj_w is null");
    if (w->talker_rope_len < c->cfg.max_seq_len || w->pred_rope_len < G + 1)
        return fail(FQ3_EINVAL, "rope tables shorter than the caches");
    c->wt = *w;
    c->tl.assign(w->talker_layers, w->talker_layers + c->cfg.talker.n_layers);
    c->pl.assign(w->predictor_layers, w->predictor_layers + c->cfg.predictor.n_layers);
    c->pemb.assign(w->predictor_embeddings, w->predictor_embeddings + (G - 1));
    c->lmh.assign(w->lm_heads, w->lm_heads + (G - 1));
    for (auto& l : c->tl) if (!l.qkv || !l.o || !l.gate_up || !l.down || !l.input_norm || !l.post_norm || !l.q_norm || !l.k_norm)
        return fail(FQ3_EINVAL, "talker layer has null weights");
    for (auto& l : c->pl) if (!l.qkv || !l.o || !l.gate_up || !l.down || !l.input_norm || !l.post_norm || !l.q_norm || !l.k_norm)
        return fail(FQ3_EINVAL, "predictor layer has null weights");
    HIPCHK(hipMemcpy(c->d_pemb, c->pemb.data(), (G - 1) * sizeof(void*), hipMemcpyHostToDevice));
    // fragment-major copies for the weight-stationary GEMMs (a re-bind first returns the previous table's references)
    (void)hipDeviceSynchronize();
    packed_release_all(c);
    if (int r = packed_acquire_all(c)) { packed_release_all(c); return r; }
    c->bound = true;
    fq3_graph_reset(c);
    return FQ3_OK;
}

// -------------------------------------------------------------------------------------------------
// GEMV dispatch
// -------------------------------------------------------------------------------------------------
static thread_local int g_rmax = 2;      // set from the context before a launch sequence (fq3_set_option "rows_per_wave_max")
template <typename T, int NCH, int PRO, int EPI, bool NT, int M = 1>
static void launch_gemv_n(const GemvArgs& a, hipStream_t s) {
    // rows per wave: 2 for the big matrices (512-768 workgroups), 1 when N <= 1024 or a row is long
    const int rmax = g_rmax;
    int R = (a.N + 1023) / 1024;
    if (R > MaxRows<NCH, EPI>::v) R = MaxRows<NCH, EPI>::v;
    if (rmax > 0 && R > rmax) R = rmax;
    if (R < 1) R = 1;
    const int grid = (a.N + 4 * R - 1) / (4 * R);
    const size_t shm = PRO == PRO_COMBINE ? (size_t)M * a.K * sizeof(float) : 0;
    if constexpr (MaxRows<NCH, EPI>::v >= 2) {
        if (R == 2) { hipLaunchKernelGGL((gemv_kernel<T, NCH, PRO, EPI, NT, M, 2>), dim3(grid), dim3(256), shm, s, a); return; }
    }
    hipLaunchKernelGGL((gemv_kernel<T, NCH, PRO, EPI, NT, M, 1>), dim3(grid), dim3(256), shm, s, a);
}
// two-token launches: code predictor only (default cache policy).  Inner dimensions: a normalising GEMV reads the hidden size
// (<= 2048: four 512-element chunks per lane), the others q_dim / the intermediate size (<= 3072 for a two-token pass: the row
// registers of both tokens would not fit beyond -- the 12-chunk instantiations of earlier rounds spilled and were never reached by a
// supported model).
template <typename T, int PRO, int EPI>
static int launch_gemv2_t(const GemvArgs& a, hipStream_t s) {
    const int need = (a.K + 511) / 512;
    if (need <= 1) launch_gemv_n<T, 1, PRO, EPI, false, 2>(a, s);
    else if (need <= 2) launch_gemv_n<T, 2, PRO, EPI, false, 2>(a, s);
    else if (need <= 4) launch_gemv_n<T, 4, PRO, EPI, false, 2>(a, s);
    else if (PRO != PRO_NORM && need <= 6) { if constexpr (PRO != PRO_NORM) launch_gemv_n<T, 6, PRO, EPI, false, 2>(a, s); }
    else return fail(FQ3_EUNSUPPORTED, PRO == PRO_NORM ? "code predictor hidden size above 2048" : "code predictor GEMV inner dimension above 3072");
    return 0;
}
template <int PRO, int EPI>
static int launch_gemv2(const fq3_ctx* c, const GemvArgs& a, hipStream_t s) {
    g_rmax = c->opt_rmax;
    return c->cfg.dtype == FQ3_BF16 ? launch_gemv2_t<bf16_t, PRO, EPI>(a, s) : launch_gemv2_t<float, PRO, EPI>(a, s);
}
template <typename T, int PRO, int EPI, bool NT>
static int launch_gemv_t(const GemvArgs& a, hipStream_t s) {
    const int need = (a.K + 511) / 512;
    if (need <= 1) launch_gemv_n<T, 1, PRO, EPI, NT>(a, s);
    else if (need <= 2) launch_gemv_n<T, 2, PRO, EPI, NT>(a, s);
    else if (need <= 4) launch_gemv_n<T, 4, PRO, EPI, NT>(a, s);
    else if constexpr (PRO == PRO_NORM) return fail(FQ3_EUNSUPPORTED, "hidden size above 2048");      // a normalising GEMV reads K = hidden
    else if (need <= 6) launch_gemv_n<T, 6, PRO, EPI, NT>(a, s);
    else if (need <= 12) launch_gemv_n<T, 12, PRO, EPI, NT>(a, s);
    else return fail(FQ3_EUNSUPPORTED, "GEMV inner dimension above 6144");
    return 0;
}
template <int PRO, int EPI>
static int launch_gemv(const fq3_ctx* c, const GemvArgs& a, bool nt, hipStream_t s) {
    g_rmax = c->opt_rmax;
    if (c->cfg.dtype == FQ3_BF16)
        return nt ? launch_gemv_t<bf16_t, PRO, EPI, true>(a, s) : launch_gemv_t<bf16_t, PRO, EPI, false>(a, s);
    return nt ? launch_gemv_t<float, PRO, EPI, true>(a, s) : launch_gemv_t<float, PRO, EPI, false>(a, s);
}

template <typename T, bool PAGED>
static void launch_attn_p(const AttnArgs& a, int rep, int workers, hipStream_t s) {
    dim3 grid(a.n_kv, workers);
    if (rep == 1) hipLaunchKernelGGL((attn_decode_kernel<T, 1, PAGED>), grid, dim3(256), 0, s, a);
    else if (rep == 2) hipLaunchKernelGGL((attn_decode_kernel<T, 2, PAGED>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_decode_kernel<T, 4, PAGED>), grid, dim3(256), 0, s, a);
}
template <typename T>
static void launch_attn_t(const AttnArgs& a, int rep, int workers, hipStream_t s) {
    if (a.table) launch_attn_p<T, true>(a, rep, workers, s); else launch_attn_p<T, false>(a, rep, workers, s);
}

// One token through every layer of a stack (no final norm).  Layer 0 reads its input (and its
// residual) from x0 [+ (*x0_idx) * hidden]; later layers work in place on c->h.
struct StepSrc { const void* x0; const int* pos_ptr; int pos_imm; bool kv_only_tail = false; };

static int run_stack(fq3_ctx* c, bool talker, const StepSrc& src, hipStream_t s) {
    const fq3_stack_dims& d = talker ? c->cfg.talker : c->cfg.predictor;
    const std::vector<fq3_layer_weights>& L = talker ? c->tl : c->pl;
    StackBufs& kv = talker ? c->tk : c->pk;
    const int rep = d.n_heads / d.n_kv_heads;
    const int q_dim = d.n_heads * kHeadDim, kv_dim = d.n_kv_heads * kHeadDim;
    // talker weights stream once per frame (non-temporal); predictor weights are re-read 16x per frame and
    // should stay in the 256 MB Infinity Cache (default policy).  fq3_set_option("weight_nt"): 0 none, 1 talker, 2 all.
    const bool nt = c->opt_nt == 2 || (c->opt_nt == 1 && talker);
    // RoPE row: immediate position -> table row chosen on the host; device position -> the row the
    // frame's embed_sum kernel staged in rope_now
    const float *cos_row, *sin_row;
    if (src.pos_ptr) { cos_row = c->rope_now; sin_row = c->rope_now + 64; }
    else {
        const int rl = talker ? c->wt.talker_rope_len : c->wt.pred_rope_len;
        int rp = src.pos_imm + (talker ? c->rope_delta : 0);
        rp = rp < 0 ? 0 : (rp >= rl ? rl - 1 : rp);
        cos_row = (talker ? c->wt.talker_cos : c->wt.pred_cos) + (size_t)rp * 64;
        sin_row = (talker ? c->wt.talker_sin : c->wt.pred_sin) + (size_t)rp * 64;
    }
    for (int i = 0; i < d.n_layers; ++i) {
        const fq3_layer_weights& w = L[i];
        const void* xin = i == 0 ? src.x0 : c->h;
        GemvArgs g{};
        g.eps = d.rms_eps;
        // 1. norm + qkv
        g.W = w.qkv; g.N = q_dim + 2 * kv_dim; g.K = d.hidden; g.x = xin;
        g.norm_w = w.input_norm; g.y = c->qkv;
        if (int r = launch_gemv<PRO_NORM, EPI_STORE>(c, g, nt, s)) return r;
        // the first predictor prefill token only feeds K/V to later passes: its last layer needs nothing after the
        // cache append (which the attention launch performs)
        const bool tail_skip = src.kv_only_tail && i == d.n_layers - 1;
        // 2+3. attention, then o_proj + residual
        GemvArgs o{};
        o.W = w.o; o.N = d.hidden; o.K = q_dim; o.y = c->h; o.res = xin; o.rep = rep;
        AttnArgs a{};
        a.qkv = c->qkv; a.q_norm_w = w.q_norm; a.k_norm_w = w.k_norm; a.eps = d.rms_eps;
        a.cos_row = cos_row; a.sin_row = sin_row;
        a.kcache = kv.k[i]; a.vcache = kv.v[i]; a.max_seq = kv.max_seq;
        a.table = kv.d_table; a.blk_stride = kv.pool ? (int)kv.pool->blk_elems : 0;       // talker: paged; predictor: contiguous (null table)
        a.pos_ptr = src.pos_ptr; a.pos_imm = src.pos_imm;
        a.done_ptr = src.pos_ptr ? &c->st->done : nullptr;              // the fused loop (device position): a finished loop never appends
        a.n_pad = talker ? c->n_pad : 0;
        a.n_kv = d.n_kv_heads; a.part = c->part;
        a.scale = 1.0f / sqrtf((float)kHeadDim);
        a.rep = rep; a.out = c->attn_out;
        const bool pred_attn = c->opt_pred_attn && !talker && !src.pos_ptr && src.pos_imm <= 16;
        if (pred_attn) {
            // short context: one wave per q head, final head output written directly -> plain o_proj, no merge
            if (c->cfg.dtype == FQ3_BF16) hipLaunchKernelGGL((attn_pred_kernel<bf16_t>), dim3(d.n_heads), dim3(64), 0, s, a);
            else hipLaunchKernelGGL((attn_pred_kernel<float>), dim3(d.n_heads), dim3(64), 0, s, a);
            if (tail_skip) break;
            o.x = c->attn_out;
            if (int r = launch_gemv<PRO_PLAIN, EPI_RESIDUAL>(c, o, nt, s)) return r;
        } else {
            if (c->cfg.dtype == FQ3_BF16) launch_attn_t<bf16_t>(a, rep, kv.workers, s);
            else launch_attn_t<float>(a, rep, kv.workers, s);
            if (tail_skip) break;
            o.part = c->part; o.n_part = kv.workers;
            if (int r = launch_gemv<PRO_COMBINE, EPI_RESIDUAL>(c, o, nt, s)) return r;
        }
        // 4. norm + gate/up + SwiGLU
        GemvArgs m{};
        m.eps = d.rms_eps; m.W = w.gate_up; m.N = d.inter; m.K = d.hidden; m.x = c->h;
        m.norm_w = w.post_norm; m.y = c->act; m.up_off = d.inter;
        if (int r = launch_gemv<PRO_NORM, EPI_SWIGLU>(c, m, nt, s)) return r;
        // 5. down + residual
        GemvArgs dn{};
        dn.W = w.down; dn.N = d.hidden; dn.K = d.inter; dn.x = c->act; dn.y = c->h; dn.res = c->h;
        if (int r = launch_gemv<PRO_PLAIN, EPI_RESIDUAL>(c, dn, nt, s)) return r;
    }
    return 0;
}

// The code predictor's two-token prefill (predictor_graph.py:121-128) as ONE pass over the weights: every GEMV
// is launched with M = 2 (weights read once), attention runs once per token (token B attends to A and B).
// Token A = x_a at cache slot 0, token B = x_b at slot 1; results: c->h (token A), c->h2 (token B).
static int run_predictor_pair(fq3_ctx* c, const void* x_a, const void* x_b, hipStream_t s) {
    const fq3_stack_dims& d = c->cfg.predictor;
    StackBufs& kv = c->pk;
    const int rep = d.n_heads / d.n_kv_heads;
    const int q_dim = d.n_heads * kHeadDim, kv_dim = d.n_kv_heads * kHeadDim;
    for (int i = 0; i < d.n_layers; ++i) {
        const fq3_layer_weights& w = c->pl[i];
        const void* xa = i == 0 ? x_a : c->h;
        const void* xb = i == 0 ? x_b : c->h2;
        GemvArgs g{};
        g.eps = d.rms_eps; g.W = w.qkv; g.N = q_dim + 2 * kv_dim; g.K = d.hidden; g.x = xa; g.x2 = xb;
        g.norm_w = w.input_norm; g.y = c->qkv; g.y2 = c->qkv2;
        if (int r = launch_gemv2<PRO_NORM, EPI_STORE>(c, g, s)) return r;
        for (int m = 0; m < 2; ++m) {
            AttnArgs a{};
            a.qkv = m == 0 ? c->qkv : c->qkv2; a.q_norm_w = w.q_norm; a.k_norm_w = w.k_norm; a.eps = d.rms_eps;
            a.cos_row = c->wt.pred_cos + (size_t)m * 64; a.sin_row = c->wt.pred_sin + (size_t)m * 64;
            a.kcache = kv.k[i]; a.vcache = kv.v[i]; a.max_seq = kv.max_seq;
            a.pos_ptr = nullptr; a.pos_imm = m; a.n_pad = 0; a.n_kv = d.n_kv_heads;
            a.part = c->part + (size_t)m * c->part_stride; a.scale = 1.0f / sqrtf((float)kHeadDim);
            a.rep = rep; a.out = m == 0 ? c->attn_out : c->attn_out2;
            if (c->opt_pred_attn) {
                if (c->cfg.dtype == FQ3_BF16) hipLaunchKernelGGL((attn_pred_kernel<bf16_t>), dim3(d.n_heads), dim3(64), 0, s, a);
                else hipLaunchKernelGGL((attn_pred_kernel<float>), dim3(d.n_heads), dim3(64), 0, s, a);
            } else if (c->cfg.dtype == FQ3_BF16) launch_attn_t<bf16_t>(a, rep, kv.workers, s);
            else launch_attn_t<float>(a, rep, kv.workers, s);
        }
        GemvArgs o{};
        o.W = w.o; o.N = d.hidden; o.K = q_dim; o.y = c->h; o.y2 = c->h2; o.res = xa; o.res2 = xb; o.rep = rep;
        if (c->opt_pred_attn) {
            o.x = c->attn_out; o.x2 = c->attn_out2;
            if (int r = launch_gemv2<PRO_PLAIN, EPI_RESIDUAL>(c, o, s)) return r;
        } else {
            o.part = c->part; o.part_stride2 = c->part_stride; o.n_part = kv.workers;
            if (int r = launch_gemv2<PRO_COMBINE, EPI_RESIDUAL>(c, o, s)) return r;
        }
        GemvArgs m{};
        m.eps = d.rms_eps; m.W = w.gate_up; m.N = d.inter; m.K = d.hidden; m.x = c->h; m.x2 = c->h2;
        m.norm_w = w.post_norm; m.y = c->act; m.y2 = c->act2; m.up_off = d.inter;
        if (int r = launch_gemv2<PRO_NORM, EPI_SWIGLU>(c, m, s)) return r;
        GemvArgs dn{};
        dn.W = w.down; dn.N = d.hidden; dn.K = d.inter; dn.x = c->act; dn.x2 = c->act2; dn.y = c->h; dn.y2 = c->h2;
        dn.res = c->h; dn.res2 = c->h2;
        if (int r = launch_gemv2<PRO_PLAIN, EPI_RESIDUAL>(c, dn, s)) return r;
    }
    return 0;
}

#define NEED_BOUND(c) do { if (!(c)) return fail(FQ3_EINVAL, "null ctx"); if (!(c)->bound) return fail(FQ3_ESTATE, "weights not bound"); } while (0)
#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

extern "C" int fq3_set_generation_state(fq3_ctx* c, int n_pad, int rope_delta) {
    if (!c) return fail(FQ3_EINVAL, "null ctx");
    if (n_pad < 0 || n_pad >= c->cfg.max_seq_len) return fail(FQ3_EINVAL, "n_pad out of range");
    if (c->n_pad != n_pad || c->rope_delta != rope_delta) fq3_graph_reset(c);   // baked into captured launches
    c->n_pad = n_pad; c->rope_delta = rope_delta;
    return FQ3_OK;
}

// rows [0, L) of one layer between the paged cache and a dense [n_kv][L][128] tensor (the HF layout), 16 bytes per thread
template <typename T, bool TO_CACHE>
__global__ __launch_bounds__(256) void kv_paged_copy_kernel(T* cache, const int* table, int blk_stride, T* dense, int n_kv, int L) {
    constexpr int EPC = 16 / (int)sizeof(T), CPR = kHeadDim / EPC;          // 16-byte chunks per row
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_kv * L * CPR) return;
    const int ch = i % CPR, row = i / CPR, key = row % L, h = row / L;
    T* cp = cache + (size_t)table[key / kKeysPerTile] * blk_stride + ((size_t)h * kKeysPerTile + key % kKeysPerTile) * kHeadDim + ch * EPC;
    T* dp = dense + ((size_t)h * L + key) * kHeadDim + ch * EPC;
    if (TO_CACHE) *reinterpret_cast<u32x4*>(cp) = *reinterpret_cast<const u32x4*>(dp);
    else *reinterpret_cast<u32x4*>(dp) = *reinterpret_cast<const u32x4*>(cp);
}
template <typename T, bool TO_CACHE>
static void kv_paged_copy(fq3_ctx* c, int layer, void* k, void* v, int L, hipStream_t s) {
    const int nk = c->cfg.talker.n_kv_heads, n = nk * L * (kHeadDim * (int)sizeof(T) / 16);
    if (n <= 0) return;
    const int bs = (int)c->tk.pool->blk_elems;
    hipLaunchKernelGGL((kv_paged_copy_kernel<T, TO_CACHE>), dim3((n + 255) / 256), dim3(256), 0, s, (T*)c->tk.k[layer], c->tk.d_table, bs, (T*)k, nk, L);
    hipLaunchKernelGGL((kv_paged_copy_kernel<T, TO_CACHE>), dim3((n + 255) / 256), dim3(256), 0, s, (T*)c->tk.v[layer], c->tk.d_table, bs, (T*)v, nk, L);
}

extern "C" int fq3_kv_import(fq3_ctx* c, int layer, const void* k, const void* v, int L, void* stream) {
    if (!c || !k || !v) return fail(FQ3_EINVAL, "null argument");
    if (layer < 0 || layer >= c->cfg.talker.n_layers) return fail(FQ3_EINVAL, "layer out of range");
    if (L > c->cfg.max_seq_len) {
        char b[256];
        snprintf(b, sizeof b, "Input is too long: prefill has %d tokens but max_seq_len=%d. Use shorter text or shorter reference audio.",
                 L, c->cfg.max_seq_len);
        return fail(FQ3_ETOOLONG, b);
    }
    hipStream_t s = (hipStream_t)stream;
    if (int r = fq3_kv_ensure_(c, L, s)) return r;
    if (c->cfg.dtype == FQ3_BF16) kv_paged_copy<bf16_t, true>(c, layer, const_cast<void*>(k), const_cast<void*>(v), L, s);
    else kv_paged_copy<float, true>(c, layer, const_cast<void*>(k), const_cast<void*>(v), L, s);
    LAUNCH_CHECK();
    return FQ3_OK;
}

extern "C" int fq3_kv_export(fq3_ctx* c, int layer, void* k, void* v, int L, void* stream) {
    if (!c || !k || !v) return fail(FQ3_EINVAL, "null argument");
    if (layer < 0 || layer >= c->cfg.talker.n_layers || L > c->cfg.max_seq_len) return fail(FQ3_EINVAL, "range");
    if (L > (int)c->tk.blocks.size() * kKeysPerTile) return fail(FQ3_EINVAL, "fq3_kv_export: rows beyond the blocks this context owns");
    hipStream_t s = (hipStream_t)stream;
    if (c->cfg.dtype == FQ3_BF16) kv_paged_copy<bf16_t, false>(c, layer, k, v, L, s);
    else kv_paged_copy<float, false>(c, layer, k, v, L, s);
    LAUNCH_CHECK();
    return FQ3_OK;
}

// fq3_kv_adopt between contexts of DIFFERENT pools: whole blocks [0, ceil(L / 64)) of every talker layer, block to block, in one
// launch: grid (2 * layers, blocks)
struct KvAdoptTab { void* dst[128]; const void* src[128]; };
template <typename T>
__global__ __launch_bounds__(256) void kv_adopt_kernel(KvAdoptTab t, const int* dst_table, const int* src_table, int blk_elems) {
    const int which = blockIdx.x, b = blockIdx.y;
    const u32x4* s = reinterpret_cast<const u32x4*>(reinterpret_cast<const T*>(t.src[which]) + (size_t)src_table[b] * blk_elems);
    u32x4* d = reinterpret_cast<u32x4*>(reinterpret_cast<T*>(t.dst[which]) + (size_t)dst_table[b] * blk_elems);
    const int n16 = blk_elems * (int)sizeof(T) / 16;
    for (int i = threadIdx.x; i < n16; i += 256) d[i] = s[i];
}

extern "C" int fq3_kv_adopt(fq3_ctx* dst, fq3_ctx* src, int L, void* stream) {
    if (!dst || !src) return fail(FQ3_EINVAL, "null ctx");
    if (dst == src) return fail(FQ3_EINVAL, "fq3_kv_adopt: source and destination are the same context");
    const auto &a = dst->cfg, &b = src->cfg;
    if (a.dtype != b.dtype || a.talker.n_layers != b.talker.n_layers || a.talker.n_kv_heads != b.talker.n_kv_heads)
        return fail(FQ3_EINVAL, "fq3_kv_adopt: contexts of different shape");
    if (L < 0 || L > src->tk.max_seq || L > (int)src->tk.blocks.size() * kKeysPerTile) return fail(FQ3_EINVAL, "fq3_kv_adopt: L outside the source cache");
    if (L > dst->tk.max_seq) {
        char m[256];
        snprintf(m, sizeof m, "Input is too long: prefill has %d tokens but max_seq_len=%d. Use shorter text or shorter reference audio.", L, a.max_seq_len);
        return fail(FQ3_ETOOLONG, m);
    }
    const int nl = a.talker.n_layers;
    if (2 * nl > 128) return fail(FQ3_EUNSUPPORTED, "fq3_kv_adopt: more than 64 layers");
    if (L == 0) return FQ3_OK;
    hipStream_t s = (hipStream_t)stream;
    const int nb = (L + kKeysPerTile - 1) / kKeysPerTile;
    if (dst->tk.pool == src->tk.pool) {
        // one pool: hand the blocks over.  dst returns what it holds and takes ALL of src's block ids -- the prompt's and whatever
        // src had reserved beyond it for the frames to come (fq3_kv_reserve) -- up to its own table length; its device table is
        // rewritten by one tiny launch on `stream`; src keeps nothing.  No KV row moves.
        (void)fq3_kv_release(dst, 0);
        const size_t take = std::min(src->tk.blocks.size(), (size_t)dst->tk.max_blocks);
        dst->tk.blocks.assign(src->tk.blocks.begin(), src->tk.blocks.begin() + take);
        if (take < src->tk.blocks.size()) {
            std::lock_guard<std::mutex> lk(src->tk.pool->mu);
            for (size_t i = take; i < src->tk.blocks.size(); ++i) src->tk.pool->free_list.push_back(src->tk.blocks[i]);
        }
        src->tk.blocks.clear();
        table_write(dst, dst->tk.blocks.data(), 0, (int)take, s);
        LAUNCH_CHECK();
        return FQ3_OK;
    }
    if (int r = fq3_kv_ensure_(dst, L, s)) return r;
    KvAdoptTab t{};
    for (int l = 0; l < nl; ++l) { t.dst[2 * l] = dst->tk.k[l]; t.src[2 * l] = src->tk.k[l]; t.dst[2 * l + 1] = dst->tk.v[l]; t.src[2 * l + 1] = src->tk.v[l]; }
    const dim3 grid(2 * nl, nb);
    const int be = (int)dst->tk.pool->blk_elems;
    if (a.dtype == FQ3_BF16) hipLaunchKernelGGL((kv_adopt_kernel<bf16_t>), grid, dim3(256), 0, s, t, dst->tk.d_table, src->tk.d_table, be);
    else hipLaunchKernelGGL((kv_adopt_kernel<float>), grid, dim3(256), 0, s, t, dst->tk.d_table, src->tk.d_table, be);
    LAUNCH_CHECK();
    return FQ3_OK;
}

static int final_norm(fq3_ctx* c, bool talker, const void* x, void* y, hipStream_t s) {
    const fq3_stack_dims& d = talker ? c->cfg.talker : c->cfg.predictor;
    const void* w = talker ? c->wt.talker_final_norm : c->wt.predictor_final_norm;
    if (c->cfg.dtype == FQ3_BF16)
        hipLaunchKernelGGL((rmsnorm_kernel<bf16_t>), dim3(1), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, d.hidden, d.rms_eps);
    else
        hipLaunchKernelGGL((rmsnorm_kernel<float>), dim3(1), dim3(256), 0, s, (const float*)x, (const float*)w, (float*)y, d.hidden, d.rms_eps);
    return 0;
}

extern "C" int fq3_talker_step(fq3_ctx* c, const void* embeds, int position, void* out_hidden, void* stream) {
    NEED_BOUND(c);
    if (!embeds || !out_hidden) return fail(FQ3_EINVAL, "null argument");
    if (position < 0 || position >= c->cfg.max_seq_len) return fail(FQ3_EINVAL, "position outside the static cache");
    hipStream_t s = (hipStream_t)stream;
    if (int r = fq3_kv_ensure_(c, position + 1, s)) return r;
    StepSrc src{embeds, nullptr, position};
    if (int r = run_stack(c, true, src, s)) return r;
    final_norm(c, true, c->h, out_hidden, s);
    LAUNCH_CHECK();
    return FQ3_OK;
}

extern "C" int fq3_codec_head(fq3_ctx* c, const void* hidden, void* out_logits, void* stream) {
    NEED_BOUND(c);
    if (!hidden || !out_logits) return fail(FQ3_EINVAL, "null argument");
    GemvArgs g{};
    g.W = c->wt.codec_head; g.N = c->cfg.talker.vocab; g.K = c->cfg.talker.hidden; g.x = hidden; g.y = out_logits;
    if (int r = launch_gemv<PRO_PLAIN, EPI_STORE>(c, g, true, (hipStream_t)stream)) return r;
    LAUNCH_CHECK();
    return FQ3_OK;
}

// Prefill: matrix-core GEMMs over all prompt rows (csrc/fq3_prefill.hip) by default; prefill_mode 1 walks the prompt
// token by token through the decode-step kernels (the arithmetic the decode parity tests pin; kept as a cross-check).
extern "C" int fq3_prefill(fq3_ctx* c, const void* embeds, int L, int n_pad, void* out_logits, void* out_hidden,
                           void* stream) {
    NEED_BOUND(c);
    if (!embeds || L <= 0) return fail(FQ3_EINVAL, "bad prompt");
    if (L > c->cfg.max_seq_len) {
        char b[256];
        snprintf(b, sizeof b, "Input is too long: prefill has %d tokens but max_seq_len=%d. Use shorter text or shorter reference audio.",
                 L, c->cfg.max_seq_len);
        return fail(FQ3_ETOOLONG, b);
    }
    if (n_pad < 0 || n_pad >= L) return fail(FQ3_EINVAL, "n_pad");
    if (int r = fq3_set_generation_state(c, n_pad, -n_pad)) return r;
    hipStream_t s = (hipStream_t)stream;
    if (int r = fq3_kv_ensure_(c, L, s)) return r;
    void* hid = out_hidden ? out_hidden : c->tmp_hidden;
    if (c->prefill_mode != 1 && L - n_pad >= 4) {
        // matrix-core prefill: GEMMs over all prompt rows, causal attention, KV written straight to the cache
        if (int r = fq3_prefill_mfma_(c, embeds, L, n_pad, out_logits, hid, s)) return r;
        LAUNCH_CHECK();
        return FQ3_OK;
    }
    const size_t rowb = (size_t)c->cfg.talker.hidden * c->esz;
    for (int i = n_pad; i < L; ++i) {
        StepSrc src{(const char*)embeds + rowb * i, nullptr, i};
        if (int r = run_stack(c, true, src, s)) return r;
    }
    final_norm(c, true, c->h, hid, s);
    if (out_logits) {
        GemvArgs g{};
        g.W = c->wt.codec_head; g.N = c->cfg.talker.vocab; g.K = c->cfg.talker.hidden; g.x = hid; g.y = out_logits;
        if (int r = launch_gemv<PRO_PLAIN, EPI_STORE>(c, g, true, s)) return r;
    }
    LAUNCH_CHECK();
    return FQ3_OK;
}

extern "C" int fq3_prefill_reserve(fq3_ctx* c) {
    NEED_BOUND(c);
    return fq3_prefill_reserve_(c);
}

extern "C" int fq3_prefill_batch(fq3_ctx* const* ctxs, int n, const void* const* embeds, const int* L, const int* n_pad,
                                 void* const* out_logits, void* const* out_hidden, void* stream) {
    if (!ctxs || !embeds || !L || !n_pad || !out_hidden || n < 1 || n > 64) return fail(FQ3_EINVAL, "fq3_prefill_batch: bad argument");
    long total = 0;
    bool packed = true;
    for (int q = 0; q < n; ++q) {
        fq3_ctx* c = ctxs[q];
        NEED_BOUND(c);
        if (!embeds[q] || !out_hidden[q] || L[q] <= 0) return fail(FQ3_EINVAL, "fq3_prefill_batch: bad prompt");
        if (L[q] > c->cfg.max_seq_len) {
            char b[256];
            snprintf(b, sizeof b, "Input is too long: prefill has %d tokens but max_seq_len=%d. Use shorter text or shorter reference audio.",
                     L[q], c->cfg.max_seq_len);
            return fail(FQ3_ETOOLONG, b);
        }
        if (n_pad[q] < 0 || n_pad[q] >= L[q]) return fail(FQ3_EINVAL, "n_pad");
        total += L[q];
        // one pass over the weights needs ONE set of weights, and the matrix-core path for every prompt
        const fq3_ctx* c0 = ctxs[0];
        packed = packed && c->cfg.dtype == c0->cfg.dtype && c->wt.codec_head == c0->wt.codec_head && c->tl.size() == c0->tl.size() &&
                 !c->tl.empty() && c->tl[0].qkv == c0->tl[0].qkv && c->prefill_mode != 1 && L[q] - n_pad[q] >= 4;
        for (int p = 0; p < q; ++p) if (ctxs[p] == c) return fail(FQ3_EINVAL, "fq3_prefill_batch: a context appears twice");
    }
    packed = packed && n > 1 && total <= ctxs[0]->cfg.max_seq_len;
    if (!packed) {                                       // not packable: the n single prefills
        for (int q = 0; q < n; ++q)
            if (int r = fq3_prefill(ctxs[q], embeds[q], L[q], n_pad[q], out_logits ? out_logits[q] : nullptr, out_hidden[q], stream)) return r;
        return FQ3_OK;
    }
    for (int q = 0; q < n; ++q)
        if (int r = fq3_set_generation_state(ctxs[q], n_pad[q], -n_pad[q])) return r;
    {   // the blocks of every prompt, all or nothing: a short pool leaves every context as it was
        std::vector<int> had(n);
        for (int q = 0; q < n; ++q) had[q] = (int)ctxs[q]->tk.blocks.size();
        for (int q = 0; q < n; ++q)
            if (int r = fq3_kv_ensure_(ctxs[q], L[q], (hipStream_t)stream)) {
                const std::string m = g_err;
                for (int p = 0; p < q; ++p) (void)fq3_kv_release(ctxs[p], had[p] * kKeysPerTile);
                g_err = m;
                return r;
            }
    }
    if (int r = fq3_prefill_batch_mfma_(ctxs, n, embeds, L, n_pad, out_logits, out_hidden, (hipStream_t)stream)) return r;
    LAUNCH_CHECK();
    return FQ3_OK;
}

int fq3_codec_head_launch_(fq3_ctx* c, const void* hidden, void* out_logits, hipStream_t s) {
    GemvArgs g{};
    g.W = c->wt.codec_head; g.N = c->cfg.talker.vocab; g.K = c->cfg.talker.hidden; g.x = hidden; g.y = out_logits;
    return launch_gemv<PRO_PLAIN, EPI_STORE>(c, g, true, s);
}

// test / debugging hook: 0 = auto (MFMA prefill), 1 = force the token-by-token walk through the decode kernels
extern "C" int fq3_set_prefill_mode(fq3_ctx* c, int mode) { return fq3_set_option(c, "prefill_mode", mode); }

// -------------------------------------------------------------------------------------------------
// Sampler dispatch: single-wave kernels when top_p >= 1 (default), workgroup kernel otherwise
// -------------------------------------------------------------------------------------------------
__global__ void build_seen_kernel(const int64_t* history, int n, unsigned char* seen, int V) {
    for (int i = threadIdx.x; i < V; i += blockDim.x) seen[i] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) { const int id = (int)history[i]; if (id >= 0 && id < V) seen[id] = 1; }
}
template <typename F>
static void dispatch_nc(int V, F&& f) {
    if (V <= 2048) f(std::integral_constant<int, 1>{});
    else f(std::integral_constant<int, 2>{});
}

template <typename T>
static void launch_sample_pred(const DecodeState* st, const T* lg, int V, int cb, const SampleCfg& cfg, const T* nz,
                               int* codes, int G, int64_t* out64, const T* next_emb, T* next_in, int H, bool wave,
                               const TeacherForcing* tf, hipStream_t s) {
    if (wave) dispatch_nc(V, [&](auto nc) {
        constexpr int NC = decltype(nc)::value;
        hipLaunchKernelGGL((sample_pred_wave_kernel<T, NC>), dim3(1), dim3(256), 0, s, st, lg, V, cb, cfg, nz, codes, G,
                           out64, next_emb, next_in, H, tf);
    });
    else hipLaunchKernelGGL((sample_pred_kernel<T>), dim3(1), dim3(256), 0, s, st, lg, V, cb, cfg, nz, codes, G, out64,
                            next_emb, next_in, H, tf);
}
template <typename T>
static void launch_sample_talker(DecodeState* st, const T* lg, int V, const unsigned char* seen, int G, bool wave,
                                 const TeacherForcing* tf, hipStream_t s) {
    if (wave) dispatch_nc(V, [&](auto nc) {
        constexpr int NC = decltype(nc)::value;
        hipLaunchKernelGGL((sample_talker_wave_kernel<T, NC>), dim3(1), dim3(256), 0, s, st, lg, V, seen, G, tf);
    });
    else hipLaunchKernelGGL((sample_talker_kernel<T>), dim3(1), dim3(256), 0, s, st, lg, V, seen, G, tf);
}
template <typename T>
static void launch_sample_api(fq3_ctx* c, const T* lg, int V, const SampleCfg& cfg, const int64_t* history, int n_hist,
                              const T* noise, int64_t* out, hipStream_t s) {
    if (cfg.top_p >= 1.0f) {
        const unsigned char* seen = nullptr;
        if (n_hist > 0) {
            hipLaunchKernelGGL(build_seen_kernel, dim3(1), dim3(256), 0, s, history, n_hist, c->seen_api, V);
            seen = c->seen_api;
        }
        dispatch_nc(V, [&](auto nc) {
            constexpr int NC = decltype(nc)::value;
            hipLaunchKernelGGL((sample_api_wave_kernel<T, NC>), dim3(1), dim3(256), 0, s, lg, V, cfg, seen, noise, out);
        });
    } else {
        hipLaunchKernelGGL((sample_api_kernel<T>), dim3(1), dim3(256), 0, s, lg, V, cfg, history, n_hist, noise, out);
    }
}

// -------------------------------------------------------------------------------------------------
// Sampler
// -------------------------------------------------------------------------------------------------
static SampleCfg to_cfg(const fq3_sampling& s) {
    SampleCfg c{};
    c.temperature = s.temperature; c.top_k = s.top_k; c.top_p = s.top_p; c.do_sample = s.do_sample;
    c.rep_penalty = s.repetition_penalty; c.sup_lo = 0; c.sup_hi = 0; c.keep_id = -1; c.sup_extra = -1;
    return c;
}

extern "C" int fq3_sample(fq3_ctx* c, const void* logits, int V, const fq3_sampling* sp, const int64_t* history,
                          int n_hist, int sup_lo, int sup_hi, int keep_id, int suppress_eos, const void* noise,
                          int64_t* out_token, void* stream) {
    if (!c || !logits || !sp || !out_token) return fail(FQ3_EINVAL, "null argument");
    if (V <= 0 || V > kMaxVocab || V % 8) return fail(FQ3_EUNSUPPORTED, "vocab must be a multiple of 8 and at most 4096");
    if (sp->do_sample && !noise) return fail(FQ3_EINVAL, "do_sample needs a noise vector");
    if (sp->do_sample && !(sp->temperature > 0.f)) return fail(FQ3_EINVAL, "temperature must be > 0");
    SampleCfg cfg = to_cfg(*sp);
    cfg.sup_lo = sup_lo; cfg.sup_hi = sup_hi; cfg.keep_id = keep_id;
    cfg.sup_extra = suppress_eos ? c->cfg.codec_eos_token_id : -1;
    hipStream_t s = (hipStream_t)stream;
    if (c->cfg.dtype == FQ3_BF16)
        launch_sample_api<bf16_t>(c, (const bf16_t*)logits, V, cfg, history, history ? n_hist : 0, (const bf16_t*)noise, out_token, s);
    else
        launch_sample_api<float>(c, (const float*)logits, V, cfg, history, history ? n_hist : 0, (const float*)noise, out_token, s);
    LAUNCH_CHECK();
    return FQ3_OK;
}

// apply_repetition_penalty (sampling.py:10-29) as a standalone in-place operation on a logits row
template <typename T>
__global__ __launch_bounds__(256) void rep_penalty_kernel(T* logits, int V, const unsigned char* seen, float penalty) {
    for (int i = threadIdx.x; i < V; i += 256) {
        if (!seen[i]) continue;
        const float v = DT<T>::ld(logits + i);
        DT<T>::st(logits + i, v > 0.f ? v / penalty : v * penalty);
    }
}
extern "C" int fq3_apply_repetition_penalty(fq3_ctx* c, void* logits, int V, const int64_t* history, int n_hist,
                                            float penalty, void* stream) {
    if (!c || !logits) return fail(FQ3_EINVAL, "null argument");
    if (V <= 0 || V > kMaxVocab) return fail(FQ3_EUNSUPPORTED, "vocab must be at most 4096");
    if (penalty == 1.0f || n_hist <= 0 || !history) return FQ3_OK;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(build_seen_kernel, dim3(1), dim3(256), 0, s, history, n_hist, c->seen_api, V);
    if (c->cfg.dtype == FQ3_BF16) hipLaunchKernelGGL((rep_penalty_kernel<bf16_t>), dim3(1), dim3(256), 0, s, (bf16_t*)logits, V, c->seen_api, penalty);
    else hipLaunchKernelGGL((rep_penalty_kernel<float>), dim3(1), dim3(256), 0, s, (float*)logits, V, c->seen_api, penalty);
    LAUNCH_CHECK();
    return FQ3_OK;
}

extern "C" int fq3_set_predictor_sampling(fq3_ctx* c, const fq3_sampling* s) {
    if (!c || !s) return fail(FQ3_EINVAL, "null argument");
    if (s->do_sample && !(s->temperature > 0.f)) return fail(FQ3_EINVAL, "temperature must be > 0");
    if ((s->top_p >= 1.0f) != (c->pred_sampling.top_p >= 1.0f)) fq3_graph_reset(c);
    c->pred_sampling = *s;
    return FQ3_OK;
}

// -------------------------------------------------------------------------------------------------
// Predictor loop (predictor_graph.py:115-167).  `st` == nullptr: immediate mode (API call);
// otherwise sampling policy, noise row and output slot come from the on-device loop state.
// -------------------------------------------------------------------------------------------------
template <typename T>
static int predictor_passes_t(fq3_ctx* c, const DecodeState* st_dev, const void* pred_input, const void* noise_imm,
                              int64_t* out64, void* out_logits, hipStream_t s) {
    const fq3_stack_dims &p = c->cfg.predictor, &t = c->cfg.talker;
    const int G = c->cfg.num_code_groups, Vp = p.vocab, Ht = t.hidden;
    SampleCfg scfg = to_cfg(c->pred_sampling);
    // pass schedule: token A (past_hidden, pos 0), token B (embed(tok0), pos 1), then 14 single-token passes
    for (int pass = 0; pass < G; ++pass) {              // G = 16 token passes
        const void* hsrc = c->h;                        // where this pass leaves its last hidden state
        if (pass == 0 && c->opt_m2) {
            // two-token prefill in one pass over the weights
            const T* xa = (const T*)pred_input;
            const T* xb = (const T*)pred_input + Ht;
            if (c->cfg.has_projection) {
                GemvArgs g{};
                g.W = c->wt.proj_w; g.bias = c->wt.proj_b; g.N = p.hidden; g.K = Ht; g.x = xa; g.x2 = xb; g.y = c->pred_x; g.y2 = c->pred_x2;
                if (int r = launch_gemv2<PRO_PLAIN, EPI_STORE>(c, g, s)) return r;
                xa = (const T*)c->pred_x; xb = (const T*)c->pred_x2;
            }
            if (int r = run_predictor_pair(c, xa, xb, s)) return r;
            pass = 1;                                   // token B (slot 1) produced codebook 0 below
            hsrc = c->h2;
        } else {
        const T* x_talker = pass < 2 ? (const T*)pred_input + (size_t)pass * Ht : (const T*)c->pred_next;
        const void* x0 = x_talker;
        if (c->cfg.has_projection) {                    // small_to_mtp_projection (predictor_graph.py:118,145)
            GemvArgs g{};
            g.W = c->wt.proj_w; g.bias = c->wt.proj_b; g.N = p.hidden; g.K = Ht; g.x = x_talker; g.y = c->pred_x;
            if (int r = launch_gemv<PRO_PLAIN, EPI_STORE>(c, g, false, s)) return r;
            x0 = c->pred_x;
        }
        StepSrc src{x0, nullptr, pass};
        src.kv_only_tail = pass == 0;
        if (int r = run_stack(c, false, src, s)) return r;
        }
        if (pass == 0) continue;                        // first prefill token: only its K/V are needed
        const int cb = pass - 1;                        // codebook produced by this pass
        T* lg = out_logits ? (T*)out_logits + (size_t)cb * Vp : (T*)c->plogits + (size_t)cb * Vp;
        GemvArgs hgm{};
        hgm.eps = p.rms_eps; hgm.W = c->lmh[cb]; hgm.N = Vp; hgm.K = p.hidden; hgm.x = hsrc;
        hgm.norm_w = c->wt.predictor_final_norm; hgm.y = lg;
        if (int r = launch_gemv<PRO_NORM, EPI_STORE>(c, hgm, false, s)) return r;
        const T* nz = noise_imm ? (const T*)noise_imm + (size_t)cb * Vp : nullptr;
        const T* next_emb = cb + 1 < G - 1 ? (const T*)c->pemb[cb] : nullptr;     // codec_embeds[cb](tok) feeds pass cb+1
        launch_sample_pred<T>(st_dev, (const T*)lg, Vp, cb, scfg, nz, st_dev ? c->codes : nullptr, G, out64, next_emb,
                              (T*)c->pred_next, Ht, c->pred_sampling.top_p >= 1.0f, st_dev ? c->tf : nullptr, s);
    }
    return 0;
}

extern "C" int fq3_predictor_loop(fq3_ctx* c, const void* pred_input, const void* noise, int64_t* out_ids,
                                  void* out_logits, void* stream) {
    NEED_BOUND(c);
    if (!pred_input || !out_ids) return fail(FQ3_EINVAL, "null argument");
    if (c->pred_sampling.do_sample && !noise) return fail(FQ3_EINVAL, "predictor sampling enabled but no noise given");
    hipStream_t s = (hipStream_t)stream;
    int r = c->cfg.dtype == FQ3_BF16 ? predictor_passes_t<bf16_t>(c, nullptr, pred_input, noise, out_ids, out_logits, s)
                                     : predictor_passes_t<float>(c, nullptr, pred_input, noise, out_ids, out_logits, s);
    if (r) return r;
    LAUNCH_CHECK();
    return FQ3_OK;
}

// -------------------------------------------------------------------------------------------------
// Fused on-device decode loop
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void decode_arm_kernel(DecodeState* st, DecodeState init, unsigned char* seen, int seen_bytes,
                                                         uint32_t* past_hidden, const uint32_t* ph_src, int ph_words) {
    for (int i = threadIdx.x; i < seen_bytes / 4; i += 256) reinterpret_cast<uint32_t*>(seen)[i] = 0u;
    for (int i = threadIdx.x; i < ph_words; i += 256) past_hidden[i] = ph_src[i];
    if (threadIdx.x == 0) *st = init;
}
__global__ void decode_set_forced_kernel(TeacherForcing* tf, const int* forced, int* decisions) {
    tf->forced = forced; tf->decisions = decisions;
}

extern "C" int fq3_decode_begin(fq3_ctx* c, const fq3_decode_params* p, void* stream) {
    NEED_BOUND(c);
    if (!p || !p->past_hidden || !p->tts_pad_embed) return fail(FQ3_EINVAL, "null argument");
    if (p->trailing_len > 0 && !p->trailing_text) return fail(FQ3_EINVAL, "trailing_text");
    if (p->talker.do_sample && (!p->talker_noise || p->noise_frames <= 0)) return fail(FQ3_EINVAL, "talker sampling needs a noise ring");
    if (c->pred_sampling.do_sample && (!p->pred_noise || p->noise_frames <= 0)) return fail(FQ3_EINVAL, "predictor sampling needs a noise ring");
    if (p->talker.do_sample && !(p->talker.temperature > 0.f)) return fail(FQ3_EINVAL, "temperature must be > 0");
    if (p->max_new_tokens > c->cfg.max_frames) return fail(FQ3_EINVAL, "max_new_tokens exceeds the context's max_frames");
    if (p->prefill_len <= 0 || p->prefill_len > c->cfg.max_seq_len) return fail(FQ3_EINVAL, "prefill_len");
    if (p->first_token < 0 || p->first_token >= c->cfg.talker.vocab) return fail(FQ3_EINVAL, "first_token outside the codec vocabulary");
    if (p->gen_step < 0 || p->min_new_tokens < 0 || p->max_new_tokens < 0) return fail(FQ3_EINVAL, "negative counters");
    const bool wave = p->talker.top_p >= 1.0f;
    if (wave != c->talker_wave) { fq3_graph_reset(c); c->talker_wave = wave; }
    hipStream_t s = (hipStream_t)stream;
    // the blocks the loop can reach: frame f appends slot prefill_len + f, the loop stops at max_new_tokens or at max_seq_len - 1
    if (int r = fq3_kv_ensure_(c, p->prefill_len + p->max_new_tokens + 1, s)) return r;
    DecodeState h{};
    h.token = p->first_token; h.frame = 0; h.pos = p->prefill_len; h.gen_step = p->gen_step; h.done = 0;
    h.min_new = p->min_new_tokens; h.max_new = p->max_new_tokens; h.trailing_len = p->trailing_len;
    h.noise_frames = p->noise_frames > 0 ? p->noise_frames : 1;
    h.eos_id = c->cfg.codec_eos_token_id; h.max_seq = c->cfg.max_seq_len;
    h.n_pad = c->n_pad; h.rope_delta = c->rope_delta;
    h.sup_lo = c->cfg.talker.vocab - 1024 > 0 ? c->cfg.talker.vocab - 1024 : 0; h.sup_hi = c->cfg.talker.vocab;
    h.t_temperature = p->talker.temperature; h.t_top_k = p->talker.top_k; h.t_top_p = p->talker.top_p;
    h.t_do_sample = p->talker.do_sample; h.t_rep_penalty = p->talker.repetition_penalty;
    h.p_temperature = c->pred_sampling.temperature; h.p_top_k = c->pred_sampling.top_k; h.p_top_p = c->pred_sampling.top_p;
    h.p_do_sample = c->pred_sampling.do_sample;
    h.trailing_text = p->trailing_text; h.tts_pad = p->tts_pad_embed; h.talker_noise = p->talker_noise; h.pred_noise = p->pred_noise;
    // one launch arms everything; the state travels BY VALUE in the kernel arguments, so nothing on this stack frame
    // is read after the call returns and the host does not wait for the stream
    hipLaunchKernelGGL(decode_arm_kernel, dim3(1), dim3(256), 0, s, c->st, h, c->seen, (int)kMaxVocab,
                       (uint32_t*)c->past_hidden, (const uint32_t*)p->past_hidden, (int)((size_t)c->cfg.talker.hidden * c->esz / 4));
    LAUNCH_CHECK();
    return FQ3_OK;
}

__global__ void decode_cancel_kernel(DecodeState* st) { st->done = 1; }
extern "C" int fq3_decode_cancel(fq3_ctx* c, void* stream) {
    if (!c) return fail(FQ3_EINVAL, "null ctx");
    hipLaunchKernelGGL(decode_cancel_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, c->st);
    LAUNCH_CHECK();
    return FQ3_OK;
}

extern "C" int fq3_decode_set_forced(fq3_ctx* c, const int32_t* forced_codes, int32_t* decisions, void* stream) {
    NEED_BOUND(c);
    if (!c->tf) {
        // the first request for teacher forcing on this context: its samplers take the object's address from now on (a captured
        // graph -- this context's, and a lock-step batch's that was captured before -- still holds the null it was built with)
        if (!forced_codes && !decisions) return FQ3_OK;
        if (int r = dmalloc(c, (void**)&c->tf, sizeof(TeacherForcing))) return r;
        fq3_graph_reset(c);
    }
    hipLaunchKernelGGL(decode_set_forced_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, c->tf, forced_codes, decisions);
    LAUNCH_CHECK();
    return FQ3_OK;
}

template <typename T>
static int enqueue_frame_t(fq3_ctx* c, hipStream_t s) {
    const fq3_stack_dims& t = c->cfg.talker;
    const int G = c->cfg.num_code_groups, H = t.hidden;
    DecodeState* st = c->st;
    hipLaunchKernelGGL((frame_begin_kernel<T>), dim3(1), dim3(256), 0, s, st, (const T*)c->wt.codec_embedding,
                       (const T*)c->past_hidden, (T*)c->pred_in, c->codes, c->seen, H, G);
    if (int r = predictor_passes_t<T>(c, st, c->pred_in, nullptr, nullptr, nullptr, s)) return r;
    EmbTables tabs{};
    tabs.t[0] = c->wt.codec_embedding;
    for (int i = 1; i < G; ++i) tabs.t[i] = c->pemb[i - 1];
    if (G != 16) return fail(FQ3_EUNSUPPORTED, "the fused loop is built for 16 code groups");
    hipLaunchKernelGGL((embed_sum_kernel<T, 16>), dim3(1), dim3(256), 0, s, st, tabs, c->codes, (T*)c->xin, H,
                       c->wt.talker_cos, c->wt.talker_sin, c->wt.talker_rope_len, c->rope_delta, c->rope_now);
    StepSrc src{c->xin, &st->pos, 0};
    if (int r = run_stack(c, true, src, s)) return r;
    // final norm fused into the codec_head GEMV; block 0 also stores the normed hidden = next frame's past_hidden
    GemvArgs g{};
    g.eps = t.rms_eps; g.W = c->wt.codec_head; g.N = t.vocab; g.K = H; g.x = c->h;
    g.norm_w = c->wt.talker_final_norm; g.y = c->logits; g.xn_out = c->past_hidden;
    if (int r = launch_gemv<PRO_NORM, EPI_STORE>(c, g, true, s)) return r;
    launch_sample_talker<T>(st, (const T*)c->logits, t.vocab, c->seen, G, c->talker_wave, c->tf, s);
    return 0;
}
static int enqueue_frame(fq3_ctx* c, hipStream_t s) {
    return c->cfg.dtype == FQ3_BF16 ? enqueue_frame_t<bf16_t>(c, s) : enqueue_frame_t<float>(c, s);
}

extern "C" int fq3_graph_capture(fq3_ctx* c, void* stream) {
    NEED_BOUND(c);
    if (c->exec) return FQ3_OK;
    (void)stream;
    hipStream_t cs = c->cap_stream;
    // relaxed mode: other host threads (other utterances' contexts, the allocator) may issue HIP calls meanwhile
    HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
    int r = enqueue_frame(c, cs);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cs, &g);
    if (r) { if (g) (void)hipGraphDestroy(g); return r; }
    if (e != hipSuccess) return fail(FQ3_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    c->graph = g;
    HIPCHK(hipGraphInstantiate(&c->exec, c->graph, nullptr, nullptr, 0));
    return FQ3_OK;
}

extern "C" int fq3_decode_frames(fq3_ctx* c, int n_frames, void* stream) {
    NEED_BOUND(c);
    hipStream_t s = (hipStream_t)stream;
    for (int i = 0; i < n_frames; ++i) {
        if (c->exec) { HIPCHK(hipGraphLaunch(c->exec, s)); }
        else if (int r = enqueue_frame(c, s)) return r;
    }
    LAUNCH_CHECK();
    return FQ3_OK;
}

extern "C" int fq3_decode_poll(fq3_ctx* c, int* n_frames_total, int* done, void* stream) {
    if (!c) return fail(FQ3_EINVAL, "null ctx");
    hipStream_t s = (hipStream_t)stream;
    DecodeState h;
    HIPCHK(hipMemcpyAsync(&h, c->st, sizeof h, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (n_frames_total) *n_frames_total = h.frame;
    if (done) *done = h.done || h.token == h.eos_id;
    return FQ3_OK;
}

__global__ void codes_to_i64_kernel(const int* src, int64_t* dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

extern "C" int fq3_decode_codes(fq3_ctx* c, int from, int count, int64_t* out, void* stream) {
    if (!c || !out) return fail(FQ3_EINVAL, "null argument");
    if (from < 0 || count < 0 || from + count > c->cfg.max_frames) return fail(FQ3_EINVAL, "range");
    const int G = c->cfg.num_code_groups, n = count * G;
    if (n == 0) return FQ3_OK;
    hipLaunchKernelGGL(codes_to_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       c->codes + (size_t)from * G, out, n);
    LAUNCH_CHECK();
    return FQ3_OK;
}
