// libfq3hip.so: batched decode -- B single-stream contexts ("lanes") advanced in lock-step by one launch chain.
// See batch_kernels.cuh for the design; the C ABI is declared in include/fq3hip.h (fq3_batch_*).
#define FQ3_SKINNY_EXTERN           // skinny_gemm.cuh: the weight-stationary GEMM kernels are instantiated in fq3_prefill.hip only
#include "fq3_ctx.h"
#include "batch_kernels.cuh"
#include "skinny_gemm.cuh"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

using namespace fq3;

#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fq3_fail_(FQ3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct fq3_batch {
    std::vector<fq3_ctx*> lanes;
    int B = 0;
    // activations, one row per lane
    void *h = nullptr, *xin = nullptr, *qkv = nullptr, *act = nullptr, *attn_out = nullptr, *logits = nullptr;
    void *pred_in = nullptr, *pred_x = nullptr, *pred_next = nullptr, *plogits = nullptr;
    float *part = nullptr, *rope_now = nullptr;
    size_t part_stride = 0;
    int Hm = 0, Im = 0, qkvm = 0;
    LaneTab tab{};                // host copies of the per-lane pointer tables ...
    LaneSt lst{};                 // (the loop states alone: batch poll)
    std::vector<LaneKV> tkv, pkv;
    LaneTabs ttab{};              // (the lanes' block tables, paged talker KV)
    LaneForced lf{};              // (teacher-forcing objects as of the last upload: null in product use)
    // ... and the device copies the kernels read (batch_kernels.cuh: 128 lanes of pointers do not fit a launch's 4 KB of arguments)
    LaneTab* d_tab = nullptr;
    LaneKV *d_tkv = nullptr, *d_pkv = nullptr;      // [n_layers]
    LaneTabs* d_ttab = nullptr;
    LaneForced* d_lf = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t cap_stream = nullptr;
    std::vector<void*> allocs;
    const float* h_ssq = nullptr; // set by run_stack_batch: the partials of what it left in `h` (null: none)
    float* ssq = nullptr;         // [B][Hm / 16]: sum-of-squares partials of the rows of `h` (written by the o_proj / down epilogues, read by the next normalising GEMM)
    int norm_fused = 0;           // a MEASURED NEGATIVE (round 5, profiles/r05_normfuse.txt), off: above norm_skinny_above lanes the RMSNorm of qkv / gate | up / lm heads
                                  // inside the weight-stationary GEMM (sum-of-squares partials from the residual GEMM's epilogue).  It removes the 216 normalisation
                                  // launches of a frame, but every one of a GEMM's 256 workgroups then normalises every token it stages: +0.3..1.2 us per GEMM at
                                  // hidden 1024, +3..10 us at 2048, against 1.9 us for the launch it saves inside a graph ("norm_fused" 1 selects it)
    int packed = 1;               // the weight-stationary GEMMs read the fragment-major copies of the layer matrices (round 6; bit-identical); 0 = the row-major matrices
    int pred_attn_group = 1;      // predictor attention: one wave per (kv group, lane), live rows only (round 6; bit-identical); 0 = one wave per (q head, lane), all 16 slots
    int rows2 = 1;                // 2: the activation buffers hold 2 B rows (the pair pass can engage), 1: B rows
    int pred_pair = 1;            // the predictor's two-token prefill as one pass over 2 B rows where that is bit-identical (see enqueue_batch_frame_t); 0 = two passes
    int attn_lane = 1;            // talker attention as one workgroup per (kv head, lane), final outputs, no merge launch: 0 never, 1 from attn_lane_from lanes (bf16), 2 always
    int attn_lane_from = 4 * kTokTile;
    int attn_lane_keys = 8;       // keys per load step of that kernel: 8 (two register sets of 2 K + 2 V rows per lane group: 122 VGPRs, four workgroups per CU -- measured 29.6 vs 30.6 us per launch at 128 lanes) or 16
    void* xn = nullptr;           // [B][Hm]: pre-normalised tokens of the weight-stationary form of the normalising GEMVs (above 32 lanes)
    int norm_skinny = 1;          // above norm_skinny_above lanes: qkv / gate | up / heads as rmsnorm_batch_kernel + skinny_gemm_kernel ("norm_skinny" 0: the panel kernels at every lane count)
    int norm_skinny_above = 2 * kTokTile;       // measured (profiles/r04_batch_norm_skinny.txt): the panel kernels win up to 32 lanes, the weight-stationary form from 48
    int norm_dual = 1;            // 17..32 lanes, hidden <= 1024: the normalising GEMVs prepare both token tiles before the first MFMA (bit-identical; "norm_dual" 0 = one panel, tile by tile)
    int use_skinny = 1;           // above 16 lanes the two residual GEMVs of a layer (o_proj, down) run on the weight-stationary prefill kernel
    int use_mfma = 0;             // bf16 GEMVs on the matrix cores: default ON for bf16 (ids verified against the oracle by teacher forcing
                                  // at full depth, 8 and 16 lanes); fq3_batch_set_option("mfma", 0) selects the VALU kernels, whose
                                  // lanes are bit-identical to the single-stream path
    // ---- lane groups (round 4; a measurement switch, OFF by default -- see auto_groups): the lanes split into 2..4 independent lock-step
    // chains, each a batch of its own ("kid") with its own activations and frame graph, advanced CONCURRENTLY on streams that were probed
    // for a hardware queue of their own.  Lanes are independent, so the groups meet only at the end of an fq3_batch_frames call (one
    // event each way).  A lane's values do not depend on its group (tests/test_gpu_batch.py::test_lane_groups_are_bit_identical).
    int groups_opt = 0;           // "groups": 0 = automatic (one chain, see auto_groups), 1..4 = as told
    bool is_kid = false;
    std::vector<fq3_batch*> kids;
    std::vector<hipStream_t> kid_streams;       // [g] for kid g >= 1 ([0] unused: kid 0 runs on the caller's stream)
    std::vector<hipStream_t> given_streams;     // fq3_batch_set_group_streams: the host's own (probed) streams, not owned
    hipStream_t probed_for = nullptr;           // the caller's stream the kid streams were probed against
    bool probed = false;
    hipEvent_t ev_fork = nullptr;
    std::vector<hipEvent_t> ev_join;
    // ---- batch poll (round 4): every lane's (frames done, finished) in ONE launch + ONE small copy, in stream order, into one of four
    // slots the host waits on later -- so a scheduler can queue the NEXT frames behind the poll and read the poll's result while they run
    // (64 x fq3_decode_poll = 64 copies + 64 stream synchronisations = ~2 ms of idle GPU per poll of a 64-lane batch)
    int* poll_dev = nullptr;                    // [kPollSlots][2 kMaxLanes]
    int* poll_host = nullptr;                   // pinned, same layout
    hipEvent_t poll_ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool poll_armed[4] = {false, false, false, false};
};
constexpr int kPollSlots = 4;

static int bmalloc(fq3_batch* b, void** p, size_t bytes) {
    HIPCHK(hipMalloc(p, bytes));
    HIPCHK(hipMemset(*p, 0, bytes));
    b->allocs.push_back(*p);
    return 0;
}

// the two-panel normalising GEMVs need up to ~83 KB of dynamic LDS: raise the limit of every instantiation once, outside any capture
template <int KS, int EPI>
static bool norm_dual_attr() {
    constexpr int NR = EPI == EPI_SWIGLU ? 2 : 1;
    const size_t shm2 = (((size_t)2 * kTokTile * (KS * 128 + 8) * 2 + 15) & ~(size_t)15) + (size_t)2 * 4 * NR * 256 * sizeof(float);
    if (shm2 <= 48 * 1024) return true;
    const bool r2 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_batch_mfma_norm_kernel<KS, EPI, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2) == hipSuccess;
    const bool r3 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_batch_mfma_norm_kernel<KS, EPI, 3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2) == hipSuccess;
    const bool r4 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_batch_mfma_norm_kernel<KS, EPI, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2) == hipSuccess;
    const bool r0 = hipFuncSetAttribute(reinterpret_cast<const void*>(gemv_batch_mfma_norm_kernel<KS, EPI, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm2) == hipSuccess;
    return r2 && r3 && r4 && r0;
}
static bool norm_dual_prepare() {
    static int done = -1;
    if (done < 0) {
        const bool r[] = {norm_dual_attr<2, EPI_STORE>(), norm_dual_attr<4, EPI_STORE>(), norm_dual_attr<8, EPI_STORE>(),
                          norm_dual_attr<2, EPI_SWIGLU>(), norm_dual_attr<4, EPI_SWIGLU>(), norm_dual_attr<8, EPI_SWIGLU>()};
        bool ok = true;
        for (bool b : r) ok = ok && b;
        done = ok ? 1 : 0;
    }
    return done == 1;
}

extern "C" int fq3_batch_graph_reset(fq3_batch* b) {
    if (!b) return fq3_fail_(FQ3_EINVAL, "null batch");
    if (b->exec) { (void)hipGraphExecDestroy(b->exec); b->exec = nullptr; }
    if (b->graph) { (void)hipGraphDestroy(b->graph); b->graph = nullptr; }
    for (fq3_batch* k : b->kids) fq3_batch_graph_reset(k);
    return FQ3_OK;
}

static void drop_group_streams(fq3_batch* b) {
    for (size_t g = 1; g < b->kid_streams.size(); ++g) {
        bool mine = true;
        for (hipStream_t h : b->given_streams) if (h == b->kid_streams[g]) mine = false;
        if (mine && b->kid_streams[g]) (void)hipStreamDestroy(b->kid_streams[g]);
    }
    b->kid_streams.clear();
    b->probed = false; b->probed_for = nullptr;
}
static void drop_kids(fq3_batch* b) {
    for (fq3_batch* k : b->kids) fq3_batch_destroy(k);
    b->kids.clear();
    drop_group_streams(b);
    for (hipEvent_t e : b->ev_join) if (e) (void)hipEventDestroy(e);
    b->ev_join.clear();
    if (b->ev_fork) { (void)hipEventDestroy(b->ev_fork); b->ev_fork = nullptr; }
}

extern "C" int fq3_batch_destroy(fq3_batch* b) {
    if (!b) return FQ3_OK;
    if (!b->is_kid) (void)hipDeviceSynchronize();
    fq3_batch_graph_reset(b);
    drop_kids(b);
    for (int i = 0; i < kPollSlots; ++i) if (b->poll_ev[i]) (void)hipEventDestroy(b->poll_ev[i]);
    if (b->poll_host) (void)hipHostFree(b->poll_host);
    if (b->cap_stream) (void)hipStreamDestroy(b->cap_stream);
    for (void* p : b->allocs) (void)hipFree(p);
    delete b;
    return FQ3_OK;
}

// One chain at every lane count.  Measured on MI355X (profiles/r04_batch_groups.txt): two concurrent chains LOSE -- 0.6B 64 lanes 5.57 ms
// per frame as one chain, 7.15 ms as 2 x 32; 32 lanes 4.36 vs 6.16 ms (2 x 16); 1.7B 64 lanes 7.09 vs 9.00 ms; three and four chains
// worse still -- only ~20 % better than running the chains one after another.  One chain's launches already occupy every CU with
// per-token work (normalisation VALU, token reads L2 -> CU, epilogues); a second chain does not find idle units, it queues behind the
// same ones, and a further token tile in the SAME launch (+0.6 ms per 16 lanes) shares the weight fragments, the launch and the ramp.
static int auto_groups(int B) { (void)B; return 1; }

static int batch_create_(fq3_ctx* const* lanes, int n_lanes, fq3_batch** out, bool is_kid);
static int poll_prepare(fq3_batch* b);
// a lane group runs the kernels the whole batch would run: the choice "above 16 lanes o_proj / down take the weight-stationary kernel"
// follows the BATCH's lane count, not the group's (a last group of a few lanes must not change its lanes' summation order)
static void sync_kid_options(fq3_batch* b) {
    for (fq3_batch* k : b->kids) {
        k->norm_dual = b->norm_dual; k->use_mfma = b->use_mfma; k->norm_skinny = b->norm_skinny; k->norm_fused = b->norm_fused; k->attn_lane = b->attn_lane; k->attn_lane_keys = b->attn_lane_keys; k->pred_pair = b->pred_pair; k->pred_attn_group = b->pred_attn_group; k->packed = b->packed;
        k->attn_lane_from = b->B >= b->attn_lane_from ? 0 : (1 << 30);      // the BATCH's lane count decides
        k->norm_skinny_above = b->B > b->norm_skinny_above ? 0 : (1 << 30);      // the BATCH's lane count decides, as for "skinny"
        k->use_skinny = (b->use_skinny == 1 && b->B > kTokTile) ? 2 : b->use_skinny;
    }
}
// (re)build the lane groups: `groups` chains of whole 16-lane token tiles, as even as the tile count allows
static int build_groups(fq3_batch* b, int groups) {
    drop_kids(b);
    const int tiles = (b->B + kTokTile - 1) / kTokTile;
    groups = std::max(1, std::min(groups, tiles));
    if (groups <= 1) return FQ3_OK;
    int l0 = 0;
    for (int g = 0; g < groups; ++g) {
        const int gt = tiles / groups + (g < tiles % groups ? 1 : 0);
        const int n = std::min(gt * kTokTile, b->B - l0);
        fq3_batch* k = nullptr;
        if (int r = batch_create_(b->lanes.data() + l0, n, &k, true)) { drop_kids(b); return r; }
        b->kids.push_back(k);
        l0 += n;
    }
    sync_kid_options(b);
    if (hipEventCreateWithFlags(&b->ev_fork, hipEventDisableTiming) != hipSuccess) { drop_kids(b); return fq3_fail_(FQ3_EHIP, "hipEventCreate"); }
    b->ev_join.assign(groups, nullptr);
    for (int g = 1; g < groups; ++g)
        if (hipEventCreateWithFlags(&b->ev_join[g], hipEventDisableTiming) != hipSuccess) { drop_kids(b); return fq3_fail_(FQ3_EHIP, "hipEventCreate"); }
    return FQ3_OK;
}

extern "C" int fq3_batch_create(fq3_ctx* const* lanes, int n_lanes, fq3_batch** out) {
    if (!lanes || !out) return fq3_fail_(FQ3_EINVAL, "null argument");
    fq3_batch* b = nullptr;
    if (int r = batch_create_(lanes, n_lanes, &b, false)) return r;
    if (b->use_mfma) if (int r = build_groups(b, auto_groups(b->B))) { fq3_batch_destroy(b); return r; }
    if (int r = poll_prepare(b)) { fq3_batch_destroy(b); return r; }        // the poll slots exist before the first frame: no allocation under a running batch
    *out = b;
    return FQ3_OK;
}

static int batch_create_(fq3_ctx* const* lanes, int n_lanes, fq3_batch** out, bool is_kid) {
    if (!lanes || !out) return fq3_fail_(FQ3_EINVAL, "null argument");
    if (n_lanes < 1 || n_lanes > kMaxLanes) return fq3_fail_(FQ3_EINVAL, "n_lanes must be 1..128");
    for (int i = 0; i < n_lanes; ++i) {
        if (!lanes[i] || !lanes[i]->bound) return fq3_fail_(FQ3_ESTATE, "every lane needs a context with bound weights");
        for (int j = 0; j < i; ++j) if (lanes[i] == lanes[j]) return fq3_fail_(FQ3_EINVAL, "a context can fill only one lane");
    }
    const fq3_ctx* c0 = lanes[0];
    for (int i = 1; i < n_lanes; ++i) {
        const fq3_ctx* c = lanes[i];
        const bool same_cfg = c->cfg.dtype == c0->cfg.dtype && c->cfg.num_code_groups == c0->cfg.num_code_groups &&
            c->cfg.has_projection == c0->cfg.has_projection && c->cfg.codec_eos_token_id == c0->cfg.codec_eos_token_id &&
            c->cfg.talker.hidden == c0->cfg.talker.hidden && c->cfg.talker.inter == c0->cfg.talker.inter &&
            c->cfg.talker.n_layers == c0->cfg.talker.n_layers && c->cfg.talker.n_heads == c0->cfg.talker.n_heads &&
            c->cfg.talker.n_kv_heads == c0->cfg.talker.n_kv_heads && c->cfg.talker.vocab == c0->cfg.talker.vocab &&
            c->cfg.predictor.hidden == c0->cfg.predictor.hidden && c->cfg.predictor.inter == c0->cfg.predictor.inter &&
            c->cfg.predictor.n_layers == c0->cfg.predictor.n_layers && c->cfg.predictor.vocab == c0->cfg.predictor.vocab &&
            c->tk.max_seq == c0->tk.max_seq && c->tk.workers == c0->tk.workers && c->tk.pool->blk_elems == c0->tk.pool->blk_elems;
        // one weight replica: every lane must have been bound to the same table
        const bool same_w = c->wt.codec_head == c0->wt.codec_head && c->wt.codec_embedding == c0->wt.codec_embedding &&
            c->tl[0].qkv == c0->tl[0].qkv && c->pl[0].qkv == c0->pl[0].qkv && c->wt.talker_cos == c0->wt.talker_cos;
        if (!same_cfg || !same_w) return fq3_fail_(FQ3_EINVAL, "lanes must share one config, one max_seq_len and one weight replica");
    }
    if (c0->cfg.num_code_groups != 16) return fq3_fail_(FQ3_EUNSUPPORTED, "the fused loop is built for 16 code groups");
    fq3_batch* b = new fq3_batch();
    b->is_kid = is_kid;
    b->lanes.assign(lanes, lanes + n_lanes);
    b->B = n_lanes;
    const fq3_stack_dims &t = c0->cfg.talker, &p = c0->cfg.predictor;
    const int esz = c0->esz, G = c0->cfg.num_code_groups, B = n_lanes;
    b->Hm = std::max(t.hidden, p.hidden); b->Im = std::max(t.inter, p.inter);
    b->qkvm = std::max(t.n_heads + 2 * t.n_kv_heads, p.n_heads + 2 * p.n_kv_heads) * kHeadDim;
    const int Vm = std::max(t.vocab, p.vocab);
    int r;
    auto A = [&](void** ptr, size_t n) { return bmalloc(b, ptr, n); };
    // h, xn, qkv, act, attn_out, pred_x, ssq hold 2 B rows where the predictor's two-token prefill can run as ONE pass over 2 B token rows
    // (bf16, more than norm_skinny_above lanes at the default setting, 2 B rows within the weight-stationary kernel's range -- see `pair` in
    // enqueue_batch_frame_t); smaller, fp32 and lane-group batches take one row per lane (round-5 advisor)
    const size_t R2 = (c0->cfg.dtype == FQ3_BF16 && B > 2 * kTokTile && 2 * B <= kSkinnyMaxRows) ? 2 : 1;
    b->rows2 = (int)R2;
    if ((r = A(&b->h, R2 * B * b->Hm * esz)) || (r = A(&b->xin, (size_t)B * b->Hm * esz)) ||
        (r = A(&b->xn, R2 * B * b->Hm * esz)) || (r = A((void**)&b->ssq, R2 * B * (b->Hm / 16 + 4) * sizeof(float))) || (r = A(&b->qkv, R2 * B * b->qkvm * esz)) || (r = A(&b->act, R2 * B * b->Im * esz)) ||
        (r = A(&b->attn_out, R2 * B * b->qkvm * esz)) || (r = A(&b->logits, (size_t)B * Vm * esz)) ||
        (r = A(&b->pred_in, (size_t)B * 2 * t.hidden * esz)) || (r = A(&b->pred_x, R2 * B * p.hidden * esz)) ||
        (r = A(&b->pred_next, (size_t)B * t.hidden * esz)) || (r = A(&b->plogits, (size_t)B * (G - 1) * p.vocab * esz))) {
        fq3_batch_destroy(b); return r;
    }
    b->part_stride = (size_t)std::max(t.n_kv_heads, p.n_kv_heads) * kMaxWorkers * 4 * kPartStride;
    if ((r = A((void**)&b->part, (size_t)B * b->part_stride * sizeof(float))) ||
        (r = A((void**)&b->rope_now, (size_t)B * kHeadDim * sizeof(float)))) { fq3_batch_destroy(b); return r; }
    for (int l = 0; l < B; ++l) {
        fq3_ctx* c = lanes[l];
        b->tab.st[l] = c->st; b->tab.codes[l] = c->codes; b->tab.seen[l] = c->seen; b->tab.past_hidden[l] = c->past_hidden;
        b->lst.st[l] = c->st;
    }
    b->tkv.resize(t.n_layers); b->pkv.resize(p.n_layers);
    for (int i = 0; i < t.n_layers; ++i) for (int l = 0; l < B; ++l) { b->tkv[i].k[l] = lanes[l]->tk.k[i]; b->tkv[i].v[l] = lanes[l]->tk.v[i]; }
    for (int l = 0; l < B; ++l) b->ttab.t[l] = lanes[l]->tk.d_table;
    b->ttab.blk_stride = (int)c0->tk.pool->blk_elems;
    for (int i = 0; i < p.n_layers; ++i) for (int l = 0; l < B; ++l) { b->pkv[i].k[l] = lanes[l]->pk.k[i]; b->pkv[i].v[l] = lanes[l]->pk.v[i]; }
    {   // the device copies (once: the pointers never change while the batch lives)
        auto up = [&](void** d, const void* h, size_t bytes) -> int {
            if (int rr = bmalloc(b, d, bytes)) return rr;
            HIPCHK(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
            return 0;
        };
        if ((r = up((void**)&b->d_tab, &b->tab, sizeof(LaneTab))) || (r = up((void**)&b->d_ttab, &b->ttab, sizeof(LaneTabs))) ||
            (r = up((void**)&b->d_tkv, b->tkv.data(), sizeof(LaneKV) * t.n_layers)) ||
            (r = up((void**)&b->d_pkv, b->pkv.data(), sizeof(LaneKV) * p.n_layers)) ||
            (r = up((void**)&b->d_lf, &b->lf, sizeof(LaneForced)))) { fq3_batch_destroy(b); return r; }
    }
    if (hipStreamCreateWithFlags(&b->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        fq3_batch_destroy(b); return fq3_fail_(FQ3_EHIP, "hipStreamCreateWithFlags");
    }
    b->use_mfma = c0->cfg.dtype == FQ3_BF16 ? 1 : 0;
    if (c0->cfg.dtype == FQ3_BF16 && !norm_dual_prepare()) {
        fq3_batch_destroy(b); return fq3_fail_(FQ3_EHIP, "batch: could not raise the LDS limit of the two-panel normalising GEMV kernels");
    }
    if (c0->cfg.dtype == FQ3_BF16 && !(skinny_prepare<SK_STORE>() && skinny_prepare<SK_SWIGLU>())) {
        fq3_batch_destroy(b); return fq3_fail_(FQ3_EHIP, "batch: could not raise the LDS limit of the weight-stationary GEMM kernels");
    }
    if (c0->cfg.dtype == FQ3_BF16 && !skinny_prepare<SK_RESIDUAL>()) {
        fq3_batch_destroy(b); return fq3_fail_(FQ3_EHIP, "batch: could not raise the LDS limit of the weight-stationary GEMV kernels");
    }
    *out = b;
    return FQ3_OK;
}

extern "C" int fq3_batch_set_option(fq3_batch* b, const char* key, int value) {
    if (!b || !key) return fq3_fail_(FQ3_EINVAL, "null argument");
    if (std::string(key) == "norm_dual") b->norm_dual = value;
    else if (std::string(key) == "norm_fused") b->norm_fused = value;     // RMSNorm inside the weight-stationary GEMM (default 0: a measured negative; 1 selects it)
    else if (std::string(key) == "pred_pair") b->pred_pair = value;
    else if (std::string(key) == "pred_attn_group") b->pred_attn_group = value;
    else if (std::string(key) == "packed_weights") b->packed = value;
    else if (std::string(key) == "attn_lane") b->attn_lane = value;
    else if (std::string(key) == "attn_lane_from") b->attn_lane_from = value;
    else if (std::string(key) == "attn_lane_keys") { if (value != 8 && value != 16) return fq3_fail_(FQ3_EINVAL, "attn_lane_keys must be 8 or 16"); b->attn_lane_keys = value; }
    else if (std::string(key) == "norm_skinny_above") b->norm_skinny_above = value;   // the lane count above which "norm_skinny" applies (default 32; measurement switch)
    else if (std::string(key) == "norm_skinny") b->norm_skinny = value;   // above 32 lanes: the normalising GEMVs as pre-normalise + weight-stationary GEMM (default 1)
    else if (std::string(key) == "skinny") b->use_skinny = value;  // o_proj / down of 17..32 lanes on skinny_gemm_kernel (default 1); 0: gemv_batch_mfma_plain_kernel;
                                                              // 2: at every lane count (a measurement switch: below 17 lanes the kernel's two-tile group is half empty)
    else if (std::string(key) == "mfma") b->use_mfma = value;      // bf16 GEMVs on the matrix cores (fp32 summation order differs from the single-stream kernels)
    else if (std::string(key) == "groups") {                       // lane groups on concurrent streams: 0 = automatic, 1 = one chain, 2..4
        if (value < 0 || value > 4) return fq3_fail_(FQ3_EINVAL, "groups must be 0..4");
        if (b->is_kid) return fq3_fail_(FQ3_EINVAL, "not an option of a lane group");
        (void)hipDeviceSynchronize();
        b->groups_opt = value;
        if (int r = build_groups(b, value ? value : (b->use_mfma ? auto_groups(b->B) : 1))) return r;
    }
    else return fq3_fail_(FQ3_EINVAL, std::string("unknown batch option: ") + key);
    sync_kid_options(b);
    return fq3_batch_graph_reset(b);
}

// The host's own side streams for the lane groups 1.. (e.g. probed against every other stream it keeps busy); without them the
// library probes streams of its own against the stream of the first fq3_batch_frames call.
extern "C" int fq3_batch_set_group_streams(fq3_batch* b, void* const* streams, int n) {
    if (!b || (n > 0 && !streams)) return fq3_fail_(FQ3_EINVAL, "null argument");
    if (n < 0 || n > 3) return fq3_fail_(FQ3_EINVAL, "at most three side streams");
    (void)hipDeviceSynchronize();
    drop_group_streams(b);
    b->given_streams.clear();
    for (int i = 0; i < n; ++i) b->given_streams.push_back((hipStream_t)streams[i]);
    return FQ3_OK;
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T, int PRO, int EPI>
static int launch_gemv_batch_t(BatchGemvArgs a, int esz, hipStream_t s) {
    const int need = (a.K + 511) / 512;
    const int grid = (a.N + 3) / 4;
    // tokens per LDS pass: as many as fit ~150 KB (16 lanes x K = 6144 x fp32 would need 393 KB), at most kGroupLanes
    int group = std::min(a.B, batch_group_max<T>(need));
    while (group > 1 && (size_t)group * a.K * esz > 150 * 1024) group = (group + 1) / 2;
    const size_t shm = (size_t)group * a.K * esz;
    if (shm > 150 * 1024) return fq3_fail_(FQ3_EUNSUPPORTED, "a single token of this inner dimension does not fit the 160 KB LDS");
    a.group = group;
    auto go = [&](auto nch) -> int {
        constexpr int NCH = decltype(nch)::value;
        auto kern = gemv_batch_kernel<T, NCH, PRO, EPI>;
        if (shm > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shm, s, a);
        return 0;
    };
    if (need <= 1) return go(std::integral_constant<int, 1>{});
    if (need <= 2) return go(std::integral_constant<int, 2>{});
    if (need <= 4) return go(std::integral_constant<int, 4>{});
    if constexpr (PRO == PRO_NORM) return fq3_fail_(FQ3_EUNSUPPORTED, "hidden size above 2048");      // a normalising GEMV reads K = hidden
    else {
        if (need <= 6) return go(std::integral_constant<int, 6>{});
        if (need <= 12) return go(std::integral_constant<int, 12>{});
        return fq3_fail_(FQ3_EUNSUPPORTED, "GEMV inner dimension above 6144");
    }
}
// matrix-core variants: bf16, built step counts; return -1000 when the shape is not covered (the VALU kernel takes over)
static thread_local int g_batch_norm_dual = 1;   // set per enqueue from fq3_batch::norm_dual
static thread_local int g_batch_norm_skinny = 1; // set per enqueue from fq3_batch::norm_skinny
static thread_local int g_batch_norm_skinny_above = 2 * kTokTile;
static thread_local int g_batch_packed = 1;      // set per enqueue from fq3_batch::packed (fragment-major weight copies, fq3_ctx.h)
template <int EPI>
static int launch_gemv_batch_mfma_norm(const BatchGemvArgs& a0, hipStream_t s) {
    if (a0.K % 128) return -1000;
    BatchGemvArgs a = a0;                                          // (a copy: the panel path below adds the fragment-major weight pointer)
    // above 32 lanes (three token tiles and more): normalise once, then the weight-stationary GEMM (batch_kernels.cuh::rmsnorm_batch_kernel)
    const int n_w = EPI == EPI_SWIGLU ? 2 * a.N : a.N;                  // weight rows: [gate | up] for SwiGLU
    if (g_batch_norm_skinny && a.xn_ws && a.B > g_batch_norm_skinny_above && !a.bias && skinny_k_ok(a.K) && a.K <= 2048 && n_w % 32 == 0 &&
        a.x_stride % 8 == 0 && a.y_stride % 4 == 0 && (EPI == EPI_STORE || EPI == EPI_SWIGLU)) {
        if (a.ssq_in && !a.xn_out && skinny_norm_ok(a.K, a.B)) {
            // the rows' sum-of-squares partials came with them (the residual GEMM that stored them): normalise inside the GEMM
            SkinnyArgs k{};
            k.X = reinterpret_cast<const bf16_t*>(a.x); k.ldx = a.x_stride; k.M = a.B; k.W = reinterpret_cast<const bf16_t*>(a.W); k.N = n_w;
            k.Y = reinterpret_cast<bf16_t*>(a.y); k.ldy = a.y_stride;
            k.ssq = a.ssq_in; k.gain = reinterpret_cast<const bf16_t*>(a.norm_w); k.eps = a.eps;
            if (g_batch_packed) k.Wp = reinterpret_cast<const bf16_t*>(fq3_packed_find_(a.W, EPI == EPI_SWIGLU ? 1 : 0));
            if constexpr (EPI == EPI_SWIGLU) skinny_launch<SK_SWIGLU>(k, a.K, s);
            else skinny_launch<SK_STORE>(k, a.K, s);
            return 0;
        }
        bf16_t* xn = reinterpret_cast<bf16_t*>(a.xn_ws);
        const dim3 grid((a.B + 3) / 4);
        if (a.K <= 1024) hipLaunchKernelGGL((rmsnorm_batch_kernel<2>), grid, dim3(256), 0, s, (const bf16_t*)a.x, a.x_stride, (const bf16_t*)a.norm_w, a.eps, a.K, a.B, xn, a.K, a.xn_out);
        else hipLaunchKernelGGL((rmsnorm_batch_kernel<4>), grid, dim3(256), 0, s, (const bf16_t*)a.x, a.x_stride, (const bf16_t*)a.norm_w, a.eps, a.K, a.B, xn, a.K, a.xn_out);
        SkinnyArgs k{};
        k.X = xn; k.ldx = a.K; k.M = a.B; k.W = reinterpret_cast<const bf16_t*>(a.W); k.N = n_w;
        k.Y = reinterpret_cast<bf16_t*>(a.y); k.ldy = a.y_stride;
        if (g_batch_packed) k.Wp = reinterpret_cast<const bf16_t*>(fq3_packed_find_(a.W, EPI == EPI_SWIGLU ? 1 : 0));
        if constexpr (EPI == EPI_SWIGLU) skinny_launch<SK_SWIGLU>(k, a.K, s);
        else skinny_launch<SK_STORE>(k, a.K, s);
        return 0;
    }
    const int grid = (a.N + 15) / 16;
    constexpr int NR = EPI == EPI_SWIGLU ? 2 : 1;
    const size_t shm = (((size_t)kTokTile * (a.K + 8) * 2 + 15) & ~(size_t)15) + (size_t)4 * NR * 256 * sizeof(float);
    const int nt = (a.B + kTokTile - 1) / kTokTile;            // token tiles: 1..4
    // the panel kernels' weight fragments from the fragment-major copy in 16-row blocks (kind 0), where the shape has whole blocks
    a.Wp = (g_batch_packed && a.N % 16 == 0 && (EPI != EPI_SWIGLU || a.up_off % 16 == 0)) ? fq3_packed_find_(a.W, 0) : nullptr;
    auto go = [&](auto ks) -> int {
        constexpr int KS = decltype(ks)::value;
        if constexpr (KS <= 8) {
            if (nt >= 2 && g_batch_norm_dual) {
                // two panels + a pair of tiles' partial sums (66 + 8..16 KB at K = 1024; the limit is raised in fq3_batch_create, never
                // inside a graph capture); three and four tiles repeat the scheme per pair
                const size_t shm2 = (((size_t)2 * kTokTile * (a.K + 8) * 2 + 15) & ~(size_t)15) + (size_t)2 * 4 * NR * 256 * sizeof(float);
                if (nt == 2) hipLaunchKernelGGL((gemv_batch_mfma_norm_kernel<KS, EPI, 2, true>), dim3(grid), dim3(256), shm2, s, a);
                else if (nt == 3) hipLaunchKernelGGL((gemv_batch_mfma_norm_kernel<KS, EPI, 3, true>), dim3(grid), dim3(256), shm2, s, a);
                else if (nt == 4) hipLaunchKernelGGL((gemv_batch_mfma_norm_kernel<KS, EPI, 4, true>), dim3(grid), dim3(256), shm2, s, a);
                else { BatchGemvArgs an = a; an.ntiles = nt; hipLaunchKernelGGL((gemv_batch_mfma_norm_kernel<KS, EPI, 0, true>), dim3(grid), dim3(256), shm2, s, an); }   // 65..128 lanes: rolled pair loop
                return 0;
            }
        }
        auto one = [&](auto ntc) -> int {
            constexpr int NTC = decltype(ntc)::value;
            auto kern = gemv_batch_mfma_norm_kernel<KS, EPI, NTC>;
            if (shm > 48 * 1024) HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            BatchGemvArgs an = a; an.ntiles = nt;
            hipLaunchKernelGGL(kern, dim3(grid), dim3(256), shm, s, an);
            return 0;
        };
        switch (nt) {
            case 1: return one(std::integral_constant<int, 1>{});
            case 2: return one(std::integral_constant<int, 2>{});
            case 3: return one(std::integral_constant<int, 3>{});
            case 4: return one(std::integral_constant<int, 4>{});
            default: return one(std::integral_constant<int, 0>{});          // 65..128 lanes: rolled tile loop
        }
    };
    switch (a.K / 128) {                      // hidden sizes: 256 (tests), 512, 1024 (0.6B, predictor), 2048 (1.7B)
        case 2: return go(std::integral_constant<int, 2>{});
        case 4: return go(std::integral_constant<int, 4>{});
        case 8: return go(std::integral_constant<int, 8>{});
        case 16: return go(std::integral_constant<int, 16>{});
        default: return -1000;
    }
}
static thread_local int g_batch_skinny = 1;      // set per enqueue from fq3_batch::use_skinny
// does this residual GEMV run on the weight-stationary kernel (whose epilogue can leave the rows' sum-of-squares partials)?
static bool residual_on_skinny(const BatchGemvArgs& a) {
    return g_batch_skinny && (a.B > kTokTile || g_batch_skinny >= 2) && !a.bias && skinny_k_ok(a.K) && a.N % 32 == 0 && a.x_stride % 8 == 0 &&
           a.y_stride % 4 == 0 && a.res_stride % 4 == 0;
}
template <int EPI>
static int launch_gemv_batch_mfma_plain(const BatchGemvArgs& a, hipStream_t s) {
    // 17..32 lanes, residual epilogue (o_proj, down: N = hidden only -- 64 or 128 workgroups of the one-row-block-per-workgroup GEMV, each
    // re-reading every lane's K-long token row): the prefill's weight-stationary kernel splits the two token tiles over workgroups and
    // K over 8 waves with per-wave LDS staging (skinny_gemm.cuh); measured 12.1 -> ~8 us (down) and 9.3 -> ~6.5 us (o_proj) per launch
    if constexpr (EPI == EPI_RESIDUAL) {
        if (residual_on_skinny(a)) {
            SkinnyArgs k{};
            k.ssq_out = a.ssq_out; k.ssq_ld = a.N / 16;
            k.X = reinterpret_cast<const bf16_t*>(a.x); k.ldx = a.x_stride; k.M = a.B; k.W = reinterpret_cast<const bf16_t*>(a.W); k.N = a.N;
            k.res = reinterpret_cast<const bf16_t*>(a.res); k.ldr = a.res_stride; k.Y = reinterpret_cast<bf16_t*>(a.y); k.ldy = a.y_stride;
            if (g_batch_packed) k.Wp = reinterpret_cast<const bf16_t*>(fq3_packed_find_(a.W, 0));
            skinny_launch<SK_RESIDUAL>(k, a.K, s);
            return 0;
        }
    }
    const int grid = (a.N + 15) / 16;
    auto go = [&](auto ks, auto nw) -> int {
        constexpr int KS = decltype(ks)::value, NW = decltype(nw)::value;
        BatchGemvArgs an = a;
        an.Wp = (g_batch_packed && a.N % 16 == 0) ? fq3_packed_find_(a.W, 0) : nullptr;      // fragment-major copy in 16-row blocks, where there is one
        an.ntiles = (a.B + kTokTile - 1) / kTokTile;         // token tiles
        switch (an.ntiles) {
            case 1: hipLaunchKernelGGL((gemv_batch_mfma_plain_kernel<KS, NW, EPI, 1>), dim3(grid), dim3(64 * NW), 0, s, an); break;
            case 2: hipLaunchKernelGGL((gemv_batch_mfma_plain_kernel<KS, NW, EPI, 2>), dim3(grid), dim3(64 * NW), 0, s, an); break;
            case 3: hipLaunchKernelGGL((gemv_batch_mfma_plain_kernel<KS, NW, EPI, 3>), dim3(grid), dim3(64 * NW), 0, s, an); break;
            case 4: hipLaunchKernelGGL((gemv_batch_mfma_plain_kernel<KS, NW, EPI, 4>), dim3(grid), dim3(64 * NW), 0, s, an); break;
            default: hipLaunchKernelGGL((gemv_batch_mfma_plain_kernel<KS, NW, EPI, 0>), dim3(grid), dim3(64 * NW), 0, s, an); break;   // 65..128 lanes
        }
        return 0;
    };
#define FQ3_PLAIN(KS, NW) return go(std::integral_constant<int, KS>{}, std::integral_constant<int, NW>{})
    switch (a.K) {                            // o_proj (q_dim), down (intermediate), small_to_mtp projection (hidden)
        case 256: FQ3_PLAIN(2, 4);            // tiny test config
        case 512: FQ3_PLAIN(4, 4);
        case 768: FQ3_PLAIN(6, 4);
        case 1024: FQ3_PLAIN(8, 4);
        case 2048: FQ3_PLAIN(8, 8);
        case 3072: FQ3_PLAIN(12, 8);
        case 4096: FQ3_PLAIN(16, 8);
        case 6144: FQ3_PLAIN(24, 8);
        default: return -1000;
    }
#undef FQ3_PLAIN
}
static thread_local int g_batch_mfma = 0;        // set per enqueue from fq3_batch::use_mfma
template <int PRO, int EPI>
static int launch_gemv_batch(const fq3_ctx* c, const BatchGemvArgs& a, hipStream_t s) {
    if (g_batch_mfma && c->cfg.dtype == FQ3_BF16) {
        int r;
        if constexpr (PRO == PRO_NORM) r = launch_gemv_batch_mfma_norm<EPI>(a, s);
        else r = launch_gemv_batch_mfma_plain<EPI>(a, s);
        if (r != -1000) return r;
    }
    return c->cfg.dtype == FQ3_BF16 ? launch_gemv_batch_t<bf16_t, PRO, EPI>(a, 2, s) : launch_gemv_batch_t<float, PRO, EPI>(a, 4, s);
}

struct BatchSrc { const void* x0; int x0_stride; int pos_imm; bool talker; bool kv_only_tail; bool pair = false; };
// `pair` (code predictor only): the stack runs over 2 B token rows -- row 2 l = lane l's token A (cache slot 0), row 2 l + 1 = its token B
// (slot 1) -- i.e. the two-token prefill of predictor_graph.py:121-128 as ONE pass over the weights, as the single stream does
// (run_predictor_pair).  Every row's arithmetic is that of the one-row-per-lane passes (the weight-stationary GEMMs and the row kernels
// do not depend on the row count), so the ids are those of the two-pass form bit for bit.

template <typename T>
static int run_stack_batch(fq3_batch* b, const BatchSrc& src, hipStream_t s) {
    fq3_ctx* c = b->lanes[0];
    const bool talker = src.talker;
    const fq3_stack_dims& d = talker ? c->cfg.talker : c->cfg.predictor;
    const std::vector<fq3_layer_weights>& L = talker ? c->tl : c->pl;
    const int rep = d.n_heads / d.n_kv_heads;
    const int B = src.pair ? 2 * b->B : b->B;            // token rows of this pass
    const int q_dim = d.n_heads * kHeadDim, kv_dim = d.n_kv_heads * kHeadDim;
    // RMSNorm inside the GEMM pair (the lane counts that take the weight-stationary form of the normalising GEMVs): the residual GEMMs
    // leave the sum-of-squares partials of the rows of `h`, the next normalising GEMM picks them up
    const bool fuse = b->norm_fused && g_batch_mfma && c->cfg.dtype == FQ3_BF16 && g_batch_norm_skinny && B > g_batch_norm_skinny_above &&
                      skinny_norm_ok(d.hidden, B) && b->Hm % 64 == 0;
    const float* h_ssq = nullptr;                      // partials of the current rows of b->h, or null
    b->h_ssq = nullptr;
    for (int i = 0; i < d.n_layers; ++i) {
        const fq3_layer_weights& w = L[i];
        const void* xin = i == 0 ? src.x0 : b->h;
        const int xin_stride = i == 0 ? src.x0_stride : b->Hm;
        BatchGemvArgs g{};
        g.ssq_in = i == 0 ? nullptr : h_ssq;
        g.B = B; g.eps = d.rms_eps; g.W = w.qkv; g.N = q_dim + 2 * kv_dim; g.K = d.hidden; g.x = xin; g.x_stride = xin_stride;
        g.norm_w = w.input_norm; g.y = b->qkv; g.y_stride = b->qkvm; g.xn_ws = b->xn;
        if (int r = launch_gemv_batch<PRO_NORM, EPI_STORE>(c, g, s)) return r;
        const bool tail_skip = src.kv_only_tail && i == d.n_layers - 1;
        AttnArgs a{};
        a.qkv = b->qkv; a.q_norm_w = w.q_norm; a.k_norm_w = w.k_norm; a.eps = d.rms_eps;
        a.n_kv = d.n_kv_heads; a.scale = 1.0f / sqrtf((float)kHeadDim); a.rep = rep;
        BatchGemvArgs o{};
        o.B = B; o.W = w.o; o.N = d.hidden; o.K = q_dim; o.y = b->h; o.y_stride = b->Hm; o.res = xin; o.res_stride = xin_stride;
        o.x_stride = b->qkvm;
        o.ssq_out = fuse && residual_on_skinny(o) ? b->ssq : nullptr;
        if (talker) {
            a.max_seq = c->tk.max_seq; a.part = b->part;
            const dim3 grid(d.n_kv_heads, c->tk.workers, B);
            const LaneKV* kvp = b->d_tkv + i;
            const LaneTabs* ttp = b->d_ttab; const LaneTab* tp = b->d_tab;
            // from 64 lanes (bf16, matrix-core path): one workgroup per (kv head, lane) that writes the final head outputs -- no partial
            // slots, no merge launch (batch_kernels.cuh::attn_decode_lane_kernel; "attn_lane": 0 = never, 1 = from attn_lane_from lanes, 2 = always)
            const bool lane_attn = b->attn_lane == 2 || (b->attn_lane == 1 && g_batch_mfma && c->cfg.dtype == FQ3_BF16 && B >= b->attn_lane_from);
            if (lane_attn) {
                a.out = b->attn_out;
                const dim3 lgrid(d.n_kv_heads, B);
                auto go = [&](auto ni) {
                    constexpr int NI = decltype(ni)::value;
                    if (rep == 1) hipLaunchKernelGGL((attn_decode_lane_kernel<T, 1, NI>), lgrid, dim3(256), 0, s, a, kvp, ttp, tp, b->qkvm, b->rope_now, b->qkvm);
                    else if (rep == 2) hipLaunchKernelGGL((attn_decode_lane_kernel<T, 2, NI>), lgrid, dim3(256), 0, s, a, kvp, ttp, tp, b->qkvm, b->rope_now, b->qkvm);
                    else hipLaunchKernelGGL((attn_decode_lane_kernel<T, 4, NI>), lgrid, dim3(256), 0, s, a, kvp, ttp, tp, b->qkvm, b->rope_now, b->qkvm);
                };
                if (b->attn_lane_keys == 8) go(std::integral_constant<int, 2>{}); else go(std::integral_constant<int, 4>{});
                o.x = b->attn_out; o.x_stride = b->qkvm;
                if (int r = launch_gemv_batch<PRO_PLAIN, EPI_RESIDUAL>(c, o, s)) return r;
            } else {
            if (rep == 1) hipLaunchKernelGGL((attn_decode_batch_kernel<T, 1>), grid, dim3(256), 0, s, a, kvp, ttp, tp, b->qkvm, b->rope_now, b->part_stride);
            else if (rep == 2) hipLaunchKernelGGL((attn_decode_batch_kernel<T, 2>), grid, dim3(256), 0, s, a, kvp, ttp, tp, b->qkvm, b->rope_now, b->part_stride);
            else hipLaunchKernelGGL((attn_decode_batch_kernel<T, 4>), grid, dim3(256), 0, s, a, kvp, ttp, tp, b->qkvm, b->rope_now, b->part_stride);
            hipLaunchKernelGGL((combine_batch_kernel<T>), dim3((q_dim / 8 + 255) / 256, B), dim3(256), 0, s, (const float*)b->part, b->part_stride,
                               c->tk.workers, rep, q_dim, (T*)b->attn_out, b->qkvm);
            o.x = b->attn_out; o.x_stride = b->qkvm;
            if (int r = launch_gemv_batch<PRO_PLAIN, EPI_RESIDUAL>(c, o, s)) return r;
            }
        } else {
            int rp = src.pos_imm; const int rl = c->wt.pred_rope_len;
            rp = rp < 0 ? 0 : (rp >= rl ? rl - 1 : rp);
            a.cos_row = c->wt.pred_cos + (size_t)rp * 64; a.sin_row = c->wt.pred_sin + (size_t)rp * 64;
            a.max_seq = c->pk.max_seq; a.pos_ptr = nullptr; a.pos_imm = src.pos_imm; a.n_pad = 0; a.out = b->attn_out;
            // one wave per (kv group, lane) serving the group's q heads from one read of the live K / V rows ("pred_attn_group", default 1;
            // bit-identical to the one-wave-per-q-head form, 0)
            auto pred_attn = [&](const AttnArgs& aa, int lanes, int stride) {
                const LaneKV* kvp = b->d_pkv + i;
                if (b->pred_attn_group && rep == 1) hipLaunchKernelGGL((attn_pred_group_batch_kernel<T, 1>), dim3(d.n_kv_heads, lanes), dim3(64), 0, s, aa, kvp, stride, stride);
                else if (b->pred_attn_group && rep == 2) hipLaunchKernelGGL((attn_pred_group_batch_kernel<T, 2>), dim3(d.n_kv_heads, lanes), dim3(64), 0, s, aa, kvp, stride, stride);
                else if (b->pred_attn_group && rep == 4) hipLaunchKernelGGL((attn_pred_group_batch_kernel<T, 4>), dim3(d.n_kv_heads, lanes), dim3(64), 0, s, aa, kvp, stride, stride);
                else hipLaunchKernelGGL((attn_pred_batch_kernel<T>), dim3(d.n_heads, lanes), dim3(64), 0, s, aa, kvp, stride, stride);
            };
            if (src.pair) {
                // token A (slot 0) of every lane, then token B (slot 1, which attends to A's row just appended): rows 2 l and 2 l + 1
                for (int m = 0; m < 2; ++m) {
                    AttnArgs am = a;
                    am.qkv = (const T*)b->qkv + (size_t)m * b->qkvm; am.out = (T*)b->attn_out + (size_t)m * b->qkvm;
                    am.pos_imm = m;
                    const int rm = m < c->wt.pred_rope_len ? m : c->wt.pred_rope_len - 1;
                    am.cos_row = c->wt.pred_cos + (size_t)rm * 64; am.sin_row = c->wt.pred_sin + (size_t)rm * 64;
                    pred_attn(am, b->B, 2 * b->qkvm);
                }
            } else pred_attn(a, B, b->qkvm);
            if (tail_skip) break;
            o.x = b->attn_out; o.x_stride = b->qkvm;
            if (int r = launch_gemv_batch<PRO_PLAIN, EPI_RESIDUAL>(c, o, s)) return r;
        }
        BatchGemvArgs m{};
        m.B = B; m.eps = d.rms_eps; m.W = w.gate_up; m.N = d.inter; m.K = d.hidden; m.x = b->h; m.x_stride = b->Hm;
        m.norm_w = w.post_norm; m.y = b->act; m.y_stride = b->Im; m.up_off = d.inter; m.xn_ws = b->xn;
        m.ssq_in = o.ssq_out;
        if (int r = launch_gemv_batch<PRO_NORM, EPI_SWIGLU>(c, m, s)) return r;
        BatchGemvArgs dn{};
        dn.B = B; dn.W = w.down; dn.N = d.hidden; dn.K = d.inter; dn.x = b->act; dn.x_stride = b->Im; dn.y = b->h; dn.y_stride = b->Hm;
        dn.res = b->h; dn.res_stride = b->Hm;
        dn.ssq_out = fuse && residual_on_skinny(dn) ? b->ssq : nullptr;
        if (int r = launch_gemv_batch<PRO_PLAIN, EPI_RESIDUAL>(c, dn, s)) return r;
        h_ssq = dn.ssq_out;
    }
    b->h_ssq = h_ssq;                                  // for the head that reads `h` next
    return 0;
}

template <typename T>
static int enqueue_batch_frame_t(fq3_batch* b, hipStream_t s) {
    fq3_ctx* c = b->lanes[0];
    const fq3_stack_dims &t = c->cfg.talker, &p = c->cfg.predictor;
    const int G = c->cfg.num_code_groups, H = t.hidden, Vp = p.vocab, B = b->B;
    const LaneTab* tab = b->d_tab;
    const LaneForced* lf = b->d_lf;                    // teacher-forcing objects (parity tests; uploaded by sync_forced): null for lanes that never asked
    hipLaunchKernelGGL((frame_begin_batch_kernel<T>), dim3(B), dim3(256), 0, s, tab, (const T*)c->wt.codec_embedding,
                       (T*)b->pred_in, H, G);
    // predictor: token A (past_hidden, slot 0), token B (embed(tok0), slot 1), then 14 single-token passes
    // the two-token prefill as one pass over 2 B rows ("pred_pair": default from the lane count at which every GEMM of the pass is the
    // weight-stationary kernel anyway -- above norm_skinny_above lanes, bf16, matrix-core path -- where a row's arithmetic does not
    // depend on the row count: bit-identical to the two passes; 0 = two passes at every lane count)
    const bool pair = b->pred_pair && b->rows2 == 2 && g_batch_mfma && c->cfg.dtype == FQ3_BF16 && g_batch_norm_skinny && g_batch_skinny && B > g_batch_norm_skinny_above &&
                      2 * B <= kSkinnyMaxRows && skinny_k_ok(p.hidden) && p.hidden <= 2048;
    for (int pass = 0; pass < G; ++pass) {
        const bool pp = pair && pass == 0;                 // this iteration runs token A and token B together
        const void* x_talker = pass < 2 ? (const void*)((const T*)b->pred_in + (size_t)pass * H) : (const void*)b->pred_next;
        const int x_stride = pp ? H : (pass < 2 ? 2 * H : H);       // (pair: pred_in [B][2 H] read as [2 B][H])
        const void* x0 = x_talker; int x0_stride = x_stride;
        if (c->cfg.has_projection) {                      // small_to_mtp_projection (predictor_graph.py:118,145)
            BatchGemvArgs g{};
            g.B = pp ? 2 * B : B; g.W = c->wt.proj_w; g.bias = c->wt.proj_b; g.N = p.hidden; g.K = H; g.x = x_talker; g.x_stride = x_stride;
            g.y = b->pred_x; g.y_stride = p.hidden;
            if (int r = launch_gemv_batch<PRO_PLAIN, EPI_STORE>(c, g, s)) return r;
            x0 = b->pred_x; x0_stride = p.hidden;
        }
        BatchSrc src{x0, x0_stride, pass, false, pass == 0 && !pp, pp};
        if (int r = run_stack_batch<T>(b, src, s)) return r;
        if (pass == 0 && !pp) continue;
        if (pp) pass = 1;                                  // token B (slot 1) produces codebook 0 below, from the odd rows of `h`
        const int cb = pass - 1;
        T* lg = (T*)b->plogits + (size_t)cb * Vp;
        const size_t lstride = (size_t)(G - 1) * Vp;
        BatchGemvArgs hg{};
        hg.B = B; hg.eps = p.rms_eps; hg.W = c->lmh[cb]; hg.N = Vp; hg.K = p.hidden;
        hg.x = pp ? (const void*)((const T*)b->h + b->Hm) : (const void*)b->h; hg.x_stride = pp ? 2 * b->Hm : b->Hm;
        hg.norm_w = c->wt.predictor_final_norm; hg.y = lg; hg.y_stride = (int)lstride; hg.xn_ws = b->xn;
        hg.ssq_in = pp ? nullptr : b->h_ssq;              // (the partials are indexed by packed row: the strided read of the pair pass normalises in its own launch)
        if (int r = launch_gemv_batch<PRO_NORM, EPI_STORE>(c, hg, s)) return r;
        const T* next_emb = cb + 1 < G - 1 ? (const T*)c->pemb[cb] : nullptr;
        if (Vp <= 2048) hipLaunchKernelGGL((sample_pred_batch_kernel<T, 1>), dim3(B), dim3(256), 0, s, tab, lf, (const T*)lg, lstride, Vp, cb, G, next_emb, (T*)b->pred_next, H);
        else hipLaunchKernelGGL((sample_pred_batch_kernel<T, 2>), dim3(B), dim3(256), 0, s, tab, lf, (const T*)lg, lstride, Vp, cb, G, next_emb, (T*)b->pred_next, H);
    }
    EmbTables tabs{};
    tabs.t[0] = c->wt.codec_embedding;
    for (int i = 1; i < G; ++i) tabs.t[i] = c->pemb[i - 1];
    hipLaunchKernelGGL((embed_sum_batch_kernel<T, 16>), dim3(B), dim3(256), 0, s, tab, tabs, (T*)b->xin, H, c->wt.talker_cos,
                       c->wt.talker_sin, c->wt.talker_rope_len, b->rope_now);
    BatchSrc src{b->xin, H, 0, true, false};
    if (int r = run_stack_batch<T>(b, src, s)) return r;
    BatchGemvArgs g{};
    g.B = B; g.eps = t.rms_eps; g.W = c->wt.codec_head; g.N = t.vocab; g.K = H; g.x = b->h; g.x_stride = b->Hm;
    g.norm_w = c->wt.talker_final_norm; g.y = b->logits; g.y_stride = t.vocab; g.xn_ws = b->xn;
    g.xn_out = b->d_tab->past_hidden;                  // (an address inside the device table: never dereferenced on the host)
    if (int r = launch_gemv_batch<PRO_NORM, EPI_STORE>(c, g, s)) return r;
    if (t.vocab <= 2048) hipLaunchKernelGGL((sample_talker_batch_kernel<T, 1>), dim3(B), dim3(256), 0, s, tab, lf, (const T*)b->logits, t.vocab, G);
    else hipLaunchKernelGGL((sample_talker_batch_kernel<T, 2>), dim3(B), dim3(256), 0, s, tab, lf, (const T*)b->logits, t.vocab, G);
    return 0;
}

static int check_lanes(fq3_batch* b) {
    for (fq3_ctx* c : b->lanes) {
        if (c->tk.max_seq > 0 && (c->cfg.talker.n_heads * kHeadDim > 2048))
            return fq3_fail_(FQ3_EUNSUPPORTED, "q_dim above 2048");
    }
    return 0;
}
// the lanes' teacher-forcing objects (parity tests; all null in product use) as the device table the samplers read: uploaded when it
// differs from the last upload -- outside any capture, behind whatever the batch still has in flight on `s`
static int sync_forced(fq3_batch* b, hipStream_t s) {
    LaneForced now{};
    bool same = true;
    for (int l = 0; l < b->B; ++l) { now.tf[l] = b->lanes[l]->tf; same = same && now.tf[l] == b->lf.tf[l]; }
    if (same) return 0;
    if (s) HIPCHK(hipStreamSynchronize(s));
    HIPCHK(hipMemcpy(b->d_lf, &now, sizeof now, hipMemcpyHostToDevice));
    b->lf = now;
    return 0;
}
static int enqueue_batch_frame(fq3_batch* b, hipStream_t s) {
    g_batch_mfma = b->use_mfma;
    g_batch_skinny = b->use_skinny;
    g_batch_norm_dual = b->norm_dual;
    g_batch_norm_skinny = b->norm_skinny;
    g_batch_norm_skinny_above = b->norm_skinny_above;
    g_batch_packed = b->packed;
    return b->lanes[0]->cfg.dtype == FQ3_BF16 ? enqueue_batch_frame_t<bf16_t>(b, s) : enqueue_batch_frame_t<float>(b, s);
}

extern "C" int fq3_batch_graph_capture(fq3_batch* b, void* stream) {
    if (!b) return fq3_fail_(FQ3_EINVAL, "null batch");
    if (!b->kids.empty()) {
        for (fq3_batch* k : b->kids) if (int r = fq3_batch_graph_capture(k, stream)) return r;
        return FQ3_OK;
    }
    if (b->exec) return FQ3_OK;
    if (int r = check_lanes(b)) return r;
    if (int r = sync_forced(b, (hipStream_t)stream)) return r;
    hipStream_t cs = b->cap_stream;
    HIPCHK(hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed));
    int r = enqueue_batch_frame(b, cs);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(cs, &g);
    if (r) { if (g) (void)hipGraphDestroy(g); return r; }
    if (e != hipSuccess) return fq3_fail_(FQ3_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
    b->graph = g;
    HIPCHK(hipGraphInstantiate(&b->exec, b->graph, nullptr, nullptr, 0));
    return FQ3_OK;
}

// ---- side streams of the lane groups: HIP multiplexes streams onto a few hardware queues (four by default) and two streams on one
// queue run in submission order, so a side stream is only worth having when it PROVABLY runs beside the caller's: a spin on the
// caller's stream (and on the side streams already chosen), an empty kernel on the candidate -- the candidate's kernel finishing
// while the spins still run is the proof.  ~2 ms per candidate, once per batch.
namespace {
__global__ void group_probe_spin_kernel(long long ticks) {      // wall_clock64: a constant 100 MHz counter
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
__global__ void group_probe_touch_kernel() {}
}
static bool runs_beside(hipStream_t cand, hipStream_t main, const std::vector<hipStream_t>& others) {
    hipEvent_t e_main = nullptr, e_cand = nullptr;
    if (hipEventCreateWithFlags(&e_main, hipEventDisableTiming) != hipSuccess) return false;
    if (hipEventCreateWithFlags(&e_cand, hipEventDisableTiming) != hipSuccess) { (void)hipEventDestroy(e_main); return false; }
    hipLaunchKernelGGL(group_probe_touch_kernel, dim3(1), dim3(64), 0, cand);          // code object resident, queue created
    (void)hipStreamSynchronize(cand);
    (void)hipStreamSynchronize(main);
    for (hipStream_t o : others) if (o) hipLaunchKernelGGL(group_probe_spin_kernel, dim3(1), dim3(64), 0, o, 200000LL);
    hipLaunchKernelGGL(group_probe_spin_kernel, dim3(1), dim3(64), 0, main, 200000LL);
    (void)hipEventRecord(e_main, main);
    hipLaunchKernelGGL(group_probe_touch_kernel, dim3(1), dim3(64), 0, cand);
    (void)hipEventRecord(e_cand, cand);
    (void)hipEventSynchronize(e_cand);
    const bool beside = hipEventQuery(e_main) == hipErrorNotReady;
    (void)hipStreamSynchronize(main);
    for (hipStream_t o : others) if (o) (void)hipStreamSynchronize(o);
    (void)hipEventDestroy(e_main); (void)hipEventDestroy(e_cand);
    (void)hipGetLastError();
    return beside;
}
// kid_streams for the caller's stream s (probed once; again when the caller changes streams); false = run the groups one after another
static bool group_streams_for(fq3_batch* b, hipStream_t s) {
    if (b->probed && b->probed_for == s) return b->kid_streams.size() == b->kids.size();
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return false; }
    drop_group_streams(b);
    b->probed = true; b->probed_for = s;
    std::vector<hipStream_t> chosen{nullptr};
    for (size_t g = 1; g < b->kids.size(); ++g) {
        hipStream_t got = nullptr;
        if (g - 1 < b->given_streams.size()) got = b->given_streams[g - 1];           // the host vouches for its own streams
        for (int t = 0; t < 8 && !got; ++t) {
            hipStream_t cand = nullptr;
            if (hipStreamCreateWithFlags(&cand, hipStreamNonBlocking) != hipSuccess) break;
            if (runs_beside(cand, s, std::vector<hipStream_t>(chosen.begin() + 1, chosen.end()))) got = cand;
            else (void)hipStreamDestroy(cand);
        }
        if (!got) break;
        chosen.push_back(got);
    }
    b->kid_streams = chosen;
    return b->kid_streams.size() == b->kids.size();
}

static int batch_frames_one(fq3_batch* b, int n_frames, hipStream_t s);
extern "C" int fq3_batch_frames(fq3_batch* b, int n_frames, void* stream) {
    if (!b) return fq3_fail_(FQ3_EINVAL, "null batch");
    hipStream_t s = (hipStream_t)stream;
    if (b->kids.empty()) return batch_frames_one(b, n_frames, s);
    if (n_frames <= 0) return FQ3_OK;
    if (!group_streams_for(b, s)) {                       // no stream with a queue of its own: the chains one after another
        for (fq3_batch* k : b->kids) if (int r = batch_frames_one(k, n_frames, s)) return r;
        return FQ3_OK;
    }
    HIPCHK(hipEventRecord(b->ev_fork, s));
    for (size_t g = 1; g < b->kids.size(); ++g) {
        HIPCHK(hipStreamWaitEvent(b->kid_streams[g], b->ev_fork, 0));
        if (int r = batch_frames_one(b->kids[g], n_frames, b->kid_streams[g])) return r;
        HIPCHK(hipEventRecord(b->ev_join[g], b->kid_streams[g]));
    }
    if (int r = batch_frames_one(b->kids[0], n_frames, s)) return r;
    for (size_t g = 1; g < b->kids.size(); ++g) HIPCHK(hipStreamWaitEvent(s, b->ev_join[g], 0));
    return FQ3_OK;
}

static int batch_frames_one(fq3_batch* b, int n_frames, hipStream_t s) {
    if (int r = check_lanes(b)) return r;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cap) != hipSuccess) { (void)hipGetLastError(); cap = hipStreamCaptureStatusNone; }
    if (cap == hipStreamCaptureStatusNone) if (int r = sync_forced(b, s)) return r;
    for (int i = 0; i < n_frames; ++i) {
        if (b->exec) { HIPCHK(hipGraphLaunch(b->exec, s)); }
        else if (int r = enqueue_batch_frame(b, s)) return r;
    }
    HIPCHK(hipGetLastError());
    return FQ3_OK;
}

extern "C" int fq3_batch_size(const fq3_batch* b) { return b ? b->B : 0; }

namespace {
__global__ void poll_gather_kernel(LaneSt t, int B, int* out) {
    const int l = threadIdx.x;
    if (l < B) {
        const DecodeState* st = t.st[l];
        out[2 * l] = st->frame;
        out[2 * l + 1] = (st->done || st->token == st->eos_id) ? 1 : 0;          // fq3_decode_poll's `done`
    }
}
}
static int poll_prepare(fq3_batch* b) {
    if (b->poll_host) return 0;
    const size_t bytes = (size_t)kPollSlots * 2 * kMaxLanes * sizeof(int);
    void* d = nullptr;
    if (int r = bmalloc(b, &d, bytes)) return r;
    b->poll_dev = (int*)d;
    HIPCHK(hipHostMalloc((void**)&b->poll_host, bytes, hipHostMallocDefault));
    for (int i = 0; i < kPollSlots; ++i) HIPCHK(hipEventCreateWithFlags(&b->poll_ev[i], hipEventDisableTiming));
    return 0;
}
extern "C" int fq3_batch_poll_async(fq3_batch* b, int slot, void* stream) {
    if (!b) return fq3_fail_(FQ3_EINVAL, "null batch");
    if (slot < 0 || slot >= kPollSlots) return fq3_fail_(FQ3_EINVAL, "poll slot must be 0..3");
    if (int r = poll_prepare(b)) return r;
    hipStream_t s = (hipStream_t)stream;
    int* d = b->poll_dev + (size_t)slot * 2 * kMaxLanes;
    hipLaunchKernelGGL(poll_gather_kernel, dim3(1), dim3(kMaxLanes), 0, s, b->lst, b->B, d);
    HIPCHK(hipMemcpyAsync(b->poll_host + (size_t)slot * 2 * kMaxLanes, d, (size_t)2 * b->B * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(hipEventRecord(b->poll_ev[slot], s));
    b->poll_armed[slot] = true;
    return FQ3_OK;
}
extern "C" int fq3_batch_poll_wait(fq3_batch* b, int slot, int* n_frames_total, int* done) {
    if (!b) return fq3_fail_(FQ3_EINVAL, "null batch");
    if (slot < 0 || slot >= kPollSlots || !b->poll_armed[slot]) return fq3_fail_(FQ3_ESTATE, "no poll was queued in this slot");
    HIPCHK(hipEventSynchronize(b->poll_ev[slot]));
    b->poll_armed[slot] = false;
    const int* h = b->poll_host + (size_t)slot * 2 * kMaxLanes;
    for (int l = 0; l < b->B; ++l) {
        if (n_frames_total) n_frames_total[l] = h[2 * l];
        if (done) done[l] = h[2 * l + 1];
    }
    return FQ3_OK;
}
