// 12 Hz RVQ codec decoder (vocoder) behind the C ABI: replaces upstream speech_tokenizer.decode as the
// reference calls it (faster_qwen3_tts/model.py:924,1093,1122).  Structure = transformers sibling
// Qwen3OmniMoeCode2Wav (modeling_qwen3_omni_moe.py:3636-3696) with the TTS tokenizer's split-RVQ
// front-end and pre_conv / input_proj / output_proj.
//
// Weight binding contract (names are checkpoint names under "decoder."; layouts are what the host
// packs once at load time, see fq3hip/codec.py):
//   dense conv   "<p>.conv.weight"  [Cout][k][Cin]           (from torch [Cout, Cin, k])
//   transp. conv "<p>.conv.weight"  [r*Cout + co][tap][Cin]  tap0 = kernel index r, tap1 = r + stride
//   depthwise    "<p>.dwconv.conv.weight" [C][7]
//   final conv   "decoder.decoder.<n>.conv.weight" [7][C]
//   RVQ proj     "<p>.output_proj.weight" [codebook_dim][rvq_dim]
//   q/k/v        "<p>.self_attn.qkv.weight" [3*QD][hidden]
//   gate/up      "<p>.mlp.gate_up.weight" [2*I][hidden], rows interleaved in groups of 16: gate[16b..16b+16), up[16b..16b+16)
//   rope tables  "rope.cos" / "rope.sin"  fp32 [max_frames][head_dim/2]
//
// A decode is planned as a list of row-range ops (see Plan below): the transformer runs over all T frames (its
// receptive field is the whole prefix), everything after it only over the rows the requested samples depend on.
#define FQ3_SKINNY_EXTERN           // skinny_gemm.cuh: the weight-stationary GEMM kernels are instantiated in fq3_prefill.hip only
#include "../../include/fq3hip.h"
#include "codec_kernels.cuh"

#include <algorithm>
#include <climits>
#include <cmath>
#include <functional>
#include <map>
#include <string>
#include <vector>

using namespace fq3;

extern "C" const char* fq3_last_error(void);
// share the error string with the decode TU through a tiny setter exported from fq3_api.hip
extern "C" void fq3_set_error_(const char* msg);
static int cfail(int code, const std::string& m) { fq3_set_error_(m.c_str()); return code; }
#define CHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return cfail(FQ3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct fq3_codec {
    fq3_codec_config cfg{};
    int esz = 2;
    std::map<std::string, const void*> w;
    std::map<std::string, int64_t> wn;
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_elems = 0;                         // elements each of the four workspaces holds: at first one utterance of max_frames, grown by a
                                                  // batched decode whose B x elems_for(T) exceeds it (never shrunk)
    void* snake_consts = nullptr;                 // per-channel SnakeBeta constants (a, ib), built at finalize
    std::map<std::string, std::pair<const void*, const void*>> snake;     // "<prefix>" -> (a, ib)
    bool ready = false;
    int glds_cap8 = 0;                            // measurement hook ("glds_cap8"): bf16 x 2 GEMMs take the eight-wave LDS-DMA tiles up to this many 128 x 64 tiles (0 = the launcher's default)
    int fuse_units = 1;                           // 1 (default): residual units of the 96-channel block as one launch each (resunit_kernel), 2: the
                                                  // 192-channel block too, 0: two GEMMs per unit.  Bit-identical; measured per 370-frame decode on
                                                  // MI355X: 7.92 (0) / 7.45 (1) / 7.93 ms (2) -- at 192 channels the 128 x C tile loses more against the
                                                  // 256 x 256 ring tile the k7 conv otherwise gets than the saved `mid` round trip brings
};

// State of the frame-level front end over a REFERENCE PREFIX (round 6; SURVEY section 7 step 9, the call sites model.py:1085-1115 and
// :919-937 re-decode `ref_codes + everything so far` for every phase-1 chunk and for every utterance of a voice): what rows >= ref_len
// of a later decode read from rows < ref_len --
//   pre   the RVQ projection sum of the last 2 rows (the k = 3 causal pre_conv's left context)
//   kv    per transformer layer the post-RoPE k | v rows of the last (sliding_window - 1) rows
//   out   the front end's output rows of the last kPrefixOutRows rows (the conv stack's halo reaches ~13 frames back)
// Every value is what a full decode computes for that row (the front end is causal and a row's arithmetic does not depend on the row
// count), so a decode that starts at row ref_len on top of this state is bit-identical to the full one.
constexpr int kPrefixOutRows = 48;
struct fq3_codec_prefix {
    fq3_codec* owner = nullptr;
    int ref_len = 0, n_pre = 0, n_kv = 0, n_out = 0;
    void *pre = nullptr, *kv = nullptr, *out = nullptr;       // [n_pre][codebook_dim], [n_layers][n_kv][2 QD], [n_out][latent_dim] in the codec's element type
};

static int64_t samples_for(const fq3_codec_config& c, int64_t T) {
    int64_t n = T;
    for (int i = 0; i < c.n_upsample; ++i) n *= c.upsampling_ratios[i];
    for (int i = 0; i < c.n_rates; ++i) n = (n - 1) * c.upsample_rates[i];     // k = 2r transposed conv trims r on both sides
    return n > 0 ? n : 0;
}

extern "C" int64_t fq3_codec_num_samples(const fq3_codec* c, int T) { return c ? samples_for(c->cfg, T) : -1; }

// elements of the largest activation of ONE utterance of T frames: walk the stages
static size_t elems_for(const fq3_codec_config& cfg, int64_t T) {
    int64_t rows = T, mx = T * std::max(std::max(cfg.latent_dim * 4, 3 * cfg.n_heads * cfg.head_dim), cfg.decoder_dim);
    mx = std::max(mx, 2 * T * (int64_t)cfg.rvq_dim);                             // the two RVQ sums share a workspace
    for (int i = 0; i < cfg.n_upsample; ++i) { rows *= cfg.upsampling_ratios[i]; mx = std::max(mx, rows * cfg.latent_dim * 4); }
    mx = std::max(mx, rows * cfg.decoder_dim);
    int ch = cfg.decoder_dim;
    for (int i = 0; i < cfg.n_rates; ++i) { rows = (rows - 1) * cfg.upsample_rates[i]; ch /= 2; mx = std::max(mx, rows * ch); }
    return (size_t)mx + 64;
}

extern "C" int fq3_codec_create(const fq3_codec_config* cfg, fq3_codec** out) {
    if (!cfg || !out) return cfail(FQ3_EINVAL, "null argument");
    if (cfg->dtype != FQ3_BF16 && cfg->dtype != FQ3_F32 && cfg->dtype != FQ3_BF16X2) return cfail(FQ3_EINVAL, "dtype");
    if (cfg->n_upsample < 0 || cfg->n_upsample > 4 || cfg->n_rates < 1 || cfg->n_rates > 8) return cfail(FQ3_EINVAL, "upsample lists");
    if (cfg->num_quantizers > 32) return cfail(FQ3_EUNSUPPORTED, "codec dims");
    if (cfg->head_dim != 32 && cfg->head_dim != 64 && cfg->head_dim != 128) return cfail(FQ3_EUNSUPPORTED, "codec head_dim must be 32, 64 or 128");
    if (cfg->sliding_window < 1 || cfg->sliding_window > 128) return cfail(FQ3_EUNSUPPORTED, "codec sliding_window must be in 1..128");
    auto m32 = [](int v) { return v % 32 == 0; };
    int ch = cfg->decoder_dim;
    bool ok = m32(cfg->rvq_dim) && m32(cfg->codebook_dim) && m32(cfg->latent_dim) && m32(cfg->hidden) && m32(cfg->inter) &&
              m32(cfg->n_heads * cfg->head_dim) && m32(ch);
    for (int i = 0; i < cfg->n_rates; ++i) { ch /= 2; ok = ok && m32(ch); }
    if (!ok) return cfail(FQ3_EUNSUPPORTED, "every channel count must be a multiple of 32 (MFMA K step)");
    fq3_codec* c = new fq3_codec();
    c->cfg = *cfg;
    c->esz = cfg->dtype == FQ3_BF16 ? 2 : 4;      // (FQ3_BF16X2: one 32-bit word per element, bf16 high part + bf16 residual)
    c->buf_elems = elems_for(*cfg, cfg->max_frames);
    for (int i = 0; i < 4; ++i) {
        hipError_t e = hipMalloc(&c->buf[i], c->buf_elems * c->esz);
        if (e != hipSuccess) { for (int j = 0; j < i; ++j) (void)hipFree(c->buf[j]); delete c; return cfail(FQ3_EHIP, "hipMalloc codec workspace"); }
    }
    *out = c;
    return FQ3_OK;
}

extern "C" int fq3_codec_destroy(fq3_codec* c) {
    if (!c) return FQ3_OK;
    (void)hipDeviceSynchronize();
    for (int i = 0; i < 4; ++i) if (c->buf[i]) (void)hipFree(c->buf[i]);
    if (c->snake_consts) (void)hipFree(c->snake_consts);
    delete c;
    return FQ3_OK;
}

extern "C" int fq3_codec_set_option(fq3_codec* c, const char* key, int value) {
    if (!c || !key) return cfail(FQ3_EINVAL, "null argument");
    if (std::string(key) == "glds_cap8") c->glds_cap8 = value;
    else if (std::string(key) == "fuse_units") c->fuse_units = value;      // 0: two GEMMs per residual unit; 1 (default): 96-channel units fused; 2: 192 too
    else return cfail(FQ3_EINVAL, std::string("unknown codec option: ") + key);
    return FQ3_OK;
}

extern "C" int fq3_codec_bind(fq3_codec* c, const char* name, const void* ptr, int64_t numel) {
    if (!c || !name || !ptr) return cfail(FQ3_EINVAL, "null argument");
    c->w[name] = ptr; c->wn[name] = numel; c->ready = false;
    return FQ3_OK;
}

static int need(fq3_codec* c, const std::string& n, int64_t numel, const void** out) {
    auto it = c->w.find(n);
    if (it == c->w.end()) return cfail(FQ3_ESTATE, "codec weight not bound: " + n);
    if (numel > 0 && c->wn[n] != numel) return cfail(FQ3_EINVAL, "codec weight has wrong size: " + n + " (" + std::to_string(c->wn[n]) + " vs " + std::to_string(numel) + ")");
    if (out) *out = it->second;
    return 0;
}

// every SnakeBeta of the decoder: (prefix of its alpha/beta tensors, channels)
static std::vector<std::pair<std::string, int>> snake_sites(const fq3_codec_config& g) {
    std::vector<std::pair<std::string, int>> v;
    int ch = g.decoder_dim;
    for (int i = 0; i < g.n_rates; ++i) {
        const std::string b = "decoder.decoder." + std::to_string(i + 1) + ".block.";
        v.push_back({b + "0.", ch});
        ch /= 2;
        for (int j = 2; j <= 4; ++j) { v.push_back({b + std::to_string(j) + ".act1.", ch}); v.push_back({b + std::to_string(j) + ".act2.", ch}); }
    }
    v.push_back({"decoder.decoder." + std::to_string(g.n_rates + 1) + ".", ch});
    return v;
}

extern "C" int fq3_codec_finalize(fq3_codec* c, void* stream) {
    if (!c) return cfail(FQ3_EINVAL, "null codec");
    const auto& g = c->cfg;
    const int QD = g.n_heads * g.head_dim;
    int r;
    // spot-check the weights whose shapes define the contract; every other name is checked on use
    for (int j = 0; j < g.num_quantizers; ++j) {
        const bool first = j < g.num_semantic;
        const std::string n = std::string("decoder.quantizer.") + (first ? "rvq_first" : "rvq_rest") + ".vq.layers." +
                              std::to_string(first ? j : j - g.num_semantic) + "._codebook.embedding";
        if ((r = need(c, n, (int64_t)g.codebook_size * g.rvq_dim, nullptr))) return r;
    }
    if ((r = need(c, "decoder.pre_conv.conv.weight", (int64_t)g.latent_dim * 3 * g.codebook_dim, nullptr))) return r;
    if ((r = need(c, "decoder.pre_transformer.layers.0.self_attn.qkv.weight", (int64_t)3 * QD * g.hidden, nullptr))) return r;
    if ((r = need(c, "decoder.pre_transformer.layers.0.mlp.gate_up.weight", (int64_t)2 * g.inter * g.hidden, nullptr))) return r;
    if (g.inter % 16) return cfail(FQ3_EUNSUPPORTED, "codec intermediate size must be a multiple of 16");
    if ((r = need(c, "rope.cos", (int64_t)g.max_frames * (g.head_dim / 2), nullptr))) return r;
    if ((r = need(c, "rope.sin", (int64_t)g.max_frames * (g.head_dim / 2), nullptr))) return r;
    // SnakeBeta constants: a = exp(alpha), ib = 1 / (exp(beta) + 1e-9), rounded like the Torch ops, once per bind
    const auto sites = snake_sites(g);
    size_t total = 0;
    for (auto& s : sites) total += 2 * (size_t)s.second;
    if (c->snake_consts) { (void)hipFree(c->snake_consts); c->snake_consts = nullptr; }
    CHIP(hipMalloc(&c->snake_consts, total * c->esz));
    size_t off = 0;
    hipStream_t st = (hipStream_t)stream;
    for (auto& s : sites) {
        const void *al = nullptr, *be = nullptr;
        if ((r = need(c, s.first + "alpha", s.second, &al)) || (r = need(c, s.first + "beta", s.second, &be))) return r;
        char* base = (char*)c->snake_consts + off * c->esz;
        void* pa = base; void* pib = base + (size_t)s.second * c->esz;
        if (g.dtype == FQ3_BF16) hipLaunchKernelGGL((snake_consts_kernel<bf16_t>), dim3((s.second + 255) / 256), dim3(256), 0, st, (const bf16_t*)al, (const bf16_t*)be, (bf16_t*)pa, (bf16_t*)pib, s.second);
        else if (g.dtype == FQ3_BF16X2) hipLaunchKernelGGL((snake_consts_kernel<bfs_t>), dim3((s.second + 255) / 256), dim3(256), 0, st, (const bfs_t*)al, (const bfs_t*)be, (bfs_t*)pa, (bfs_t*)pib, s.second);
        else hipLaunchKernelGGL((snake_consts_kernel<float>), dim3((s.second + 255) / 256), dim3(256), 0, st, (const float*)al, (const float*)be, (float*)pa, (float*)pib, s.second);
        c->snake[s.first] = {pa, pib};
        off += 2 * (size_t)s.second;
    }
    CHIP(hipStreamSynchronize(st));
    c->ready = true;
    return FQ3_OK;
}

namespace {
// ---- decode plan -----------------------------------------------------------------------------------------------
// Tensors are logical values (ids); an op produces `out` (and optionally `out2`) for rows [lo, rows) and reads its inputs
// at rows derived from lo.  need[] is propagated backwards from the requested first PCM sample.
enum { DEP_SAME = 0, DEP_BACK = 1, DEP_DIV = 2, DEP_ALL = 3 };
struct Dep { int t; int kind; int p; };
struct Op {
    int out = -1, out2 = -1;
    std::vector<Dep> in;
    int unit = 1;                        // rows of `out` per row of the launch (r for transposed convs)
    std::function<void(int lo)> run;     // lo in launch rows (= out rows / unit)
};
struct Plan {
    std::vector<Op> ops;
    int n_tensors = 0;
    int tensor() { return n_tensors++; }
    void add(Op&& o) { ops.push_back(std::move(o)); }
};
}  // namespace

template <typename T>
static GemmArgs lin(const void* A, int M, int Kc, const void* W, int N, const void* bias, void* Y) {
    GemmArgs a{}; a.A = A; a.lda = Kc; a.M = M; a.a_rows = M; a.n_taps = 1; a.tap_off[0] = 0; a.Cin = Kc; a.W = W; a.N = N;
    a.bias = bias; a.bias_mod = N; a.Y = Y; a.ldy = N; return a;
}
static GemmArgs conv(const void* A, int rows, int Cin, const void* W, int Cout, const void* bias, void* Y, int k, int dil) {
    GemmArgs a{}; a.A = A; a.lda = Cin; a.M = rows; a.a_rows = rows; a.n_taps = k; a.Cin = Cin; a.W = W; a.N = Cout;
    for (int i = 0; i < k; ++i) a.tap_off[i] = -(k - 1 - i) * dil;
    a.bias = bias; a.bias_mod = Cout; a.Y = Y; a.ldy = Cout; return a;
}

// every GEMM of a batched decode: NS problems of this shape, tensors compact per utterance
static GemmArgs segmented(GemmArgs a, int NS) {
    a.n_seg = NS; a.a_seg = (long)a.a_rows * a.lda; a.y_seg = (long)a.M * a.ldy; a.r_seg = (long)a.M * a.ldr;
    return a;
}

// NS utterances of Tn frames each in one set of launches (codes [NS][Tn][nq], pcm [NS][samples - first_sample]): every tensor
// is laid out [utterance][rows][C], every launch carries the utterance in a grid dimension, rows / taps / causal padding are
// local to an utterance.  Per utterance the arithmetic is that of a single decode (same tiles per row and column, same chains):
// bit-identical PCM.
// pfx (or null): NS prefix states of one ref_len -- the front end runs over rows [ref_len, Tn) only; when the requested samples reach
// further back than the cached output rows the call returns kPrefixTooShort and the caller takes the full path.
// cap (or null; NS = 1, front end only): the call fills this prefix state from a decode of the Tn = ref_len reference rows.
constexpr int kPrefixTooShort = -1000;
template <typename T>
static int decode_t(fq3_codec* c, const int64_t* codes, int NS, int Tn, int64_t first_sample, float* pcm, hipStream_t s,
                    const fq3_codec_prefix* const* pfx = nullptr, fq3_codec_prefix* cap = nullptr) {
    const auto& g = c->cfg;
    int err = 0;
    const int f0 = pfx ? pfx[0]->ref_len : 0;          // first row the front end computes
    PrefixSrc src_pre{}, src_kv{}, src_out{};
    if (pfx) for (int u = 0; u < NS; ++u) { src_pre.p[u] = pfx[u]->pre; src_kv.p[u] = pfx[u]->kv; src_out.p[u] = pfx[u]->out; }
    auto W = [&](const std::string& n) -> const void* { const void* p = nullptr; if (!err) err = need(c, n, 0, &p); return p; };
    auto SN = [&](const std::string& prefix) -> std::pair<const void*, const void*> {
        auto it = c->snake.find(prefix);
        if (it == c->snake.end()) { if (!err) err = cfail(FQ3_ESTATE, "snake constants missing: " + prefix); return {nullptr, nullptr}; }
        return it->second;
    };
    T* B0 = (T*)c->buf[0]; T* B1 = (T*)c->buf[1]; T* B2 = (T*)c->buf[2]; T* B3 = (T*)c->buf[3];
    const std::string D = "decoder.";
    auto el = [&](size_t n) { return dim3((unsigned)((n + 255) / 256), (unsigned)NS); };      // elementwise launches: utterance = blockIdx.y
    const dim3 rows4((Tn - f0 + 3) / 4, NS);                                                   // one wave per (new) row of the frame-level front end
    Plan P;
    auto gemm_op = [&](GemmArgs a, int out, int out2, std::vector<Dep> in, int unit = 1) {
        Op o; o.out = out; o.out2 = out2; o.in = std::move(in); o.unit = unit;
        a = segmented(a, NS);
        a.glds_cap8 = c->glds_cap8;
        o.run = [a, s](int lo) mutable { a.m_lo = lo; gemm_launch<T>(a, s); };
        P.add(std::move(o));
    };
    auto G = [&](GemmArgs a) { a.m_lo = f0; a.glds_cap8 = c->glds_cap8; gemm_launch<T>(segmented(a, NS), s); };             // (front end: rows [f0, Tn))
    // cached prefix rows -> every utterance's copy of a front-end tensor (rows [f0 - n, f0)); the reverse copy fills `cap`
    auto put_rows = [&](const PrefixSrc& src, size_t src_off, int src_ld, T* dst, int dst_ld, int dst_col0, int n_rows, int n_cols) {
        if (n_rows <= 0) return;
        constexpr int EPC = 16 / (int)sizeof(T);
        hipLaunchKernelGGL((prefix_rows_kernel<T>), dim3((n_rows * (n_cols / EPC) + 255) / 256, NS), dim3(256), 0, s, src, src_off, src_ld, dst,
                           (size_t)Tn * dst_ld, dst_ld, f0 - n_rows, dst_col0, n_rows, n_cols);
    };
    auto keep_rows = [&](void* dst, size_t dst_off, int dst_ld, const T* src, int src_ld, int src_col0, int n_rows, int n_cols) {
        if (n_rows <= 0) return;
        (void)hipMemcpy2DAsync((T*)dst + dst_off, (size_t)dst_ld * sizeof(T), src + (size_t)(Tn - n_rows) * src_ld + src_col0, (size_t)src_ld * sizeof(T),
                               (size_t)n_cols * sizeof(T), n_rows, hipMemcpyDeviceToDevice, s);
    };

    // ---- frame-level front end: RVQ, pre_conv, transformer (all T rows: the attention stack sees the whole prefix) ----
    RvqArgs ra{}; ra.nq = g.num_quantizers; ra.n_first = g.num_semantic; ra.dim = g.rvq_dim;
    for (int j = 0; j < g.num_quantizers; ++j) {
        const bool first = j < g.num_semantic;
        ra.books[j] = W(D + "quantizer." + (first ? "rvq_first" : "rvq_rest") + ".vq.layers." +
                        std::to_string(first ? j : j - g.num_semantic) + "._codebook.embedding");
    }
    if (err) return err;
    const int t_front = P.tensor();            // everything up to the transformer output is one "all rows" tensor
    {
        Op o; o.out = t_front;
        const void* w_first = W(D + "quantizer.rvq_first.output_proj.weight");
        const void* w_rest = W(D + "quantizer.rvq_rest.output_proj.weight");
        const void* w_pre = W(D + "pre_conv.conv.weight"); const void* b_pre = W(D + "pre_conv.conv.bias");
        const std::string TR = D + "pre_transformer.";
        const void* w_in = W(TR + "input_proj.weight"); const void* b_in = W(TR + "input_proj.bias");
        const int QD = g.n_heads * g.head_dim;
        const float* cosT = (const float*)W("rope.cos"); const float* sinT = (const float*)W("rope.sin");
        struct LayerW { const void *ln1, *qkv, *o, *ls1, *ln2, *gu, *down, *ls2; };
        std::vector<LayerW> LW(g.n_layers);
        for (int i = 0; i < g.n_layers; ++i) {
            const std::string L = TR + "layers." + std::to_string(i) + ".";
            LW[i] = {W(L + "input_layernorm.weight"), W(L + "self_attn.qkv.weight"), W(L + "self_attn.o_proj.weight"),
                     W(L + "self_attn_layer_scale.scale"), W(L + "post_attention_layernorm.weight"), W(L + "mlp.gate_up.weight"),
                     W(L + "mlp.down_proj.weight"), W(L + "mlp_layer_scale.scale")};
        }
        const void* w_norm = W(TR + "norm.weight");
        const void* w_out = W(TR + "output_proj.weight"); const void* b_out = W(TR + "output_proj.bias");
        if (err) return err;
        o.run = [=](int) {
            T* qf = B1; T* qr = B1 + (size_t)NS * Tn * g.rvq_dim;
            hipLaunchKernelGGL((rvq_gather_kernel<T>), dim3(Tn - f0, NS), dim3(256), 0, s, ra, codes, qf, qr, Tn, f0);
            G(lin<T>(qf, Tn, g.rvq_dim, w_first, g.codebook_dim, nullptr, B0));
            { GemmArgs a = lin<T>(qr, Tn, g.rvq_dim, w_rest, g.codebook_dim, nullptr, B0); a.res = B0; a.ldr = g.codebook_dim; G(a); }
            if (pfx) put_rows(src_pre, 0, g.codebook_dim, B0, g.codebook_dim, 0, pfx[0]->n_pre, g.codebook_dim);      // the k = 3 pre_conv's left context
            if (cap) keep_rows(cap->pre, 0, g.codebook_dim, B0, g.codebook_dim, 0, cap->n_pre, g.codebook_dim);
            G(conv(B0, Tn, g.codebook_dim, w_pre, g.latent_dim, b_pre, B1, 3, 1));
            G(lin<T>(B1, Tn, g.latent_dim, w_in, g.hidden, b_in, B0));                     // x = B0
            for (int i = 0; i < g.n_layers; ++i) {
                const LayerW& w = LW[i];
                hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), rows4, dim3(256), 0, s, (const T*)B0, (const T*)w.ln1, B1, f0, Tn, g.hidden, g.rms_eps);
                G(lin<T>(B1, Tn, g.hidden, w.qkv, 3 * QD, nullptr, B2));
                hipLaunchKernelGGL((rope_rows_kernel<T>), el((size_t)(Tn - f0) * 2 * g.n_heads * (g.head_dim / 2)), dim3(256), 0, s, B2, cosT, sinT, Tn, QD, g.head_dim, f0);
                // the layer's cached post-RoPE k | v of the last window - 1 prefix rows, in front of the new rows
                if (pfx) put_rows(src_kv, (size_t)i * pfx[0]->n_kv * 2 * QD, 2 * QD, B2, 3 * QD, QD, pfx[0]->n_kv, 2 * QD);
                if (cap) keep_rows(cap->kv, (size_t)i * cap->n_kv * 2 * QD, 2 * QD, B2, 3 * QD, QD, cap->n_kv, 2 * QD);
                const dim3 ag((Tn - f0 + 3) / 4, g.n_heads, NS);
                const float sc = 1.0f / sqrtf((float)g.head_dim);
                if (g.head_dim == 32) hipLaunchKernelGGL((swa_attn_kernel<T, 32>), ag, dim3(256), 0, s, (const T*)B2, B1, Tn, g.n_heads, g.sliding_window, sc, f0);
                else if (g.head_dim == 64) hipLaunchKernelGGL((swa_attn_kernel<T, 64>), ag, dim3(256), 0, s, (const T*)B2, B1, Tn, g.n_heads, g.sliding_window, sc, f0);
                else hipLaunchKernelGGL((swa_attn_kernel<T, 128>), ag, dim3(256), 0, s, (const T*)B2, B1, Tn, g.n_heads, g.sliding_window, sc, f0);
                { GemmArgs a = lin<T>(B1, Tn, QD, w.o, g.hidden, nullptr, B0); a.scale = w.ls1; a.res = B0; a.ldr = g.hidden; G(a); }
                hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), rows4, dim3(256), 0, s, (const T*)B0, (const T*)w.ln2, B1, f0, Tn, g.hidden, g.rms_eps);
                { GemmArgs a = lin<T>(B1, Tn, g.hidden, w.gu, 2 * g.inter, nullptr, B2); a.act = 2; a.ldy = g.inter; G(a); }     // SwiGLU epilogue
                { GemmArgs a = lin<T>(B2, Tn, g.inter, w.down, g.hidden, nullptr, B0); a.scale = w.ls2; a.res = B0; a.ldr = g.hidden; G(a); }
            }
            hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), rows4, dim3(256), 0, s, (const T*)B0, (const T*)w_norm, B1, f0, Tn, g.hidden, g.rms_eps);
            G(lin<T>(B1, Tn, g.hidden, w_out, g.latent_dim, b_out, B0));                    // h = B0 [T, latent]
            if (pfx) put_rows(src_out, 0, g.latent_dim, B0, g.latent_dim, 0, pfx[0]->n_out, g.latent_dim);           // the conv stack's halo rows
            if (cap) keep_rows(cap->out, 0, g.latent_dim, B0, g.latent_dim, 0, cap->n_out, g.latent_dim);
        };
        if (cap) { o.run(0); return err; }                     // (fq3_codec_prefix_create: the front end over the reference rows, nothing else)
        P.add(std::move(o));
    }
    // consumers of the front end's output: all of its rows exist in a full decode; behind a prefix state only rows >= ref_len - n_out do,
    // so there the true first row needed is propagated (and checked below)
    const int kFrontSame = pfx ? DEP_SAME : DEP_ALL, kFrontBack = pfx ? DEP_BACK : DEP_ALL;
    // ---- upsample: transposed conv (k = stride) + ConvNeXt ------------------------------------------------------------
    int rows = Tn;
    const int Lc = g.latent_dim;
    int t_h = t_front;                       // lives in B0
    for (int i = 0; i < g.n_upsample && !err; ++i) {
        const int f = g.upsampling_ratios[i];
        const std::string U = D + "upsample." + std::to_string(i) + ".";
        const int t_up = P.tensor(), t_dw = P.tensor(), t_ln = P.tensor(), t_pw = P.tensor(), t_o = P.tensor();
        { GemmArgs a = lin<T>(B0, rows, Lc, W(U + "0.conv.weight"), f * Lc, W(U + "0.conv.bias"), B1); a.bias_mod = Lc;
          gemm_op(a, t_up, -1, {{t_h, t_h == t_front ? kFrontSame : DEP_SAME, 0}}, f); }
        rows *= f;                                                                      // B1 = [rows, Lc]
        {
            const int R = rows;
            const T* dw = (const T*)W(U + "1.dwconv.conv.weight"); const T* db = (const T*)W(U + "1.dwconv.conv.bias");
            Op o; o.out = t_dw; o.in = {{t_up, DEP_BACK, 6}};
            o.run = [=](int lo) { if (lo < R) hipLaunchKernelGGL((dwconv7_kernel<T>), el((size_t)(R - lo) * Lc), dim3(256), 0, s, (const T*)B1, dw, db, B2, lo, R, Lc); };
            P.add(std::move(o));
            const T* nw = (const T*)W(U + "1.norm.weight"); const T* nb = (const T*)W(U + "1.norm.bias");
            Op o2; o2.out = t_ln; o2.in = {{t_dw, DEP_SAME, 0}};
            o2.run = [=](int lo) { if (lo < R) hipLaunchKernelGGL((layernorm_rows_kernel<T>), dim3((R - lo + 3) / 4, NS), dim3(256), 0, s, (const T*)B2, nw, nb, B3, lo, R, Lc, 1e-6f); };
            P.add(std::move(o2));
        }
        { GemmArgs a = lin<T>(B3, rows, Lc, W(U + "1.pwconv1.weight"), 4 * Lc, W(U + "1.pwconv1.bias"), B2); a.act = 1;
          gemm_op(a, t_pw, -1, {{t_ln, DEP_SAME, 0}}); }
        { GemmArgs a = lin<T>(B2, rows, 4 * Lc, W(U + "1.pwconv2.weight"), Lc, W(U + "1.pwconv2.bias"), B0);
          a.scale = W(U + "1.gamma"); a.res = B1; a.ldr = Lc;
          gemm_op(a, t_o, -1, {{t_pw, DEP_SAME, 0}, {t_up, DEP_SAME, 0}}); }             // h = B0 [rows, Lc]
        t_h = t_o;
    }
    // ---- decoder: every SnakeBeta is the epilogue of the GEMM that produces its input ---------------------------------
    const std::string DD = D + "decoder.";
    int ch = g.decoder_dim;
    T* bufH = B1; T* bufS = B2; T* bufM = B3; T* bufN = B0;       // raw h | snaked | mid (snaked conv1 out) | next raw
    int t_s = P.tensor();                                         // snaked input of the next GEMM
    {   // dec.0 (k7): only its SnakeBeta image (block 1's ".0" activation) is consumed
        auto sn = SN(DD + "1.block.0.");
        GemmArgs a = conv(B0, rows, Lc, W(DD + "0.conv.weight"), ch, W(DD + "0.conv.bias"), nullptr, 7, 1);
        a.sn_a = sn.first; a.sn_ib = sn.second; a.Y2 = bufS;
        gemm_op(a, -1, t_s, {{t_h, t_h == t_front ? kFrontBack : DEP_BACK, 6}});
    }
    for (int i = 0; i < g.n_rates && !err; ++i) {
        const int r = g.upsample_rates[i], co = ch / 2;
        const std::string Bk = DD + std::to_string(i + 1) + ".block.";
        int t_hraw = P.tensor(), t_s2 = P.tensor();
        {   // causal transposed conv k = 2r, stride r: out[m*r + q] = in[m+1] W[:,:,q] + in[m] W[:,:,q+r]
            auto sn = SN(Bk + "2.act1.");
            GemmArgs a{}; a.A = bufS; a.lda = ch; a.M = rows - 1; a.a_rows = rows; a.n_taps = 2; a.tap_off[0] = 1; a.tap_off[1] = 0; a.Cin = ch;
            a.W = W(Bk + "1.conv.weight"); a.N = r * co; a.bias = W(Bk + "1.conv.bias"); a.bias_mod = co; a.Y = bufH; a.ldy = r * co;
            a.sn_a = sn.first; a.sn_ib = sn.second; a.Y2 = bufM;
            gemm_op(a, t_hraw, t_s2, {{t_s, DEP_SAME, 0}}, r);
        }
        rows = (rows - 1) * r; ch = co;
        std::swap(bufS, bufM);                                    // bufS = snaked new h, bufM free
        t_s = t_s2;
        for (int j = 0; j < 3 && !err; ++j) {
            const std::string Un = Bk + std::to_string(j + 2) + ".";
            const int dil = j == 0 ? 1 : (j == 1 ? 3 : 9);
            // conv1 (k7, dilated): only snake_act2(conv1) is consumed; conv2 (1x1) + residual -> next raw h (unless nothing reads it)
            // and the next activation's image of it
            auto sn1 = SN(Un + "act2.");
            GemmArgs a1 = conv(bufS, rows, ch, W(Un + "conv1.conv.weight"), ch, W(Un + "conv1.conv.bias"), nullptr, 7, dil);
            a1.sn_a = sn1.first; a1.sn_ib = sn1.second; a1.Y2 = bufM;
            const bool last_unit = j == 2, last_block = i == g.n_rates - 1;
            const std::string next_sn = !last_unit ? Bk + std::to_string(j + 3) + ".act1."
                                        : (!last_block ? DD + std::to_string(i + 2) + ".block.0." : DD + std::to_string(g.n_rates + 1) + ".");
            auto sn = SN(next_sn);
            GemmArgs a = conv(bufM, rows, ch, W(Un + "conv2.conv.weight"), ch, W(Un + "conv2.conv.bias"), last_unit ? nullptr : (void*)bufN, 1, 1);
            a.res = bufH; a.ldr = ch; a.sn_a = sn.first; a.sn_ib = sn.second; a.Y2 = bufS;      // bufS (conv1's input) is dead by now
            const int t_hn = last_unit ? -1 : P.tensor(), t_sn = P.tensor();
            GemmArgs af = a; af.Y2 = bufM;
            if (c->fuse_units && (ch == 96 || c->fuse_units >= 2) && resunit_ok<T>(a1, af)) {
                // the two narrowest blocks: the whole unit in one launch (resunit_kernel), `mid` stays in LDS.  The new activation goes
                // to bufM (free now): other workgroups still read their halo rows of bufS while this one stores
                Op o; o.out = t_hn; o.out2 = t_sn; o.in = {{t_s, DEP_BACK, 6 * dil}, {t_hraw, DEP_SAME, 0}};
                a1 = segmented(a1, NS); af = segmented(af, NS);
                o.run = [a1, af, s](int lo) mutable { a1.m_lo = lo; (void)resunit_launch<T>(a1, af, s); };
                P.add(std::move(o));
                std::swap(bufS, bufM);
            } else {
                const int t_mid = P.tensor();
                gemm_op(a1, -1, t_mid, {{t_s, DEP_BACK, 6 * dil}});
                gemm_op(a, t_hn, t_sn, {{t_mid, DEP_SAME, 0}, {t_hraw, DEP_SAME, 0}});
            }
            if (!last_unit) { std::swap(bufH, bufN); t_hraw = t_hn; }
            t_s = t_sn;
        }
    }
    const int64_t n_out = samples_for(g, Tn);
    if (rows != (int)n_out) return cfail(FQ3_ESTATE, "internal: sample count mismatch");
    const int t_pcm = P.tensor();
    {
        const std::string F2 = DD + std::to_string(g.n_rates + 2) + ".";
        const T* fw = (const T*)W(F2 + "conv.weight"); const T* fb = (const T*)W(F2 + "conv.bias");
        const int R = rows, C = ch;
        const T* xin = bufS;
        Op o; o.out = t_pcm; o.in = {{t_s, DEP_BACK, 6}};
        size_t shm = 0;
        const int spw = final_conv_spw(C, (int)sizeof(T), &shm);
        if (spw == 0 || C % 8) return cfail(FQ3_EUNSUPPORTED, "output conv: channel count not supported (multiple of 8, window must fit the LDS)");
        o.run = [=](int lo) {
            if (lo >= R) return;
            auto kern = final_conv_kernel<T>;
            if (shm > 48 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            hipLaunchKernelGGL(kern, dim3((R - lo + spw - 1) / spw, NS), dim3(256), shm, s, xin, fw, fb, pcm, lo, R, C, spw);
        };
        P.add(std::move(o));
    }
    if (err) return err;
    if (ch > 1024) return cfail(FQ3_EUNSUPPORTED, "output conv wider than 1024 channels");
    // ---- backward pass: first row every tensor is needed from -------------------------------------------------------
    std::vector<long> need_from(P.n_tensors, LONG_MAX);
    need_from[t_pcm] = std::min<int64_t>(std::max<int64_t>(first_sample, 0), n_out);
    std::vector<int> lo_of(P.ops.size(), -1);
    for (int i = (int)P.ops.size() - 1; i >= 0; --i) {
        const Op& o = P.ops[i];
        long lo = LONG_MAX;
        if (o.out >= 0) lo = std::min(lo, need_from[o.out]);
        if (o.out2 >= 0) lo = std::min(lo, need_from[o.out2]);
        if (lo == LONG_MAX) continue;                          // nothing downstream reads this op
        const long lrow = std::max(0L, lo / o.unit);           // launch row (input row for transposed convs)
        lo_of[i] = (int)lrow;
        for (const Dep& d : o.in) {
            long v = 0;
            switch (d.kind) {
                case DEP_SAME: v = lrow; break;
                case DEP_BACK: v = lrow - d.p; break;
                case DEP_ALL: v = 0; break;
                default: v = lrow; break;
            }
            need_from[d.t] = std::min(need_from[d.t], std::max(0L, v));
        }
    }
    if (pfx && need_from[t_front] < (long)(f0 - pfx[0]->n_out)) return kPrefixTooShort;       // nothing was launched: the caller decodes in full
    for (size_t i = 0; i < P.ops.size(); ++i)
        if (lo_of[i] >= 0) P.ops[i].run(lo_of[i]);
    return 0;
}

// The four activation workspaces must hold n utterances of T frames, compact per tensor: n x elems_for(T) elements -- sized by the T of
// the CALL (round 5; it used to be n x the max_frames figure: ~24-48 GB pinned by the first 16-way vocode of 8-frame chunks).  The
// initial allocation (one utterance of max_frames) already covers every batch of short inputs; a growth (never a shrink) waits for the
// device and happens outside any graph capture.
static int reserve_batch(fq3_codec* c, int n, int T) {
    const size_t want = (size_t)n * elems_for(c->cfg, T);
    if (want <= c->buf_elems) return 0;
    CHIP(hipDeviceSynchronize());
    void* nb[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < 4; ++i)
        if (hipMalloc(&nb[i], want * c->esz) != hipSuccess) {
            for (int j = 0; j < i; ++j) (void)hipFree(nb[j]);
            (void)hipGetLastError();
            return cfail(FQ3_ENOMEM, "codec: workspace for " + std::to_string(n) + " utterances of " + std::to_string(T) + " frames could not be allocated");
        }
    for (int i = 0; i < 4; ++i) { (void)hipFree(c->buf[i]); c->buf[i] = nb[i]; }
    c->buf_elems = want;
    return 0;
}

static int decode_any(fq3_codec* c, const int64_t* codes, int B, int T, int64_t first_sample, float* pcm, void* stream,
                      const fq3_codec_prefix* const* pfx = nullptr) {
    if (!c || !codes || !pcm) return cfail(FQ3_EINVAL, "null argument");
    if (pfx) {
        if (B > 128) return cfail(FQ3_EINVAL, "codec decode behind prefix states: at most 128 utterances per call");
        for (int u = 0; u < B; ++u) {
            if (!pfx[u] || pfx[u]->owner != c) return cfail(FQ3_EINVAL, "prefix state is null or belongs to another codec");
            if (pfx[u]->ref_len != pfx[0]->ref_len) return cfail(FQ3_EINVAL, "the prefix states of one call must cover the same number of frames");
        }
        if (pfx[0]->ref_len >= T) return cfail(FQ3_EINVAL, "a decode behind a prefix state needs at least one frame after the prefix");
    }
    if (!c->ready) return cfail(FQ3_ESTATE, "codec weights not finalized");
    if (T < 1) return cfail(FQ3_EINVAL, "need at least 1 frame");
    if (B < 1 || B > 1024) return cfail(FQ3_EINVAL, "codec decode: batch size must be 1..1024");
    if (T > c->cfg.max_frames) return cfail(FQ3_ETOOLONG, "codec decode: " + std::to_string(T) + " frames exceed max_frames=" + std::to_string(c->cfg.max_frames));
    if (first_sample < 0 || first_sample > samples_for(c->cfg, T)) return cfail(FQ3_EINVAL, "first_sample outside the waveform");
    if (int r = reserve_batch(c, B, T)) return r;
    hipStream_t s = (hipStream_t)stream;
    int r;
    for (int pass = 0; pass < 2; ++pass) {
        switch (c->cfg.dtype) {
            case FQ3_BF16: r = decode_t<bf16_t>(c, codes, B, T, first_sample, pcm, s, pfx); break;
            case FQ3_BF16X2: r = decode_t<bfs_t>(c, codes, B, T, first_sample, pcm, s, pfx); break;
            default: r = decode_t<float>(c, codes, B, T, first_sample, pcm, s, pfx); break;
        }
        if (r != kPrefixTooShort) break;
        pfx = nullptr;                                         // the samples reach further back than the cached rows: the full decode
    }
    if (r) return r;
    CHIP(hipGetLastError());
    return FQ3_OK;
}

extern "C" int fq3_codec_decode(fq3_codec* c, const int64_t* codes, int T, float* pcm, void* stream) {
    return decode_any(c, codes, 1, T, 0, pcm, stream);
}

extern "C" int fq3_codec_decode_tail(fq3_codec* c, const int64_t* codes, int T, int64_t first_sample, float* pcm, void* stream) {
    return decode_any(c, codes, 1, T, first_sample, pcm, stream);
}

extern "C" int fq3_codec_decode_batch(fq3_codec* c, const int64_t* codes, int B, int T, int64_t first_sample, float* pcm, void* stream) {
    return decode_any(c, codes, B, T, first_sample, pcm, stream);
}

// ---- reference-prefix state (see fq3_codec_prefix above) ---------------------------------------------------------------------------
extern "C" int fq3_codec_prefix_destroy(fq3_codec_prefix* p) {
    if (!p) return FQ3_OK;
    (void)hipDeviceSynchronize();
    if (p->pre) (void)hipFree(p->pre);
    if (p->kv) (void)hipFree(p->kv);
    if (p->out) (void)hipFree(p->out);
    delete p;
    return FQ3_OK;
}

extern "C" int fq3_codec_prefix_create(fq3_codec* c, const int64_t* ref_codes, int ref_len, fq3_codec_prefix** out, void* stream) {
    if (!c || !ref_codes || !out) return cfail(FQ3_EINVAL, "null argument");
    if (!c->ready) return cfail(FQ3_ESTATE, "codec weights not finalized");
    if (ref_len < 1 || ref_len >= c->cfg.max_frames) return cfail(FQ3_EINVAL, "prefix length must be in 1 .. max_frames - 1");
    if (int r = reserve_batch(c, 1, ref_len)) return r;
    const auto& g = c->cfg;
    const int QD = g.n_heads * g.head_dim;
    fq3_codec_prefix* p = new fq3_codec_prefix();
    p->owner = c; p->ref_len = ref_len;
    p->n_pre = std::min(ref_len, 2); p->n_kv = std::min(ref_len, g.sliding_window - 1); p->n_out = std::min(ref_len, kPrefixOutRows);
    const size_t esz = c->esz;
    if (hipMalloc(&p->pre, std::max<size_t>(16, (size_t)p->n_pre * g.codebook_dim * esz)) != hipSuccess ||
        hipMalloc(&p->kv, std::max<size_t>(16, (size_t)g.n_layers * p->n_kv * 2 * QD * esz)) != hipSuccess ||
        hipMalloc(&p->out, std::max<size_t>(16, (size_t)p->n_out * g.latent_dim * esz)) != hipSuccess) {
        (void)hipGetLastError(); fq3_codec_prefix_destroy(p); return cfail(FQ3_ENOMEM, "codec prefix state: out of device memory");
    }
    hipStream_t s = (hipStream_t)stream;
    int r;
    switch (g.dtype) {
        case FQ3_BF16: r = decode_t<bf16_t>(c, ref_codes, 1, ref_len, 0, nullptr, s, nullptr, p); break;
        case FQ3_BF16X2: r = decode_t<bfs_t>(c, ref_codes, 1, ref_len, 0, nullptr, s, nullptr, p); break;
        default: r = decode_t<float>(c, ref_codes, 1, ref_len, 0, nullptr, s, nullptr, p); break;
    }
    if (r == FQ3_OK && hipGetLastError() != hipSuccess) r = cfail(FQ3_EHIP, "codec prefix state: a launch failed");
    if (r) { fq3_codec_prefix_destroy(p); return r; }
    *out = p;
    return FQ3_OK;
}

extern "C" int fq3_codec_prefix_frames(const fq3_codec_prefix* p) { return p ? p->ref_len : -1; }

extern "C" int fq3_codec_decode_batch_prefix(fq3_codec* c, const fq3_codec_prefix* const* prefixes, const int64_t* codes, int B, int T,
                                             int64_t first_sample, float* pcm, void* stream) {
    if (!prefixes) return cfail(FQ3_EINVAL, "null argument");
    return decode_any(c, codes, B, T, first_sample, pcm, stream, prefixes);
}
