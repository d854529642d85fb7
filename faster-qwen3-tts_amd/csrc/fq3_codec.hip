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
//   q/k/v        "<p>.self_attn.qkv.weight" [3*QD][hidden];  gate/up "<p>.mlp.gate_up.weight" [2*I][hidden]
//   rope tables  "rope.cos" / "rope.sin"  fp32 [max_frames][head_dim/2]
#include "../../include/fq3hip.h"
#include "codec_kernels.cuh"

#include <cmath>
#include <map>
#include <string>
#include <vector>

using namespace fq3;

static thread_local std::string g_cerr;
extern "C" const char* fq3_last_error(void);
static int cfail(int code, const std::string& m);

struct fq3_codec {
    fq3_codec_config cfg{};
    int esz = 2;
    std::map<std::string, const void*> w;
    std::map<std::string, int64_t> wn;
    void* buf[4] = {nullptr, nullptr, nullptr, nullptr};
    size_t buf_elems = 0;
    bool ready = false;
};

// share the error string with the decode TU through a tiny setter exported from fq3_api.hip
extern "C" void fq3_set_error_(const char* msg);
static int cfail(int code, const std::string& m) { fq3_set_error_(m.c_str()); return code; }
#define CHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return cfail(FQ3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

static int64_t samples_for(const fq3_codec_config& c, int64_t T) {
    int64_t n = T;
    for (int i = 0; i < c.n_upsample; ++i) n *= c.upsampling_ratios[i];
    for (int i = 0; i < c.n_rates; ++i) n = (n - 1) * c.upsample_rates[i];     // k = 2r transposed conv trims r on both sides
    return n > 0 ? n : 0;
}

extern "C" int64_t fq3_codec_num_samples(const fq3_codec* c, int T) { return c ? samples_for(c->cfg, T) : -1; }

extern "C" int fq3_codec_create(const fq3_codec_config* cfg, fq3_codec** out) {
    if (!cfg || !out) return cfail(FQ3_EINVAL, "null argument");
    if (cfg->dtype != FQ3_BF16 && cfg->dtype != FQ3_F32) return cfail(FQ3_EINVAL, "dtype");
    if (cfg->n_upsample < 0 || cfg->n_upsample > 4 || cfg->n_rates < 1 || cfg->n_rates > 8) return cfail(FQ3_EINVAL, "upsample lists");
    if (cfg->num_quantizers > 32 || cfg->head_dim > 128 || cfg->head_dim % 2) return cfail(FQ3_EUNSUPPORTED, "codec dims");
    auto m32 = [](int v) { return v % 32 == 0; };
    int ch = cfg->decoder_dim;
    bool ok = m32(cfg->rvq_dim) && m32(cfg->codebook_dim) && m32(cfg->latent_dim) && m32(cfg->hidden) && m32(cfg->inter) &&
              m32(cfg->n_heads * cfg->head_dim) && m32(ch);
    for (int i = 0; i < cfg->n_rates; ++i) { ch /= 2; ok = ok && m32(ch); }
    if (!ok) return cfail(FQ3_EUNSUPPORTED, "every channel count must be a multiple of 32 (MFMA K step)");
    fq3_codec* c = new fq3_codec();
    c->cfg = *cfg;
    c->esz = cfg->dtype == FQ3_BF16 ? 2 : 4;
    // largest activation: walk the stages
    const int64_t T = cfg->max_frames;
    int64_t rows = T, mx = T * std::max(std::max(cfg->latent_dim * 4, 3 * cfg->n_heads * cfg->head_dim), cfg->decoder_dim);
    for (int i = 0; i < cfg->n_upsample; ++i) { rows *= cfg->upsampling_ratios[i]; mx = std::max(mx, rows * cfg->latent_dim * 4); }
    mx = std::max(mx, rows * cfg->decoder_dim);
    ch = cfg->decoder_dim;
    for (int i = 0; i < cfg->n_rates; ++i) { rows = (rows - 1) * cfg->upsample_rates[i]; ch /= 2; mx = std::max(mx, rows * ch); }
    c->buf_elems = (size_t)mx + 64;
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
    delete c;
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

extern "C" int fq3_codec_finalize(fq3_codec* c, void* /*stream*/) {
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
    if ((r = need(c, "rope.cos", (int64_t)g.max_frames * (g.head_dim / 2), nullptr))) return r;
    if ((r = need(c, "rope.sin", (int64_t)g.max_frames * (g.head_dim / 2), nullptr))) return r;
    c->ready = true;
    return FQ3_OK;
}

namespace {
struct Runner {
    fq3_codec* c; hipStream_t s; int err = 0;
    const void* W(const std::string& n) { const void* p = nullptr; if (!err) err = need(c, n, 0, &p); return p; }
    template <typename T> void gemm(GemmArgs a) {
        if (err) return;
        dim3 grid((a.N + 63) / 64, (a.M + 63) / 64);
        hipLaunchKernelGGL((conv_gemm_kernel<T>), grid, dim3(256), 0, s, a);
    }
};
template <typename T> static GemmArgs lin(const void* A, int M, int Kc, const void* W, int N, const void* bias, void* Y) {
    GemmArgs a{}; a.A = A; a.lda = Kc; a.M = M; a.a_rows = M; a.n_taps = 1; a.tap_off[0] = 0; a.Cin = Kc; a.W = W; a.N = N;
    a.bias = bias; a.bias_mod = N; a.Y = Y; a.ldy = N; return a;
}
static GemmArgs conv(const void* A, int rows, int Cin, const void* W, int Cout, const void* bias, void* Y, int k, int dil) {
    GemmArgs a{}; a.A = A; a.lda = Cin; a.M = rows; a.a_rows = rows; a.n_taps = k; a.Cin = Cin; a.W = W; a.N = Cout;
    for (int i = 0; i < k; ++i) a.tap_off[i] = -(k - 1 - i) * dil;
    a.bias = bias; a.bias_mod = Cout; a.Y = Y; a.ldy = Cout; return a;
}
}  // namespace

template <typename T>
static int decode_t(fq3_codec* c, const int64_t* codes, int Tn, float* pcm, hipStream_t s) {
    const auto& g = c->cfg;
    Runner R{c, s};
    T* B0 = (T*)c->buf[0]; T* B1 = (T*)c->buf[1]; T* B2 = (T*)c->buf[2]; T* B3 = (T*)c->buf[3];
    const std::string D = "decoder.";
    auto el = [&](size_t n) { return dim3((unsigned)((n + 255) / 256)); };
    // ---- RVQ -------------------------------------------------------------------------------------------
    RvqArgs ra{}; ra.nq = g.num_quantizers; ra.n_first = g.num_semantic; ra.dim = g.rvq_dim;
    for (int j = 0; j < g.num_quantizers; ++j) {
        const bool first = j < g.num_semantic;
        ra.books[j] = R.W(D + "quantizer." + (first ? "rvq_first" : "rvq_rest") + ".vq.layers." +
                          std::to_string(first ? j : j - g.num_semantic) + "._codebook.embedding");
    }
    if (R.err) return R.err;
    T* qf = B1; T* qr = B1 + (size_t)Tn * g.rvq_dim;
    hipLaunchKernelGGL((rvq_gather_kernel<T>), dim3(Tn), dim3(256), 0, s, ra, codes, qf, qr, Tn);
    R.gemm<T>(lin<T>(qf, Tn, g.rvq_dim, R.W(D + "quantizer.rvq_first.output_proj.weight"), g.codebook_dim, nullptr, B0));
    { GemmArgs a = lin<T>(qr, Tn, g.rvq_dim, R.W(D + "quantizer.rvq_rest.output_proj.weight"), g.codebook_dim, nullptr, B0);
      a.res = B0; a.ldr = g.codebook_dim; R.gemm<T>(a); }
    // ---- pre_conv (k=3) + transformer -------------------------------------------------------------------
    R.gemm<T>(conv(B0, Tn, g.codebook_dim, R.W(D + "pre_conv.conv.weight"), g.latent_dim, R.W(D + "pre_conv.conv.bias"), B1, 3, 1));
    const std::string TR = D + "pre_transformer.";
    R.gemm<T>(lin<T>(B1, Tn, g.latent_dim, R.W(TR + "input_proj.weight"), g.hidden, R.W(TR + "input_proj.bias"), B0));   // x = B0
    const int QD = g.n_heads * g.head_dim;
    const float* cosT = (const float*)R.W("rope.cos"); const float* sinT = (const float*)R.W("rope.sin");
    for (int i = 0; i < g.n_layers && !R.err; ++i) {
        const std::string L = TR + "layers." + std::to_string(i) + ".";
        hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), dim3((Tn + 3) / 4), dim3(256), 0, s, (const T*)B0, (const T*)R.W(L + "input_layernorm.weight"), B1, Tn, g.hidden, g.rms_eps);
        R.gemm<T>(lin<T>(B1, Tn, g.hidden, R.W(L + "self_attn.qkv.weight"), 3 * QD, nullptr, B2));
        hipLaunchKernelGGL((rope_rows_kernel<T>), el((size_t)Tn * 2 * g.n_heads * (g.head_dim / 2)), dim3(256), 0, s, B2, cosT, sinT, Tn, QD, g.head_dim);
        hipLaunchKernelGGL((swa_attn_kernel<T>), dim3((Tn + 3) / 4, g.n_heads), dim3(256), 0, s, (const T*)B2, B1, Tn, g.n_heads, g.head_dim, g.sliding_window, 1.0f / sqrtf((float)g.head_dim));
        { GemmArgs a = lin<T>(B1, Tn, QD, R.W(L + "self_attn.o_proj.weight"), g.hidden, nullptr, B0);
          a.scale = R.W(L + "self_attn_layer_scale.scale"); a.res = B0; a.ldr = g.hidden; R.gemm<T>(a); }
        hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), dim3((Tn + 3) / 4), dim3(256), 0, s, (const T*)B0, (const T*)R.W(L + "post_attention_layernorm.weight"), B1, Tn, g.hidden, g.rms_eps);
        R.gemm<T>(lin<T>(B1, Tn, g.hidden, R.W(L + "mlp.gate_up.weight"), 2 * g.inter, nullptr, B2));
        hipLaunchKernelGGL((silu_mul_kernel<T>), el((size_t)Tn * g.inter), dim3(256), 0, s, (const T*)B2, B1, Tn, g.inter);
        { GemmArgs a = lin<T>(B1, Tn, g.inter, R.W(L + "mlp.down_proj.weight"), g.hidden, nullptr, B0);
          a.scale = R.W(L + "mlp_layer_scale.scale"); a.res = B0; a.ldr = g.hidden; R.gemm<T>(a); }
    }
    hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), dim3((Tn + 3) / 4), dim3(256), 0, s, (const T*)B0, (const T*)R.W(TR + "norm.weight"), B1, Tn, g.hidden, g.rms_eps);
    R.gemm<T>(lin<T>(B1, Tn, g.hidden, R.W(TR + "output_proj.weight"), g.latent_dim, R.W(TR + "output_proj.bias"), B0));   // h = B0 [T, latent]
    // ---- upsample: transposed conv (k = s) + ConvNeXt ---------------------------------------------------
    int rows = Tn;
    const int Lc = g.latent_dim;
    for (int i = 0; i < g.n_upsample && !R.err; ++i) {
        const int f = g.upsampling_ratios[i];
        const std::string U = D + "upsample." + std::to_string(i) + ".";
        { GemmArgs a = lin<T>(B0, rows, Lc, R.W(U + "0.conv.weight"), f * Lc, R.W(U + "0.conv.bias"), B1); a.bias_mod = Lc; R.gemm<T>(a); }
        rows *= f;                                                                      // B1 = [rows, Lc]
        hipLaunchKernelGGL((dwconv7_kernel<T>), el((size_t)rows * Lc), dim3(256), 0, s, (const T*)B1, (const T*)R.W(U + "1.dwconv.conv.weight"), (const T*)R.W(U + "1.dwconv.conv.bias"), B2, rows, Lc);
        hipLaunchKernelGGL((layernorm_rows_kernel<T>), dim3((rows + 3) / 4), dim3(256), 0, s, (const T*)B2, (const T*)R.W(U + "1.norm.weight"), (const T*)R.W(U + "1.norm.bias"), B3, rows, Lc, 1e-6f);
        { GemmArgs a = lin<T>(B3, rows, Lc, R.W(U + "1.pwconv1.weight"), 4 * Lc, R.W(U + "1.pwconv1.bias"), B2); a.act = 1; R.gemm<T>(a); }
        { GemmArgs a = lin<T>(B2, rows, 4 * Lc, R.W(U + "1.pwconv2.weight"), Lc, R.W(U + "1.pwconv2.bias"), B0);
          a.scale = R.W(U + "1.gamma"); a.res = B1; a.ldr = Lc; R.gemm<T>(a); }            // h = B0 [rows, Lc]
    }
    // ---- decoder ----------------------------------------------------------------------------------------
    const std::string DD = D + "decoder.";
    int ch = g.decoder_dim;
    R.gemm<T>(conv(B0, rows, Lc, R.W(DD + "0.conv.weight"), ch, R.W(DD + "0.conv.bias"), B1, 7, 1));      // y = B1 [rows, ch]
    T* y = B1; T* t0 = B0; T* t1 = B2; T* t2 = B3;
    for (int i = 0; i < g.n_rates && !R.err; ++i) {
        const int r = g.upsample_rates[i], co = ch / 2;
        const std::string Bk = DD + std::to_string(i + 1) + ".block.";
        hipLaunchKernelGGL((snake_kernel<T>), el((size_t)rows * ch), dim3(256), 0, s, (const T*)y, (const T*)R.W(Bk + "0.alpha"), (const T*)R.W(Bk + "0.beta"), t0, (size_t)rows * ch, ch);
        {   // causal transposed conv k = 2r, stride r: out[m*r + q] = in[m+1] W[:,:,q] + in[m] W[:,:,q+r]
            GemmArgs a{}; a.A = t0; a.lda = ch; a.M = rows - 1; a.a_rows = rows; a.n_taps = 2; a.tap_off[0] = 1; a.tap_off[1] = 0; a.Cin = ch;
            a.W = R.W(Bk + "1.conv.weight"); a.N = r * co; a.bias = R.W(Bk + "1.conv.bias"); a.bias_mod = co; a.Y = t1; a.ldy = r * co;
            R.gemm<T>(a);
        }
        rows = (rows - 1) * r; ch = co;
        std::swap(y, t1);                                                                  // y = [rows, ch]
        for (int j = 0; j < 3 && !R.err; ++j) {
            const std::string Un = Bk + std::to_string(j + 2) + ".";
            const int dil = j == 0 ? 1 : (j == 1 ? 3 : 9);
            hipLaunchKernelGGL((snake_kernel<T>), el((size_t)rows * ch), dim3(256), 0, s, (const T*)y, (const T*)R.W(Un + "act1.alpha"), (const T*)R.W(Un + "act1.beta"), t0, (size_t)rows * ch, ch);
            R.gemm<T>(conv(t0, rows, ch, R.W(Un + "conv1.conv.weight"), ch, R.W(Un + "conv1.conv.bias"), t1, 7, dil));
            hipLaunchKernelGGL((snake_kernel<T>), el((size_t)rows * ch), dim3(256), 0, s, (const T*)t1, (const T*)R.W(Un + "act2.alpha"), (const T*)R.W(Un + "act2.beta"), t0, (size_t)rows * ch, ch);
            { GemmArgs a = conv(t0, rows, ch, R.W(Un + "conv2.conv.weight"), ch, R.W(Un + "conv2.conv.bias"), t2, 1, 1);
              a.res = y; a.ldr = ch; R.gemm<T>(a); }
            std::swap(y, t2);
        }
    }
    const std::string F1 = DD + std::to_string(g.n_rates + 1) + ".", F2 = DD + std::to_string(g.n_rates + 2) + ".";
    hipLaunchKernelGGL((snake_kernel<T>), el((size_t)rows * ch), dim3(256), 0, s, (const T*)y, (const T*)R.W(F1 + "alpha"), (const T*)R.W(F1 + "beta"), t0, (size_t)rows * ch, ch);
    hipLaunchKernelGGL((final_conv_kernel<T>), dim3((rows + 255) / 256), dim3(256), 0, s, (const T*)t0, (const T*)R.W(F2 + "conv.weight"), (const T*)R.W(F2 + "conv.bias"), pcm, rows, ch);
    if (R.err) return R.err;
    if (rows != (int)samples_for(g, Tn)) return cfail(FQ3_ESTATE, "internal: sample count mismatch");
    return 0;
}

extern "C" int fq3_codec_decode(fq3_codec* c, const int64_t* codes, int T, float* pcm, void* stream) {
    if (!c || !codes || !pcm) return cfail(FQ3_EINVAL, "null argument");
    if (!c->ready) return cfail(FQ3_ESTATE, "codec weights not finalized");
    if (T < 1) return cfail(FQ3_EINVAL, "need at least 1 frame");
    if (T > c->cfg.max_frames) return cfail(FQ3_ETOOLONG, "codec decode: " + std::to_string(T) + " frames exceed max_frames=" + std::to_string(c->cfg.max_frames));
    hipStream_t s = (hipStream_t)stream;
    int r = c->cfg.dtype == FQ3_BF16 ? decode_t<bf16_t>(c, codes, T, pcm, s) : decode_t<float>(c, codes, T, pcm, s);
    if (r) return r;
    CHIP(hipGetLastError());
    return FQ3_OK;
}
