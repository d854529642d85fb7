// Reference-audio analysis behind the C ABI: what upstream's create_voice_clone_prompt computes for a new reference clip
// (the reference wrapper calls it at faster_qwen3_tts/model.py:430-447 and caches the result, :424-463):
//   fq3_refenc_encode   24 kHz waveform -> int64 codes [T][16]   (speech_tokenizer.encode; Mimi-style encoder)
//   fq3_refenc_speaker  24 kHz waveform -> x-vector [enc_dim]    (log-mel + ECAPA-TDNN)
// Structure follows the locally readable transformers siblings (modeling_mimi.py, modeling_qwen2_5_omni.py:2412-2707);
// the mapping of the Qwen3-TTS checkpoints onto them is [recalled], pinned by oracle/refenc_oracle.py
// against those modules with seeded weights.
//
// Weight binding contract (fq3hip/refenc.py packs once at load; all fp32):
//   encoder.encoder.layers.0.conv.weight                  [C0][k]                      (in_channels = 1)
//   encoder.encoder.layers.<i>.block.<1|3>.conv.weight    [Cout][k][Cin]               (torch [Cout, Cin, k])
//   encoder.encoder.layers.<i>.conv.weight  (stride r)    [Cout][2][r*Cin]  tap t, column j*Cin + c = torch[co][c][t*r + j]
//   encoder.encoder.layers.<last>.conv.weight             [hidden][k][Cin]
//   encoder.encoder_transformer.layers.<l>.self_attn.qkv.weight [3*QD][hidden]; o_proj / mlp.fc1 / mlp.fc2 as stored;
//       input_layernorm / post_attention_layernorm .weight/.bias; self_attn_layer_scale.scale / mlp_layer_scale.scale
//   encoder.downsample.conv.weight                        [hidden][2][2*hidden]        (k = 4, stride 2, as the strided convs)
//   encoder.quantizer.input_proj.weight                   [2*D][hidden]                (semantic rows, then acoustic rows)
//   encoder.quantizer.codebook.<lv>.embed_sum [K][D] / .cluster_usage [K]              (lv 0.. = semantic first)
//   encoder.rope.cos / encoder.rope.sin                   [max_positions][head_dim/2]
//   speaker_encoder.mel.dft [2*NB][n_fft] (hann * cos | hann * sin, rows >= n_fft/2+1 of each half zero), .mel.basis [mel][NB]
//   speaker_encoder.blocks.0.conv.weight [C][k][mel]; .blocks.<b>.{tdnn1,tdnn2}.conv.weight [C][1][C];
//   .blocks.<b>.res2net_block.blocks.<i>.conv.weight [C/s][k][C/s]; .blocks.<b>.se_block.conv{1,2}.weight;
//   .mfa.conv.weight [Cm][1][Cm]; .asp.tdnn.conv.weight_h [A][Cm], .weight_ms [A][2*Cm]; .asp.conv.weight [Cm][A]; .fc.weight [E][2*Cm]
//   (+ the matching .bias vectors)
#define FQ3_SKINNY_EXTERN           // skinny_gemm.cuh: the weight-stationary GEMM kernels are instantiated in fq3_prefill.hip only
#include "../../include/fq3hip.h"
#include "refenc_kernels.cuh"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace fq3;

extern "C" void fq3_set_error_(const char* msg);
static int rfail(int code, const std::string& m) { fq3_set_error_(m.c_str()); return code; }
#define RHIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return rfail(FQ3_EHIP, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

struct fq3_refenc {
    fq3_refenc_config cfg{};
    std::map<std::string, const void*> w;
    std::map<std::string, int64_t> wn;
    float* ws = nullptr; size_t ws_floats = 0, ws_used = 0;
    float* books = nullptr;                    // prepared codebooks: per level emb [K][D] then embT [D][K]
    bool tok_ready = false, spk_ready = false;
};

static int need(fq3_refenc* r, const std::string& n, int64_t numel, const float** out) {
    auto it = r->w.find(n);
    if (it == r->w.end()) return rfail(FQ3_ESTATE, "reference-audio weight not bound: " + n);
    if (numel > 0 && r->wn[n] != numel)
        return rfail(FQ3_EINVAL, "reference-audio weight has wrong size: " + n + " (" + std::to_string(r->wn[n]) + " vs " + std::to_string(numel) + ")");
    if (out) *out = (const float*)it->second;
    return 0;
}

extern "C" int fq3_refenc_create(const fq3_refenc_config* cfg, fq3_refenc** out) {
    if (!cfg || !out) return rfail(FQ3_EINVAL, "null argument");
    const auto& g = *cfg;
    if (g.n_ratios < 1 || g.n_ratios > 8 || g.n_enc < 3 || g.n_enc > 8) return rfail(FQ3_EINVAL, "ratio / channel lists");
    auto m32 = [](int v) { return v > 0 && v % 32 == 0; };
    bool ok = m32(g.num_filters / g.compress) && m32(g.hidden) && m32(g.inter) && m32(g.n_heads * g.head_dim) && m32(g.codebook_dim) &&
              m32(g.mel_dim) && m32(g.hop) && m32(g.n_bins_padded) && m32(g.attn_channels) && m32(g.se_channels) && m32(g.enc_dim);
    for (int i = 0; i < g.n_enc; ++i) ok = ok && m32(g.enc_channels[i]);
    if (!ok) return rfail(FQ3_EUNSUPPORTED, "every channel count must be a multiple of 32 (MFMA K step)");
    if (g.head_dim != 32 && g.head_dim != 64 && g.head_dim != 128) return rfail(FQ3_EUNSUPPORTED, "encoder head_dim must be 32, 64 or 128");
    if (g.sliding_window < 1 || g.sliding_window > 256) return rfail(FQ3_EUNSUPPORTED, "encoder sliding_window must be in 1..256");
    if (g.codebook_size != 256 && g.codebook_size != 512 && g.codebook_size != 1024 && g.codebook_size != 2048 && g.codebook_size != 4096)
        return rfail(FQ3_EUNSUPPORTED, "codebook_size must be 256, 512, 1024, 2048 or 4096");
    if (g.num_quantizers > 32 || g.num_semantic < 1 || g.num_semantic >= g.num_quantizers) return rfail(FQ3_EINVAL, "quantizer counts");
    if (g.n_fft % g.hop || g.n_fft / g.hop > kMaxTaps || g.kernel_size > kMaxTaps) return rfail(FQ3_EUNSUPPORTED, "n_fft must be a small multiple of hop");
    if (g.n_bins_padded < g.n_fft / 2 + 1) return rfail(FQ3_EINVAL, "n_bins_padded < n_fft/2 + 1");
    for (int i = 1; i < g.n_enc - 1; ++i)
        if (g.enc_channels[i] != g.enc_channels[0] || g.enc_channels[i] % g.res2net_scale || !m32(g.enc_channels[i] / g.res2net_scale))
            return rfail(FQ3_EUNSUPPORTED, "SE-Res2Net blocks must keep the channel count, in chunks that are multiples of 32");
    if (g.enc_channels[g.n_enc - 1] != g.enc_channels[0] * (g.n_enc - 2) || g.enc_kernel_sizes[g.n_enc - 1] != 1)
        return rfail(FQ3_EUNSUPPORTED, "the aggregation layer must be 1x1 over the concatenated block outputs");
    fq3_refenc* r = new fq3_refenc();
    r->cfg = g;
    *out = r;
    return FQ3_OK;
}

extern "C" int fq3_refenc_destroy(fq3_refenc* r) {
    if (!r) return FQ3_OK;
    (void)hipDeviceSynchronize();
    if (r->ws) (void)hipFree(r->ws);
    if (r->books) (void)hipFree(r->books);
    delete r;
    return FQ3_OK;
}

extern "C" int fq3_refenc_bind(fq3_refenc* r, const char* name, const void* ptr, int64_t numel) {
    if (!r || !name || !ptr) return rfail(FQ3_EINVAL, "null argument");
    r->w[name] = ptr; r->wn[name] = numel; r->tok_ready = r->spk_ready = false;
    return FQ3_OK;
}

static bool has_prefix(const fq3_refenc* r, const char* p) {
    auto it = r->w.lower_bound(p);
    return it != r->w.end() && it->first.compare(0, strlen(p), p) == 0;
}

extern "C" int fq3_refenc_finalize(fq3_refenc* r, void* stream) {
    if (!r) return rfail(FQ3_EINVAL, "null argument");
    const auto& g = r->cfg;
    hipStream_t s = (hipStream_t)stream;
    int e;
    r->tok_ready = r->spk_ready = false;
    if (has_prefix(r, "encoder.")) {
        const int K = g.codebook_size, D = g.codebook_dim;
        if ((e = need(r, "encoder.encoder.layers.0.conv.weight", (int64_t)g.num_filters * g.kernel_size, nullptr))) return e;
        if ((e = need(r, "encoder.quantizer.input_proj.weight", (int64_t)2 * D * g.hidden, nullptr))) return e;
        if ((e = need(r, "encoder.rope.cos", (int64_t)g.max_positions * (g.head_dim / 2), nullptr))) return e;
        if ((e = need(r, "encoder.rope.sin", (int64_t)g.max_positions * (g.head_dim / 2), nullptr))) return e;
        if (r->books) { (void)hipFree(r->books); r->books = nullptr; }
        RHIP(hipMalloc((void**)&r->books, (size_t)g.num_quantizers * 2 * K * D * sizeof(float)));
        for (int lv = 0; lv < g.num_quantizers; ++lv) {
            const float *es = nullptr, *us = nullptr;
            const std::string b = "encoder.quantizer.codebook." + std::to_string(lv) + ".";
            if ((e = need(r, b + "embed_sum", (int64_t)K * D, &es)) || (e = need(r, b + "cluster_usage", K, &us))) return e;
            float* emb = r->books + (size_t)lv * 2 * K * D;
            hipLaunchKernelGGL(codebook_prepare_kernel, dim3((K * D + 255) / 256), dim3(256), 0, s, es, us, emb, emb + (size_t)K * D, K, D, 1e-5f);
        }
        RHIP(hipStreamSynchronize(s));
        r->tok_ready = true;
    }
    if (has_prefix(r, "speaker_encoder.")) {
        const int NB = g.n_bins_padded;
        if ((e = need(r, "speaker_encoder.mel.dft", (int64_t)2 * NB * g.n_fft, nullptr))) return e;
        if ((e = need(r, "speaker_encoder.mel.basis", (int64_t)g.mel_dim * NB, nullptr))) return e;
        if ((e = need(r, "speaker_encoder.fc.weight", (int64_t)g.enc_dim * 2 * g.enc_channels[g.n_enc - 1], nullptr))) return e;
        r->spk_ready = true;
    }
    if (!r->tok_ready && !r->spk_ready) return rfail(FQ3_ESTATE, "no encoder.* or speaker_encoder.* weights bound");
    return FQ3_OK;
}

static int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

extern "C" int64_t fq3_refenc_num_frames(const fq3_refenc* r, int64_t n) {
    if (!r || n < 1) return -1;
    for (int i = 0; i < r->cfg.n_ratios; ++i) n = ceil_div(n, r->cfg.ratios[i]);
    return ceil_div(n, 2);
}

// bump allocator over one lazily grown device workspace (grown only between calls, after a stream sync)
static int ws_reserve(fq3_refenc* r, size_t floats, hipStream_t s) {
    r->ws_used = 0;
    if (floats <= r->ws_floats) return 0;
    RHIP(hipStreamSynchronize(s));
    if (r->ws) { (void)hipFree(r->ws); r->ws = nullptr; r->ws_floats = 0; }
    RHIP(hipMalloc((void**)&r->ws, floats * sizeof(float)));
    r->ws_floats = floats;
    return 0;
}
static float* take(fq3_refenc* r, size_t floats) {
    floats = (floats + 63) & ~(size_t)63;
    float* p = r->ws + r->ws_used;
    r->ws_used += floats;
    return r->ws_used <= r->ws_floats ? p : nullptr;
}

static GemmArgs gemm(const float* A, int lda, int M, int a_rows, int Cin, const float* W, int N, const float* bias, float* Y, int ldy) {
    GemmArgs a{}; a.A = A; a.lda = lda; a.M = M; a.a_rows = a_rows; a.n_taps = 1; a.tap_off[0] = 0; a.Cin = Cin; a.W = W; a.N = N;
    a.bias = bias; a.bias_mod = N; a.Y = Y; a.ldy = ldy; return a;
}
static dim3 el(size_t n) { return dim3((unsigned)((n + 255) / 256)); }

// ---- speech-tokenizer encoder -------------------------------------------------------------------------------------------
extern "C" int fq3_refenc_encode(fq3_refenc* r, const float* pcm, int64_t n, int64_t* codes, void* stream) {
    if (!r || !pcm || !codes) return rfail(FQ3_EINVAL, "null argument");
    if (!r->tok_ready) return rfail(FQ3_ESTATE, "speech-tokenizer encoder weights not finalized");
    if (n < 1 || n > (int64_t)1 << 26) return rfail(FQ3_EINVAL, "sample count");
    const auto& g = r->cfg;
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    auto W = [&](const std::string& name) -> const float* { const float* p = nullptr; if (!err) err = need(r, name, 0, &p); return p; };
    // the 25 Hz length bounds the transformer; the rope tables bound that
    int64_t t25 = n;
    for (int i = 0; i < g.n_ratios; ++i) t25 = ceil_div(t25, g.ratios[i]);
    if (t25 > g.max_positions) return rfail(FQ3_ETOOLONG, "reference clip longer than the encoder's position table");
    const int QD = g.n_heads * g.head_dim;
    int maxr = 2;
    for (int i = 0; i < g.n_ratios; ++i) maxr = std::max(maxr, g.ratios[i]);
    size_t big = (size_t)(n + maxr) * g.num_filters + 256;                   // rows x channels never grows along the stack ...
    {   // ... except for the zeroed tail of a strided conv's row view, which a stack of small ratios can push past it
        int64_t t = n; size_t ch = (size_t)g.num_filters;
        for (int i = 0; i < g.n_ratios; ++i) {
            const int64_t tn = ceil_div(t, g.ratios[i]);
            big = std::max(big, (size_t)(tn * g.ratios[i]) * ch + 256);
            t = tn; ch *= 2;
        }
        big = std::max(big, (size_t)t * ch + 256);
    }
    const size_t tr = (size_t)(t25 + 4) * std::max(std::max(3 * QD, g.inter), 2 * g.hidden) + 256;
    const size_t per = std::max(big, tr);
    if (ws_reserve(r, 4 * per + 1024, s)) return FQ3_EHIP;
    float* A = take(r, per); float* B = take(r, per); float* Cb = take(r, per); float* Dd = take(r, per);
    if (!Dd) return rfail(FQ3_ESTATE, "internal: workspace accounting");
    const std::string E = "encoder.encoder.layers.";
    // conv_in: raw -> A, ELU -> B
    int C = g.num_filters;
    int64_t T = n;
    hipLaunchKernelGGL(conv_in_kernel, el((size_t)T * C), dim3(256), 0, s, pcm, W(E + "0.conv.weight"), W(E + "0.conv.bias"), A, B, (long)T, C, g.kernel_size);
    if (err) return err;
    float *H = A, *Ecur = B, *Eother = Dd;
    int li = 1;
    for (int st = 0; st < g.n_ratios && !err; ++st) {
        const int rr = g.ratios[st], Cm = C / g.compress;
        for (int j = 0; j < g.n_residual_layers; ++j, ++li) {
            int dil = 1;
            for (int q = 0; q < j; ++q) dil *= g.dilation_growth_rate;
            const std::string Rb = E + std::to_string(li) + ".block.";
            const int k = g.residual_kernel_size;
            // ELU -> conv k (C -> C/compress) -> ELU, kept only activated (Cb)
            GemmArgs a = gemm(Ecur, C, (int)T, (int)T, C, W(Rb + "1.conv.weight"), Cm, W(Rb + "1.conv.bias"), Cb, Cm);
            a.n_taps = k; for (int i = 0; i < k; ++i) a.tap_off[i] = -(k - 1 - i) * dil;
            a.act = 5;
            if (err) return err;
            gemm_launch<float>(a, s);
            // conv 1x1 (C/compress -> C) + skip; raw -> H (next skip / unused), ELU -> Ecur (the k-conv above is done with it)
            GemmArgs b = gemm(Cb, Cm, (int)T, (int)T, Cm, W(Rb + "3.conv.weight"), C, W(Rb + "3.conv.bias"), H, C);
            b.res = H; b.ldr = C; b.Y2 = Ecur; b.act2 = 1;
            if (err) return err;
            gemm_launch<float>(b, s);
        }
        ++li;                                                   // the nn.ELU() entry of the module list
        // strided conv k = 2r, stride r on the [T/r][r*C] view of the activated rows; the tail of the last view row is zero
        const int64_t Tn = ceil_div(T, rr);
        if (Tn * rr > T) RHIP(hipMemsetAsync(Ecur + (size_t)T * C, 0, (size_t)(Tn * rr - T) * C * sizeof(float), s));
        GemmArgs a = gemm(Ecur, rr * C, (int)Tn, (int)Tn, rr * C, W(E + std::to_string(li) + ".conv.weight"), 2 * C,
                          W(E + std::to_string(li) + ".conv.bias"), H, 2 * C);
        a.n_taps = 2; a.tap_off[0] = -1; a.tap_off[1] = 0;
        a.Y2 = Eother; a.act2 = 1;
        if (err) return err;
        gemm_launch<float>(a, s);
        ++li;
        std::swap(Ecur, Eother);
        T = Tn; C *= 2;
    }
    ++li;                                                       // ELU
    {   // last conv (k3) -> X [T][hidden] in Cb
        const int k = g.last_kernel_size;
        GemmArgs a = gemm(Ecur, C, (int)T, (int)T, C, W(E + std::to_string(li) + ".conv.weight"), g.hidden, W(E + std::to_string(li) + ".conv.bias"), Cb, g.hidden);
        a.n_taps = k; for (int i = 0; i < k; ++i) a.tap_off[i] = -(k - 1 - i);
        if (err) return err;
        gemm_launch<float>(a, s);
    }
    // ---- transformer (pre-LayerNorm, RoPE, causal sliding window, GELU MLP, layer scale) ----
    float* X = Cb; float* N1 = A; float* QKV = B; float* AT = Dd;
    const int Tt = (int)T, Hd = g.hidden;
    const float* cosT = W("encoder.rope.cos"); const float* sinT = W("encoder.rope.sin");
    const int NP = (std::min(g.sliding_window, Tt) + 63) / 64;
    for (int l = 0; l < g.n_layers && !err; ++l) {
        const std::string L = "encoder.encoder_transformer.layers." + std::to_string(l) + ".";
        const float *ln1w = W(L + "input_layernorm.weight"), *ln1b = W(L + "input_layernorm.bias");
        const float *ln2w = W(L + "post_attention_layernorm.weight"), *ln2b = W(L + "post_attention_layernorm.bias");
        const float *wqkv = W(L + "self_attn.qkv.weight"), *wo = W(L + "self_attn.o_proj.weight");
        const float *w1 = W(L + "mlp.fc1.weight"), *w2 = W(L + "mlp.fc2.weight");
        const float *ls1 = W(L + "self_attn_layer_scale.scale"), *ls2 = W(L + "mlp_layer_scale.scale");
        if (err) return err;
        hipLaunchKernelGGL((layernorm_rows_kernel<float>), dim3((Tt + 3) / 4), dim3(256), 0, s, (const float*)X, ln1w, ln1b, N1, 0, Tt, Hd, g.norm_eps);
        gemm_launch<float>(gemm(N1, Hd, Tt, Tt, Hd, wqkv, 3 * QD, nullptr, QKV, 3 * QD), s);
        hipLaunchKernelGGL((rope_rows_kernel<float>), el((size_t)Tt * 2 * g.n_heads * (g.head_dim / 2)), dim3(256), 0, s, QKV, cosT, sinT, Tt, QD, g.head_dim, 0);
        const dim3 ag((Tt + 3) / 4, g.n_heads);
        const float sc = 1.0f / sqrtf((float)g.head_dim);
#define FQ3_WIN_ATTN(HD, NPV) hipLaunchKernelGGL((win_attn_kernel<HD, NPV>), ag, dim3(256), 0, s, (const float*)QKV, AT, Tt, g.n_heads, g.sliding_window, sc)
#define FQ3_WIN_ATTN_HD(HD) do { if (NP <= 1) FQ3_WIN_ATTN(HD, 1); else if (NP == 2) FQ3_WIN_ATTN(HD, 2); else if (NP == 3) FQ3_WIN_ATTN(HD, 3); else FQ3_WIN_ATTN(HD, 4); } while (0)
        if (g.head_dim == 32) FQ3_WIN_ATTN_HD(32); else if (g.head_dim == 64) FQ3_WIN_ATTN_HD(64); else FQ3_WIN_ATTN_HD(128);
        { GemmArgs a = gemm(AT, QD, Tt, Tt, QD, wo, Hd, nullptr, X, Hd); a.scale = ls1; a.res = X; a.ldr = Hd; gemm_launch<float>(a, s); }
        hipLaunchKernelGGL((layernorm_rows_kernel<float>), dim3((Tt + 3) / 4), dim3(256), 0, s, (const float*)X, ln2w, ln2b, N1, 0, Tt, Hd, g.norm_eps);
        { GemmArgs a = gemm(N1, Hd, Tt, Tt, Hd, w1, g.inter, nullptr, QKV, g.inter); a.act = 1; gemm_launch<float>(a, s); }
        { GemmArgs a = gemm(QKV, g.inter, Tt, Tt, g.inter, w2, Hd, nullptr, X, Hd); a.scale = ls2; a.res = X; a.ldr = Hd; gemm_launch<float>(a, s); }
    }
    if (err) return err;
    // ---- stride-2 conv (k = 4, replicate padding: 2 copies of the first row, the last row repeated to an even length) ----
    const int T5 = (Tt + 1) / 2, Lp = 2 + 2 * T5;
    hipLaunchKernelGGL(pad_rows_kernel, el((size_t)Lp * Hd), dim3(256), 0, s, (const float*)X, Hd, (const float*)nullptr, 0, A, Hd, Tt, Hd, 2, Lp, 1);
    {
        GemmArgs a = gemm(A, 2 * Hd, T5, Lp / 2, 2 * Hd, W("encoder.downsample.conv.weight"), Hd, nullptr, B, Hd);
        a.n_taps = 2; a.tap_off[0] = 0; a.tap_off[1] = 1;
        if (err) return err;
        gemm_launch<float>(a, s);
    }
    // ---- split residual VQ: both input projections as one GEMM, then one workgroup per (frame, quantizer group) ----
    const int D = g.codebook_dim, K = g.codebook_size;
    gemm_launch<float>(gemm(B, Hd, T5, T5, Hd, W("encoder.quantizer.input_proj.weight"), 2 * D, nullptr, Dd, 2 * D), s);
    if (err) return err;
    RvqEncArgs ra{}; ra.nq = g.num_quantizers; ra.n_sem = g.num_semantic; ra.K = K; ra.D = D;
    for (int lv = 0; lv < g.num_quantizers; ++lv) { ra.emb[lv] = r->books + (size_t)lv * 2 * K * D; ra.embT[lv] = ra.emb[lv] + (size_t)K * D; }
#define FQ3_RVQ(V, P) hipLaunchKernelGGL((rvq_encode_kernel<V, P>), dim3(T5, 2), dim3(256), D * sizeof(float), s, ra, (const float*)Dd, codes)
    switch (K) {
        case 256: FQ3_RVQ(1, 1); break;
        case 512: FQ3_RVQ(1, 2); break;
        case 1024: FQ3_RVQ(4, 1); break;
        case 2048: FQ3_RVQ(4, 2); break;
        default: FQ3_RVQ(4, 4); break;
    }
    RHIP(hipGetLastError());
    return FQ3_OK;
}

// ---- speaker encoder ------------------------------------------------------------------------------------------------------
// one TDNN layer with torch's padding="same", padding_mode="reflect": materialise the padded rows, then a tap GEMM over them
static void tdnn(fq3_refenc* r, hipStream_t s, const float* x, int ldx, const float* x2, int ldx2, int T, int Cin, int k, int dil,
                 const float* W, const float* bias, int Cout, float* y, int ldy, int act, float* padbuf) {
    const float* A = x; int lda = ldx, a_rows = T;
    const int pad = (k - 1) * dil / 2;
    if (k > 1 || x2) {
        hipLaunchKernelGGL(pad_rows_kernel, el((size_t)(T + 2 * pad) * Cin), dim3(256), 0, s, x, ldx, x2, ldx2, padbuf, Cin, T, Cin, pad, T + 2 * pad, 0);
        A = padbuf; lda = Cin; a_rows = T + 2 * pad;
    }
    GemmArgs a = gemm(A, lda, T, a_rows, Cin, W, Cout, bias, y, ldy);
    a.n_taps = k; for (int i = 0; i < k; ++i) a.tap_off[i] = i * dil;
    a.act = act;
    gemm_launch<float>(a, s);
}

extern "C" int fq3_refenc_speaker(fq3_refenc* r, const float* pcm, int64_t n, float* embed, float* mel_out, void* stream) {
    if (!r || !pcm || !embed) return rfail(FQ3_EINVAL, "null argument");
    if (!r->spk_ready) return rfail(FQ3_ESTATE, "speaker-encoder weights not finalized");
    const auto& g = r->cfg;
    const int pad = (g.n_fft - g.hop) / 2;
    if (n <= pad || n < g.hop || n > (int64_t)1 << 26) return rfail(FQ3_EINVAL, "clip too short (or too long) for the mel front end");
    hipStream_t s = (hipStream_t)stream;
    int err = 0;
    auto W = [&](const std::string& name) -> const float* { const float* p = nullptr; if (!err) err = need(r, name, 0, &p); return p; };
    const int64_t Lp = n + 2 * pad;
    const int F = (int)((Lp - g.n_fft) / g.hop + 1), NB = g.n_bins_padded, taps = g.n_fft / g.hop;
    const int Cb = g.enc_channels[0], Cm = g.enc_channels[g.n_enc - 1], nblk = g.n_enc - 2, Cs = Cb / g.res2net_scale, Ac = g.attn_channels;
    int maxpad = 0;
    for (int i = 0; i < g.n_enc; ++i) maxpad = std::max(maxpad, (g.enc_kernel_sizes[i] - 1) * g.enc_dilations[i]);
    const size_t Fp = (size_t)F + maxpad + 8;
    size_t total = (size_t)Lp + g.hop + (size_t)F * 2 * NB + (size_t)F * NB + (size_t)F * g.mel_dim + Fp * std::max(g.mel_dim, Cs) +
                   4 * (size_t)F * Cb + 2 * (size_t)F * Cm + (size_t)F * Ac + (size_t)F * Cm + 16 * (size_t)Cm + 64 * 32;
    if (ws_reserve(r, total, s)) return FQ3_EHIP;
    float* sig = take(r, (size_t)Lp + g.hop); float* spec = take(r, (size_t)F * 2 * NB); float* mag = take(r, (size_t)F * NB);
    float* mel = take(r, (size_t)F * g.mel_dim); float* padbuf = take(r, Fp * std::max(g.mel_dim, Cs));
    float* X0 = take(r, (size_t)F * Cb); float* U = take(r, (size_t)F * Cb); float* R = take(r, (size_t)F * Cb); float* V = take(r, (size_t)F * Cb);
    float* Cat = take(r, (size_t)F * Cm); float* Hm = take(r, (size_t)F * Cm); float* A1 = take(r, (size_t)F * Ac); float* Lg = take(r, (size_t)F * Cm);
    float* vec = take(r, 16 * (size_t)Cm);
    if (!vec) return rfail(FQ3_ESTATE, "internal: workspace accounting");
    // ---- log-mel: reflect pad, windowed DFT as a tap GEMM over hop-sized rows, magnitude, mel projection, log ----
    hipLaunchKernelGGL(pad_rows_kernel, el((size_t)Lp), dim3(256), 0, s, pcm, 1, (const float*)nullptr, 0, sig, 1, (int)n, 1, pad, (int)Lp, 0);
    {
        GemmArgs a = gemm(sig, g.hop, F, (int)(Lp / g.hop), g.hop, W("speaker_encoder.mel.dft"), 2 * NB, nullptr, spec, 2 * NB);
        a.n_taps = taps; for (int i = 0; i < taps; ++i) a.tap_off[i] = i;
        if (err) return err;
        gemm_launch<float>(a, s);
    }
    hipLaunchKernelGGL(dft_mag_kernel, el((size_t)F * NB), dim3(256), 0, s, (const float*)spec, mag, (long)F, NB);
    { GemmArgs a = gemm(mag, NB, F, F, NB, W("speaker_encoder.mel.basis"), g.mel_dim, nullptr, mel, g.mel_dim); a.act = 8; if (err) return err; gemm_launch<float>(a, s); }
    if (mel_out) RHIP(hipMemcpyAsync(mel_out, mel, (size_t)F * g.mel_dim * sizeof(float), hipMemcpyDeviceToDevice, s));
    // ---- ECAPA-TDNN ----
    const std::string S = "speaker_encoder.";
    tdnn(r, s, mel, g.mel_dim, nullptr, 0, F, g.mel_dim, g.enc_kernel_sizes[0], g.enc_dilations[0], W(S + "blocks.0.conv.weight"), W(S + "blocks.0.conv.bias"), Cb, X0, Cb, 4, padbuf);
    if (err) return err;
    const float* xin = X0; int ldin = Cb;
    float* mean = vec; float* gate1 = vec + Cm; float* gate = vec + 2 * Cm;
    for (int b = 1; b <= nblk && !err; ++b) {
        const std::string Bk = S + "blocks." + std::to_string(b) + ".";
        const int k = g.enc_kernel_sizes[b], dil = g.enc_dilations[b];
        tdnn(r, s, xin, ldin, nullptr, 0, F, Cb, 1, 1, W(Bk + "tdnn1.conv.weight"), W(Bk + "tdnn1.conv.bias"), Cb, U, Cb, 4, padbuf);
        // Res2Net: chunk 0 passes through; chunk i = TDNN(chunk_i (+ previous output))
        hipLaunchKernelGGL(pad_rows_kernel, el((size_t)F * Cs), dim3(256), 0, s, (const float*)U, Cb, (const float*)nullptr, 0, R, Cb, F, Cs, 0, F, 0);
        for (int i = 1; i < g.res2net_scale && !err; ++i) {
            const std::string Rn = Bk + "res2net_block.blocks." + std::to_string(i - 1) + ".conv.";
            tdnn(r, s, U + i * Cs, Cb, i >= 2 ? R + (i - 1) * Cs : nullptr, Cb, F, Cs, k, dil, W(Rn + "weight"), W(Rn + "bias"), Cs, R + i * Cs, Cb, 4, padbuf);
        }
        tdnn(r, s, R, Cb, nullptr, 0, F, Cb, 1, 1, W(Bk + "tdnn2.conv.weight"), W(Bk + "tdnn2.conv.bias"), Cb, V, Cb, 4, padbuf);
        // squeeze-excitation: mean over time -> 1x1 -> ReLU -> 1x1 -> sigmoid -> channel gate, + block input
        hipLaunchKernelGGL(col_stats_kernel, dim3((Cb + 63) / 64), dim3(64 * kStatSlices), 0, s, (const float*)V, Cb, (const float*)nullptr, 0, mean, (float*)nullptr, F, Cb, 0.f);
        { GemmArgs a = gemm(mean, Cb, 1, 1, Cb, W(Bk + "se_block.conv1.weight"), g.se_channels, W(Bk + "se_block.conv1.bias"), gate1, g.se_channels); a.act = 4; if (err) return err; gemm_launch<float>(a, s); }
        { GemmArgs a = gemm(gate1, g.se_channels, 1, 1, g.se_channels, W(Bk + "se_block.conv2.weight"), Cb, W(Bk + "se_block.conv2.bias"), gate, Cb); a.act = 7; if (err) return err; gemm_launch<float>(a, s); }
        float* dst = Cat + (size_t)(b - 1) * Cb;
        hipLaunchKernelGGL(se_scale_kernel, el((size_t)F * Cb), dim3(256), 0, s, (const float*)V, Cb, (const float*)gate, xin, ldin, dst, Cm, F, Cb);
        xin = dst; ldin = Cm;
    }
    if (err) return err;
    tdnn(r, s, Cat, Cm, nullptr, 0, F, Cm, 1, 1, W(S + "mfa.conv.weight"), W(S + "mfa.conv.bias"), Cm, Hm, Cm, 4, padbuf);
    // ---- attentive statistics pooling: the time-constant [mean | std] part of the attention input becomes a bias ----
    float* ms = vec + 3 * Cm;           // [mean | std] (2 Cm)
    float* bias2 = vec + 5 * Cm;        // [Ac]
    float* pooled = vec + 6 * Cm;       // [mean | std] (2 Cm)
    hipLaunchKernelGGL(col_stats_kernel, dim3((Cm + 63) / 64), dim3(64 * kStatSlices), 0, s, (const float*)Hm, Cm, (const float*)nullptr, 0, ms, ms + Cm, F, Cm, 1e-12f);
    { GemmArgs a = gemm(ms, 2 * Cm, 1, 1, 2 * Cm, W(S + "asp.tdnn.conv.weight_ms"), Ac, W(S + "asp.tdnn.conv.bias"), bias2, Ac); if (err) return err; gemm_launch<float>(a, s); }
    { GemmArgs a = gemm(Hm, Cm, F, F, Cm, W(S + "asp.tdnn.conv.weight_h"), Ac, bias2, A1, Ac); a.act = 6; if (err) return err; gemm_launch<float>(a, s); }
    { GemmArgs a = gemm(A1, Ac, F, F, Ac, W(S + "asp.conv.weight"), Cm, W(S + "asp.conv.bias"), Lg, Cm); if (err) return err; gemm_launch<float>(a, s); }
    hipLaunchKernelGGL(col_stats_kernel, dim3((Cm + 63) / 64), dim3(64 * kStatSlices), 0, s, (const float*)Hm, Cm, (const float*)Lg, Cm, pooled, pooled + Cm, F, Cm, 1e-12f);
    { GemmArgs a = gemm(pooled, 2 * Cm, 1, 1, 2 * Cm, W(S + "fc.weight"), g.enc_dim, W(S + "fc.bias"), embed, g.enc_dim); if (err) return err; gemm_launch<float>(a, s); }
    RHIP(hipGetLastError());
    return FQ3_OK;
}
