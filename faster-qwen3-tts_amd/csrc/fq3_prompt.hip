// Prompt builder on the device: replaces the ~40 small ATen launches of the reference's
// _build_talker_inputs_local (/root/reference/faster_qwen3_tts/model.py:583-805) and upstream generate_icl_prompt
// (called at model.py:699-712) with: one gather, two matrix-core GEMMs (text_projection: fc1 + SiLU, fc2) over ALL text
// tokens of the prompt at once, and one row-assembly kernel.  Rounding points are those of the module-by-module Torch
// execution: one rounding to T after fc1, after SiLU, after fc2, after the 16-way embedding sum, after text + codec.
#define FQ3_SKINNY_EXTERN           // skinny_gemm.cuh: the weight-stationary GEMM kernels are instantiated in fq3_prefill.hip only
#include "fq3_ctx.h"
#include "codec_kernels.cuh"

using namespace fq3;

namespace {

template <typename T>
__global__ __launch_bounds__(256) void text_gather_kernel(const int64_t* ids, const T* table, T* out, int n, int Ht, int vocab) {
    const int r = blockIdx.x;
    int64_t id = ids[r];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);            // F.embedding would fault; clamp keeps the launch safe
    const T* src = table + (size_t)id * Ht;
    for (int e = threadIdx.x * 8; e < Ht; e += 256 * 8) {
        Raw8<T> v; ldraw<false>(v, src + e);
        float f[8]; unpack(v, f);
        DT<T>::st8(out + (size_t)r * Ht + e, f);
    }
}

struct RowTables { const void* t[16]; };

// one workgroup per prompt row
template <typename T>
__global__ __launch_bounds__(256) void prompt_rows_kernel(const T* text_rows, int n_text, const int* prog, const int64_t* ref_codes,
                                                          int n_ref, const T* spk, RowTables tabs, int vocab0, int vocab_rest,
                                                          T* out, int H) {
    const int r = blockIdx.x;
    const int trow = prog[r * 3 + 0], kind = prog[r * 3 + 1], arg = prog[r * 3 + 2];
    const bool has_text = trow >= 0 && trow < n_text;
    for (int e = threadIdx.x * 8; e < H; e += 256 * 8) {
        float c[8], t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { c[i] = 0.f; t[i] = 0.f; }
        bool has_codec = true;
        if (kind == 1) {
            const int id = arg < 0 ? 0 : (arg >= vocab0 ? vocab0 - 1 : arg);
            Raw8<T> v; ldraw<false>(v, reinterpret_cast<const T*>(tabs.t[0]) + (size_t)id * H + e);
            unpack(v, c);
        } else if (kind == 2) {
            Raw8<T> v; ldraw<false>(v, spk + e);
            unpack(v, c);
        } else if (kind == 3) {
            const int fr = arg < 0 ? 0 : (arg >= n_ref ? n_ref - 1 : arg);
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int vmax = g == 0 ? vocab0 : vocab_rest;
                int64_t id = ref_codes[(size_t)fr * 16 + g];
                id = id < 0 ? 0 : (id >= vmax ? vmax - 1 : id);
                Raw8<T> v; ldraw<false>(v, reinterpret_cast<const T*>(tabs.t[g]) + (size_t)id * H + e);
                float f[8]; unpack(v, f);
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] += f[i];
            }
#pragma unroll
            for (int i = 0; i < 8; i += 2) DT<T>::rnd2(c[i], c[i + 1]);          // torch: cat(...).sum(1), one rounding
        } else has_codec = false;
        if (has_text) {
            Raw8<T> v; ldraw<false>(v, text_rows + (size_t)trow * H + e);
            unpack(v, t);
        }
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = (has_text && has_codec) ? t[i] + c[i] : (has_text ? t[i] : c[i]);
        DT<T>::st8(out + (size_t)r * H + e, o);
    }
}

template <typename T>
int text_project_t(fq3_ctx* c, const int64_t* ids, int n, void* out, hipStream_t s) {
    const fq3_prompt_weights& w = c->pw;
    const int Ht = w.text_hidden, H = c->cfg.talker.hidden;
    if (n > c->pw_cap) {
        const int cap = (n + 255) / 256 * 256;
        void *x = nullptr, *h = nullptr;
        if (int r = fq3_dmalloc_(c, &x, (size_t)cap * Ht * c->esz)) return r;       // old workspaces stay owned by the context
        if (int r = fq3_dmalloc_(c, &h, (size_t)cap * Ht * c->esz)) return r;
        c->pw_x = x; c->pw_h = h; c->pw_cap = cap;
    }
    hipLaunchKernelGGL((text_gather_kernel<T>), dim3(n), dim3(256), 0, s, ids, (const T*)w.text_embedding, (T*)c->pw_x, n, Ht, w.text_vocab);
    GemmArgs a{};
    a.A = c->pw_x; a.lda = Ht; a.M = n; a.a_rows = n; a.n_taps = 1; a.tap_off[0] = 0; a.Cin = Ht; a.W = w.fc1_w; a.N = Ht;
    a.bias = w.fc1_b; a.bias_mod = Ht; a.Y = c->pw_h; a.ldy = Ht; a.act = 3;
    // fc1 output and SiLU output are two module outputs in Torch (two roundings): the epilogue rounds after the bias and
    // again after the activation
    gemm_launch<T>(a, s);
    GemmArgs b{};
    b.A = c->pw_h; b.lda = Ht; b.M = n; b.a_rows = n; b.n_taps = 1; b.tap_off[0] = 0; b.Cin = Ht; b.W = w.fc2_w; b.N = H;
    b.bias = w.fc2_b; b.bias_mod = H; b.Y = out; b.ldy = H;
    gemm_launch<T>(b, s);
    return 0;
}

}  // namespace

extern "C" int fq3_bind_prompt_weights(fq3_ctx* c, const fq3_prompt_weights* w) {
    if (!c || !w) return fq3_fail_(FQ3_EINVAL, "null argument");
    if (!w->text_embedding || !w->fc1_w || !w->fc1_b || !w->fc2_w || !w->fc2_b) return fq3_fail_(FQ3_EINVAL, "prompt weight table has null entries");
    if (w->text_vocab <= 0 || w->text_hidden <= 0 || w->text_hidden % 32 || c->cfg.talker.hidden % 32)
        return fq3_fail_(FQ3_EUNSUPPORTED, "text_hidden and the talker hidden size must be multiples of 32 (MFMA K step)");
    c->pw = *w;
    c->pw_bound = true;
    return FQ3_OK;
}

extern "C" int fq3_text_project(fq3_ctx* c, const int64_t* ids, int n, void* out, void* stream) {
    if (!c || !ids || !out) return fq3_fail_(FQ3_EINVAL, "null argument");
    if (!c->pw_bound) return fq3_fail_(FQ3_ESTATE, "prompt weights not bound");
    if (n <= 0) return fq3_fail_(FQ3_EINVAL, "need at least one token");
    hipStream_t s = (hipStream_t)stream;
    int r = c->cfg.dtype == FQ3_BF16 ? text_project_t<bf16_t>(c, ids, n, out, s) : text_project_t<float>(c, ids, n, out, s);
    if (r) return r;
    if (hipGetLastError() != hipSuccess) return fq3_fail_(FQ3_EHIP, "fq3_text_project launch failed");
    return FQ3_OK;
}

extern "C" int fq3_prompt_rows(fq3_ctx* c, const void* text_rows, int n_text, const int32_t* prog, int n_rows,
                               const int64_t* ref_codes, int n_ref, const void* spk_embed, void* out, void* stream) {
    if (!c || !prog || !out) return fq3_fail_(FQ3_EINVAL, "null argument");
    if (!c->bound) return fq3_fail_(FQ3_ESTATE, "weights not bound");
    if (n_rows <= 0) return fq3_fail_(FQ3_EINVAL, "need at least one row");
    if (c->cfg.num_code_groups != 16) return fq3_fail_(FQ3_EUNSUPPORTED, "the prompt builder is built for 16 code groups");
    if (c->cfg.talker.hidden % 8) return fq3_fail_(FQ3_EUNSUPPORTED, "hidden size must be a multiple of 8");
    RowTables tabs{};
    tabs.t[0] = c->wt.codec_embedding;
    for (int i = 1; i < 16; ++i) tabs.t[i] = c->pemb[i - 1];
    hipStream_t s = (hipStream_t)stream;
    const int H = c->cfg.talker.hidden, V0 = c->cfg.talker.vocab, V1 = c->cfg.predictor.vocab;
    if (c->cfg.dtype == FQ3_BF16)
        hipLaunchKernelGGL((prompt_rows_kernel<bf16_t>), dim3(n_rows), dim3(256), 0, s, (const bf16_t*)text_rows, text_rows ? n_text : 0, prog,
                           ref_codes, ref_codes ? n_ref : 0, (const bf16_t*)spk_embed, tabs, V0, V1, (bf16_t*)out, H);
    else
        hipLaunchKernelGGL((prompt_rows_kernel<float>), dim3(n_rows), dim3(256), 0, s, (const float*)text_rows, text_rows ? n_text : 0, prog,
                           ref_codes, ref_codes ? n_ref : 0, (const float*)spk_embed, tabs, V0, V1, (float*)out, H);
    if (hipGetLastError() != hipSuccess) return fq3_fail_(FQ3_EHIP, "fq3_prompt_rows launch failed");
    return FQ3_OK;
}
