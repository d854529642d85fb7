// Matrix-core prefill of the talker (replaces the upstream eager `talker.forward(inputs_embeds=...)`,
// /root/reference/faster_qwen3_tts/generate.py:107-122): all prompt rows go through each layer as GEMMs
// on MFMA (conv_gemm_kernel with one tap), K/V are normalised, rotated and written straight into the
// static cache layout the decode kernels read, attention is causal over the live (non-padded) keys.
// Rounding points are those of the decode path (one rounding to T per Linear / norm / RoPE / residual add),
// so prefill + decode match the oracle's matrix-form prefill.
#include "fq3_ctx.h"
#include "codec_kernels.cuh"

using namespace fq3;

namespace {

// per (token, head): RMSNorm over 128 + RoPE for q (in place) and k (-> cache); v copied to the cache
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kv_kernel(T* qkv, const T* qw, const T* kw, float eps, const float* cos_tab,
                                                              const float* sin_tab, int rope_len, int rope_delta, T* kcache,
                                                              T* vcache, int max_seq, int L, int n_pad, int NH, int NKV) {
    constexpr int HD = kHeadDim;
    const int per = NH + 2 * NKV;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= L * per) return;
    const int t = w / per, v = w - t * per;
    if (t < n_pad) return;
    T* src = qkv + (size_t)t * per * HD + (size_t)v * HD;
    float x0 = DT<T>::ld(src + lane), x1 = DT<T>::ld(src + lane + 64);
    if (v < NH + NKV) {
        const T* gw = v < NH ? qw : kw;
        const float ss = wave_sum(fmaf(x0, x0, x1 * x1));
        const float rs = 1.0f / sqrtf(ss / (float)HD + eps);
        const float n0 = DT<T>::rnd(DT<T>::ld(gw + lane) * DT<T>::rnd(x0 * rs));
        const float n1 = DT<T>::rnd(DT<T>::ld(gw + lane + 64) * DT<T>::rnd(x1 * rs));
        int rp = t + rope_delta;
        rp = rp < 0 ? 0 : (rp >= rope_len ? rope_len - 1 : rp);
        const float cs = cos_tab[(size_t)rp * 64 + lane], sn = sin_tab[(size_t)rp * 64 + lane];
        x0 = DT<T>::rnd(DT<T>::rnd(n0 * cs) + DT<T>::rnd(-n1 * sn));
        x1 = DT<T>::rnd(DT<T>::rnd(n1 * cs) + DT<T>::rnd(n0 * sn));
    }
    T* dst = v < NH ? src : (v < NH + NKV ? kcache + ((size_t)(v - NH) * max_seq + t) * HD
                                           : vcache + ((size_t)(v - NH - NKV) * max_seq + t) * HD);
    DT<T>::st(dst + lane, x0);
    DT<T>::st(dst + lane + 64, x1);
}

// causal attention for the prompt: one wave per (query row, q head); 16 lanes per key (8 dims each),
// 16 keys in flight per trip; fp32 online softmax, one rounding at the end.
template <typename T>
__global__ __launch_bounds__(256) void prefill_attn_kernel(const T* qkv, const T* kcache, const T* vcache, T* out, int max_seq,
                                                           int L, int n_pad, int NH, int NKV, float scale) {
    constexpr int HD = kHeadDim;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (w >= L * NH) return;
    const int t = w / NH, h = w - t * NH;
    const int per = NH + 2 * NKV, g = h / (NH / NKV);
    const int sub = lane >> 4, c = lane & 15;
    T* op = out + ((size_t)t * NH + h) * HD;
    if (t < n_pad) {
        if (sub == 0)
#pragma unroll
            for (int i = 0; i < 8; ++i) DT<T>::st(op + c * 8 + i, 0.f);
        return;
    }
    Raw8<T> qraw;
    ldraw<false>(qraw, qkv + (size_t)t * per * HD + (size_t)h * HD + c * 8);
    float q[8];
    unpack(qraw, q);
    const T* kc = kcache + (size_t)g * max_seq * HD;
    const T* vc = vcache + (size_t)g * max_seq * HD;
    float m = -1e30f, l = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    for (int k0 = n_pad; k0 <= t; k0 += 16) {
        Raw8<T> kr[4], vr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = k0 + i * 4 + sub;
            key = key <= t ? key : t;
            ldraw<false>(kr[i], kc + (size_t)key * HD + c * 8);
            ldraw<false>(vr[i], vc + (size_t)key * HD + c * 8);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool valid = k0 + i * 4 + sub <= t;
            float kf[8], vf[8];
            unpack(kr[i], kf); unpack(vr[i], vf);
            float sc = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) sc = fmaf(q[d], kf[d], sc);
            sc += __shfl_xor(sc, 1, 64); sc += __shfl_xor(sc, 2, 64);
            sc += __shfl_xor(sc, 4, 64); sc += __shfl_xor(sc, 8, 64);
            sc = valid ? sc * scale : -INFINITY;
            const float mn = fmaxf(m, sc), al = __expf(m - mn), p = __expf(sc - mn);
            l = fmaf(l, al, p);
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] = fmaf(o[d], al, valid ? p * vf[d] : 0.f);
            m = mn;
        }
    }
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        const float mo = __shfl_xor(m, off, 64), lo = __shfl_xor(l, off, 64);
        const float M = fmaxf(m, mo), wa = __expf(m - M), wb = __expf(mo - M);
        l = l * wa + lo * wb;
#pragma unroll
        for (int d = 0; d < 8; ++d) { const float oo = __shfl_xor(o[d], off, 64); o[d] = o[d] * wa + oo * wb; }
        m = M;
    }
    if (sub == 0)
#pragma unroll
        for (int i = 0; i < 8; ++i) DT<T>::st(op + c * 8 + i, o[i] / l);
}

template <typename T>
GemmArgs lin(const void* A, int M, int K, const void* W, int N, void* Y) {
    GemmArgs a{}; a.A = A; a.lda = K; a.M = M; a.a_rows = M; a.n_taps = 1; a.tap_off[0] = 0; a.Cin = K; a.W = W; a.N = N;
    a.bias_mod = N; a.Y = Y; a.ldy = N; return a;
}
template <typename T> void gemm(const GemmArgs& a, hipStream_t s) { gemm_launch<T>(a, s); }

template <typename T>
int prefill_t(fq3_ctx* c, const void* embeds, int L, int n_pad, void* out_logits, void* out_hidden, hipStream_t s) {
    const fq3_stack_dims& d = c->cfg.talker;
    const int H = d.hidden, I = d.inter, NH = d.n_heads, NKV = d.n_kv_heads;
    const int QD = NH * kHeadDim, KVD = NKV * kHeadDim, per = QD + 2 * KVD;
    if (H % 32 || I % 32) return fq3_fail_(FQ3_EUNSUPPORTED, "MFMA prefill needs hidden and intermediate sizes that are multiples of 32");
    const size_t rows = (size_t)c->cfg.max_seq_len;
    if (!c->pf_x) {
        int r;
        if ((r = fq3_dmalloc_(c, &c->pf_x, rows * H * c->esz))) return r;
        if ((r = fq3_dmalloc_(c, &c->pf_xn, rows * H * c->esz))) return r;
        if ((r = fq3_dmalloc_(c, &c->pf_qkv, rows * per * c->esz))) return r;
        if ((r = fq3_dmalloc_(c, &c->pf_att, rows * QD * c->esz))) return r;
        if ((r = fq3_dmalloc_(c, &c->pf_gu, rows * 2 * I * c->esz))) return r;
        if ((r = fq3_dmalloc_(c, &c->pf_act, rows * I * c->esz))) return r;
    }
    T *X = (T*)c->pf_x, *XN = (T*)c->pf_xn, *QKV = (T*)c->pf_qkv, *ATT = (T*)c->pf_att, *GU = (T*)c->pf_gu, *ACT = (T*)c->pf_act;
    if (hipMemcpyAsync(X, embeds, (size_t)L * H * c->esz, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fq3_fail_(FQ3_EHIP, "prefill: copy of the prompt embeddings failed");
    const float scale = 1.0f / sqrtf((float)kHeadDim);
    for (int i = 0; i < d.n_layers; ++i) {
        const fq3_layer_weights& w = c->tl[i];
        hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), dim3((L + 3) / 4), dim3(256), 0, s, (const T*)X, (const T*)w.input_norm, XN, 0, L, H, d.rms_eps);
        gemm<T>(lin<T>(XN, L, H, w.qkv, per, QKV), s);
        hipLaunchKernelGGL((qk_norm_rope_kv_kernel<T>), dim3((L * (NH + 2 * NKV) + 3) / 4), dim3(256), 0, s, QKV, (const T*)w.q_norm,
                           (const T*)w.k_norm, d.rms_eps, c->wt.talker_cos, c->wt.talker_sin, c->wt.talker_rope_len, c->rope_delta,
                           (T*)c->tk.k[i], (T*)c->tk.v[i], c->tk.max_seq, L, n_pad, NH, NKV);
        hipLaunchKernelGGL((prefill_attn_kernel<T>), dim3((L * NH + 3) / 4), dim3(256), 0, s, (const T*)QKV, (const T*)c->tk.k[i],
                           (const T*)c->tk.v[i], ATT, c->tk.max_seq, L, n_pad, NH, NKV, scale);
        { GemmArgs a = lin<T>(ATT, L, QD, w.o, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
        hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), dim3((L + 3) / 4), dim3(256), 0, s, (const T*)X, (const T*)w.post_norm, XN, 0, L, H, d.rms_eps);
        gemm<T>(lin<T>(XN, L, H, w.gate_up, 2 * I, GU), s);
        hipLaunchKernelGGL((silu_mul_kernel<T>), dim3((unsigned)(((size_t)L * I + 255) / 256)), dim3(256), 0, s, (const T*)GU, ACT, L, I);
        { GemmArgs a = lin<T>(ACT, L, I, w.down, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
    }
    // final norm of the last row only -> past_hidden; logits through the decode-path head GEMV
    hipLaunchKernelGGL((rmsnorm_kernel<T>), dim3(1), dim3(256), 0, s, (const T*)X + (size_t)(L - 1) * H, (const T*)c->wt.talker_final_norm,
                       (T*)out_hidden, H, d.rms_eps);
    if (out_logits) return fq3_codec_head_launch_(c, out_hidden, out_logits, s);
    return 0;
}

}  // namespace

int fq3_prefill_mfma_(fq3_ctx* c, const void* embeds, int L, int n_pad, void* out_logits, void* out_hidden, hipStream_t s) {
    return c->cfg.dtype == FQ3_BF16 ? prefill_t<bf16_t>(c, embeds, L, n_pad, out_logits, out_hidden, s)
                                    : prefill_t<float>(c, embeds, L, n_pad, out_logits, out_hidden, s);
}
