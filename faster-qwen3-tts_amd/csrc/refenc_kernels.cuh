// Kernels only the reference-audio analysers use (fq3_refenc.hip): the speech-tokenizer ENCODER (24 kHz waveform ->
// 12.5 Hz codes) and the speaker encoder (waveform -> log-mel -> ECAPA-TDNN x-vector).  fp32 throughout: the products are
// integer codes and one embedding, computed once per reference clip and cached by the caller (model.py:424-463).
// The dense work (every conv / linear, the windowed DFT and the mel projection) runs through conv_gemm_kernel<float>
// (codec_kernels.cuh); what is here is the glue those GEMMs cannot express.  Activations are [time][channels].
#pragma once
#include "codec_kernels.cuh"

namespace fq3 {

// first encoder conv: 1 -> C channels, kernel K, causal (K - 1 zeros on the left).  y raw, e = ELU(y).
// w [C][K], one thread per (t, c).
__global__ void conv_in_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                               float* __restrict__ y, float* __restrict__ e, long n, int C, int K) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * C) return;
    const long t = i / C; const int c = (int)(i - t * C);
    float acc = b[c];
    for (int k = 0; k < K; ++k) {
        const long tt = t - (K - 1) + k;
        if (tt >= 0) acc = fmaf(w[c * K + k], x[tt], acc);
    }
    y[i] = acc;
    e[i] = elu1(acc);
}

// dst[r][c] = src1[idx(r)][c] (+ src2[idx(r)][c]) for r in [0, T + 2 * pad... ) with idx = r - pad_l mapped back into [0, T):
// mode 0 reflect (no edge repeat, torch "reflect"), mode 1 replicate (edge).  Sources may be channel slices (ld1 / ld2).
__global__ void pad_rows_kernel(const float* __restrict__ src1, int ld1, const float* __restrict__ src2, int ld2,
                                float* __restrict__ dst, int ldd, int T, int C, int pad_l, int rows_out, int mode) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)rows_out * C) return;
    const int r = (int)(i / C), c = (int)(i - (long)r * C);
    int t = r - pad_l;
    if (mode == 0) {
        if (t < 0) t = -t;
        if (t >= T) t = 2 * (T - 1) - t;
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);        // only reachable when T <= pad (degenerate clips)
    } else {
        t = t < 0 ? 0 : (t >= T ? T - 1 : t);
    }
    float v = src1[(size_t)t * ld1 + c];
    if (src2) v += src2[(size_t)t * ld2 + c];
    dst[(size_t)r * ldd + c] = v;
}

// Causal sliding-window attention over qkv [T][3*QD] with up to NP * 64 keys per query (Mimi: window 250, head_dim 64).
// Same shape as swa_attn_kernel (one wave per (query, head), one key per lane and pass), generalised to NP passes.
template <int HD, int NP>
__global__ __launch_bounds__(256) void win_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int Tn, int NH, int window, float scale) {
    __shared__ float qs[4][HD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * 4 + wave, h = blockIdx.y;
    const int QD = NH * HD;
    const bool live = q < Tn;
    const int qq = live ? q : Tn - 1;
    const float* qp = qkv + (size_t)qq * 3 * QD + (size_t)h * HD;
    for (int d = lane; d < HD; d += 64) qs[wave][d] = qp[d];
    __syncthreads();
    if (!live) return;
    const int k_lo = max(0, q - window + 1), nk = q - k_lo + 1;
    float sc[NP], p[NP];
    float mx = -INFINITY;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const int j = ps * 64 + lane;
        const int kk = k_lo + (j < nk ? j : 0);
        const float4* kp = reinterpret_cast<const float4*>(qkv + (size_t)kk * 3 * QD + QD + (size_t)h * HD);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < HD / 4; ++c) {
            const float4 kv = kp[c];
            s = fmaf(qs[wave][c * 4 + 0], kv.x, s); s = fmaf(qs[wave][c * 4 + 1], kv.y, s);
            s = fmaf(qs[wave][c * 4 + 2], kv.z, s); s = fmaf(qs[wave][c * 4 + 3], kv.w, s);
        }
        sc[ps] = j < nk ? s * scale : -INFINITY;
        mx = fmaxf(mx, sc[ps]);
    }
    mx = wave_max(mx);
    float l = 0.f;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) { p[ps] = expf(sc[ps] - mx); l += p[ps]; }
    l = wave_sum(l);
    float o0 = 0.f, o1 = 0.f;
    const float* vbase = qkv + (size_t)k_lo * 3 * QD + 2 * QD + (size_t)h * HD;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps) {
        const int lim = min(64, nk - ps * 64);
        for (int j = 0; j < lim; ++j) {
            const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(p[ps]), j));
            const float* vp = vbase + (size_t)(ps * 64 + j) * 3 * QD;
            if (HD >= 64 || lane < HD) o0 = fmaf(pj, vp[lane < HD ? lane : 0], o0);
            if (HD > 64) o1 = fmaf(pj, vp[lane + 64], o1);
        }
    }
    float* op = out + (size_t)q * QD + (size_t)h * HD;
    if (lane < HD) op[lane] = o0 / l;
    if (HD > 64) op[lane + 64] = o1 / l;
}

// codebook = embed_sum / clamp(cluster_usage, eps) (modeling_mimi.py:980-983), stored both ways: emb [K][D] and embT [D][K]
__global__ void codebook_prepare_kernel(const float* __restrict__ embed_sum, const float* __restrict__ usage,
                                        float* __restrict__ emb, float* __restrict__ embT, int K, int D, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)K * D) return;
    const int k = (int)(i / D), d = (int)(i - (long)k * D);
    const float v = embed_sum[i] / fmaxf(usage[k], eps);
    emb[i] = v;
    embT[(size_t)d * K + k] = v;
}

// Residual vector quantisation of one frame by one workgroup (modeling_mimi.py:1062-1079): per level, nearest codebook
// row in Euclidean distance (first index on ties, torch.argmin), residual -= that row.  blockIdx.y = 0: the semantic
// quantizer (levels [0, n_sem), input columns [0, D)); 1: the acoustic one (levels [n_sem, nq), columns [D, 2D)).
// Each thread owns PER groups of VEC consecutive codes (K = 256 * VEC * PER) and walks the D dims of the TRANSPOSED
// codebook: a wave reads VEC * 256 contiguous bytes per load, no branch sits between the loads (the compiler keeps
// them in flight), and the d loop is unrolled so that 4 * PER loads per thread overlap -- the scan is bound by the
// codebook bytes a CU can pull from L2 (2 MB per level at K = 2048, D = 256).
struct RvqEncArgs { const float* emb[32]; const float* embT[32]; int nq; int n_sem; int K; int D; };
template <int VEC, int PER>
__global__ __launch_bounds__(256) void rvq_encode_kernel(RvqEncArgs a, const float* __restrict__ proj /*[T][2D]*/, int64_t* __restrict__ codes /*[T][nq]*/) {
    extern __shared__ float rsm[];
    float* res = rsm;                       // [D]
    __shared__ float best_d[4]; __shared__ int best_i[4]; __shared__ int winner;
    const int t = blockIdx.x, part = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lvl0 = part == 0 ? 0 : a.n_sem, lvl1 = part == 0 ? a.n_sem : a.nq;
    const int D = a.D, K = a.K;
    for (int d = tid; d < D; d += 256) res[d] = proj[(size_t)t * 2 * D + (size_t)part * D + d];
    __syncthreads();
    for (int lv = lvl0; lv < lvl1; ++lv) {
        const float* eT = a.embT[lv] + tid * VEC;
        float acc[PER][VEC];
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[j][v] = 0.f;
#pragma unroll 4
        for (int d = 0; d < D; ++d) {
            const float r = res[d];
            const float* row = eT + (size_t)d * K;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float e[VEC];
                if constexpr (VEC == 4) { const float4 q = *reinterpret_cast<const float4*>(row + j * 1024); e[0] = q.x; e[1] = q.y; e[2] = q.z; e[3] = q.w; }
                else e[0] = row[j * 256];
#pragma unroll
                for (int v = 0; v < VEC; ++v) { const float df = r - e[v]; acc[j][v] = fmaf(df, df, acc[j][v]); }
            }
        }
        float bd = INFINITY; int bi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < PER; ++j)
#pragma unroll
            for (int v = 0; v < VEC; ++v)
                if (acc[j][v] < bd) { bd = acc[j][v]; bi = j * 256 * VEC + tid * VEC + v; }     // ascending index: first minimum kept
        // wave argmin (distance, then index), then across the 4 waves
        for (int off = 32; off >= 1; off >>= 1) {
            const float od = __shfl_xor(bd, off); const int oi = __shfl_xor(bi, off);
            if (od < bd || (od == bd && oi < bi)) { bd = od; bi = oi; }
        }
        if (lane == 0) { best_d[wave] = bd; best_i[wave] = bi; }
        __syncthreads();
        if (tid == 0) {
            float d0 = best_d[0]; int i0 = best_i[0];
            for (int w = 1; w < 4; ++w) if (best_d[w] < d0 || (best_d[w] == d0 && best_i[w] < i0)) { d0 = best_d[w]; i0 = best_i[w]; }
            winner = i0;
            codes[(size_t)t * a.nq + lv] = i0;
        }
        __syncthreads();
        const float* e = a.emb[lv] + (size_t)winner * D;
        for (int d = tid; d < D; d += 256) res[d] -= e[d];
        __syncthreads();
    }
}

// |X|: spec [F][2*NB] = (re | im) of the windowed DFT -> mag [F][NB] = sqrt(re^2 + im^2 + 1e-9) (BigVGAN-style mel front end)
__global__ void dft_mag_kernel(const float* __restrict__ spec, float* __restrict__ mag, long F, int NB) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= F * NB) return;
    const long f = i / NB; const int k = (int)(i - f * NB);
    const float re = spec[f * 2 * NB + k], im = spec[f * 2 * NB + NB + k];
    mag[i] = sqrtf(re * re + im * im + 1e-9f);
}

// Per-channel statistics over time of x [T][ldx] (channels [0, C)), 64 channels x 16 time slices per workgroup.
//   logits == nullptr: uniform weights 1/T (SE mean; the pooling layer's global context)
//   logits != nullptr: weights = softmax over time of logits[:, c] (attentive statistics pooling)
// mean[c] = sum w x, std[c] = sqrt(max(sum w (x - mean)^2, eps)); either output may be null.
constexpr int kStatSlices = 16;
__global__ __launch_bounds__(64 * kStatSlices) void col_stats_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ logits, int ldl,
                                                        float* __restrict__ mean, float* __restrict__ stdv, int T, int C, float eps) {
    __shared__ float red[kStatSlices][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const bool ok = c < C;
    const int cc = ok ? c : C - 1;
    auto reduce = [&](float v, bool is_max) {
        red[part][cl] = v;
        __syncthreads();
        float r = red[0][cl];
        for (int p = 1; p < kStatSlices; ++p) r = is_max ? fmaxf(r, red[p][cl]) : r + red[p][cl];
        __syncthreads();
        return r;
    };
    float mx = 0.f, denom = (float)T;
    if (logits) {
        float m = -INFINITY;
        for (int t = part; t < T; t += kStatSlices) m = fmaxf(m, logits[(size_t)t * ldl + cc]);
        mx = reduce(m, true);
        float s = 0.f;
        for (int t = part; t < T; t += kStatSlices) s += expf(logits[(size_t)t * ldl + cc] - mx);
        denom = reduce(s, false);
    }
    float s1 = 0.f;
    for (int t = part; t < T; t += kStatSlices) {
        const float w = logits ? expf(logits[(size_t)t * ldl + cc] - mx) / denom : 1.0f / denom;
        s1 = fmaf(w, x[(size_t)t * ldx + cc], s1);
    }
    const float mu = reduce(s1, false);
    if (ok && part == 0 && mean) mean[c] = mu;
    if (!stdv) return;
    float s2 = 0.f;
    for (int t = part; t < T; t += kStatSlices) {
        const float w = logits ? expf(logits[(size_t)t * ldl + cc] - mx) / denom : 1.0f / denom;
        const float d = x[(size_t)t * ldx + cc] - mu;
        s2 = fmaf(w, d * d, s2);
    }
    const float var = reduce(s2, false);
    if (ok && part == 0) stdv[c] = sqrtf(fmaxf(var, eps));
}

// squeeze-excitation output: y[t][c] = x[t][c] * gate[c] + res[t][c]
__global__ void se_scale_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gate, const float* __restrict__ res, int ldr,
                                float* __restrict__ y, int ldy, int T, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)T * C) return;
    const int t = (int)(i / C), c = (int)(i - (long)t * C);
    y[(size_t)t * ldy + c] = x[(size_t)t * ldx + c] * gate[c] + res[(size_t)t * ldr + c];
}

}  // namespace fq3
