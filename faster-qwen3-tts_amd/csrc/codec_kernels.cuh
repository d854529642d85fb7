// Kernels of the 12 Hz RVQ codec decoder (vocoder).  Activations are channels-last [time][C] in the
// context dtype T, so every dense contraction (1x1 / k7 dilated / transposed convs, Linear layers) is
// ONE implicit-GEMM kernel on the matrix cores: A rows are time rows gathered at per-tap offsets,
// B is the pre-packed weight [N][taps*Cin].  bf16 uses v_mfma_f32_16x16x32_bf16, fp32 (parity mode)
// uses the exact v_mfma_f32_16x16x4_f32.  Everything else (RVQ gather, norms, SnakeBeta, depthwise
// conv, RoPE, sliding-window attention) is HBM-bound row work with 16-byte accesses.
// Rounding points follow the Torch module execution of the sibling implementation
// (transformers modeling_qwen3_omni_moe.py:3180-3263, :3542-3696): one rounding to T per op output.
#pragma once
#include "fq3_common.cuh"

namespace fq3 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kMaxTaps = 8;
struct GemmArgs {
    const void* A; int lda; int M; int a_rows;        // rows outside [0, a_rows) read as zeros (causal left pad)
    int n_taps; int tap_off[kMaxTaps]; int Cin;       // K = n_taps * Cin; A row of (m, tap) = m + tap_off[tap]
    const void* W; int N;                             // packed weight [N][K]
    const void* bias; int bias_mod;                   // bias[n % bias_mod] (bias_mod = Cout for transposed convs)
    const void* scale;                                // optional per-n multiplier after the activation
    const void* res; int ldr;                         // optional residual [M][ldr]
    void* Y; int ldy;
    int act;                                          // 0 none, 1 exact GELU
};

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// 64x64 output tile per workgroup (4 waves as 2x2, each wave 32x32 = 2x2 MFMA tiles), K step 32.
template <typename T>
__global__ __launch_bounds__(256) void conv_gemm_kernel(GemmArgs a) {
    constexpr int BM = 64, BN = 64, BK = 32;
    constexpr int LDS_LD = BK + (sizeof(T) == 2 ? 8 : 4);      // padded row (elements): breaks the 64/128-byte stride
    __shared__ __attribute__((aligned(16))) T As[BM * LDS_LD];
    __shared__ __attribute__((aligned(16))) T Bs[BN * LDS_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int wr = wave >> 1, wc = wave & 1;
    const int K = a.n_taps * a.Cin;
    const T* A = reinterpret_cast<const T*>(a.A);
    const T* W = reinterpret_cast<const T*>(a.W);
    constexpr int EPT = 16 / sizeof(T);                 // elements per 16-byte access
    constexpr int TPR = BK / EPT;                       // threads per tile row
    constexpr int RPP = 256 / TPR;                      // rows per pass
    constexpr int NPASS = BM / RPP;
    f32x4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int lr = tid / TPR, lc = (tid % TPR) * EPT;
    u32x4 areg[NPASS], breg[NPASS];
    auto gload = [&](int k0) {
        const int tap = k0 / a.Cin, ci = k0 - tap * a.Cin;
        const int toff = a.tap_off[tap];
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = lr + p * RPP;
            const int m = m0 + r, ar = m + toff;
            areg[p] = u32x4{0u, 0u, 0u, 0u};
            if (m < a.M && ar >= 0 && ar < a.a_rows)
                areg[p] = *reinterpret_cast<const u32x4*>(A + (size_t)ar * a.lda + ci + lc);
            const int n = n0 + r;
            breg[p] = u32x4{0u, 0u, 0u, 0u};
            if (n < a.N) breg[p] = *reinterpret_cast<const u32x4*>(W + (size_t)n * K + k0 + lc);
        }
    };
    gload(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int r = lr + p * RPP;
            *reinterpret_cast<u32x4*>(As + r * LDS_LD + lc) = areg[p];
            *reinterpret_cast<u32x4*>(Bs + r * LDS_LD + lc) = breg[p];
        }
        __syncthreads();
        if (k0 + BK < K) gload(k0 + BK);                // next tile's loads fly under the MFMAs
        const int fr = lane & 15, fq = lane >> 4;
        if constexpr (sizeof(T) == 2) {
            bf16x8_t af[2], bfr[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                af[i] = *reinterpret_cast<const bf16x8_t*>(As + (wr * 32 + i * 16 + fr) * LDS_LD + fq * 8);
                bfr[i] = *reinterpret_cast<const bf16x8_t*>(Bs + (wc * 32 + i * 16 + fr) * LDS_LD + fq * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                float af[2], bfr[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    af[i] = reinterpret_cast<const float*>(As)[(wr * 32 + i * 16 + fr) * LDS_LD + kk + fq];
                    bfr[i] = reinterpret_cast<const float*>(Bs)[(wc * 32 + i * 16 + fr) * LDS_LD + kk + fq];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // epilogue: C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg
    T* Y = reinterpret_cast<T*>(a.Y);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wc * 32 + j * 16 + (lane & 15);
            if (n >= a.N) continue;
            const float b = a.bias ? DT<T>::ld(reinterpret_cast<const T*>(a.bias) + (n % a.bias_mod)) : 0.f;
            const float sc = a.scale ? DT<T>::ld(reinterpret_cast<const T*>(a.scale) + n) : 1.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * 32 + i * 16 + (lane >> 4) * 4 + r;
                if (m >= a.M) continue;
                float v = DT<T>::rnd(acc[i][j][r] + b);
                if (a.act == 1) v = DT<T>::rnd(gelu_exact(v));
                if (a.scale) v = DT<T>::rnd(sc * v);
                if (a.res) v = v + DT<T>::ld(reinterpret_cast<const T*>(a.res) + (size_t)m * a.ldr + n);
                DT<T>::st(Y + (size_t)m * a.ldy + n, v);
            }
        }
}

// ---- RVQ: sequential (rounded) sum of codebook rows; one block per frame ------------------------------
struct RvqArgs { const void* books[32]; int nq; int n_first; int dim; };
template <typename T>
__global__ void rvq_gather_kernel(RvqArgs a, const int64_t* codes, T* first, T* rest, int Tn) {
    const int t = blockIdx.x;
    for (int d = threadIdx.x; d < a.dim; d += blockDim.x) {
        float f = 0.f, r = 0.f;
        bool hf = false, hr = false;
        for (int j = 0; j < a.nq; ++j) {
            const float e = DT<T>::ld(reinterpret_cast<const T*>(a.books[j]) + (size_t)codes[(size_t)t * a.nq + j] * a.dim + d);
            if (j < a.n_first) { f = hf ? DT<T>::rnd(f + e) : e; hf = true; }
            else { r = hr ? DT<T>::rnd(r + e) : e; hr = true; }
        }
        DT<T>::st(first + (size_t)t * a.dim + d, f);
        DT<T>::st(rest + (size_t)t * a.dim + d, r);
    }
}

// ---- row norms: one wave per row ------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const T* x, const T* w, T* y, int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + (size_t)row * C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = DT<T>::ld(xr + c); ss = fmaf(v, v, ss); }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)C + eps);
    for (int c = lane; c < C; c += 64)
        DT<T>::st(y + (size_t)row * C + c, DT<T>::ld(w + c) * DT<T>::rnd(DT<T>::ld(xr + c) * rs));
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T* x, const T* w, const T* b, T* y, int rows, int C, float eps) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const T* xr = x + (size_t)row * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += DT<T>::ld(xr + c);
    const float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int c = lane; c < C; c += 64) { const float d = DT<T>::ld(xr + c) - mean; v = fmaf(d, d, v); }
    const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)C + eps);
    for (int c = lane; c < C; c += 64)
        DT<T>::st(y + (size_t)row * C + c, (DT<T>::ld(xr + c) - mean) * rstd * DT<T>::ld(w + c) + DT<T>::ld(b + c));
}

// ---- elementwise ------------------------------------------------------------------------------------------
// SnakeBeta: x + 1/(exp(beta)+1e-9) * sin(x*exp(alpha))^2, each Torch op rounded to T (modeling :3566-3580)
template <typename T>
__global__ void snake_kernel(const T* x, const T* alpha, const T* beta, T* y, size_t n, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const float a = DT<T>::rnd(expf(DT<T>::ld(alpha + c)));
    const float b = DT<T>::rnd(expf(DT<T>::ld(beta + c)));
    const float ib = DT<T>::rnd(1.0f / DT<T>::rnd(b + 1e-9f));
    const float v = DT<T>::ld(x + i);
    const float s = DT<T>::rnd(sinf(DT<T>::rnd(v * a)));
    DT<T>::st(y + i, v + DT<T>::rnd(ib * DT<T>::rnd(s * s)));
}

// causal depthwise conv k=7 over time, channels-last
template <typename T>
__global__ void dwconv7_kernel(const T* x, const T* w /*[C][7]*/, const T* b, T* y, int rows, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const int tt = t - 6 + k;
        if (tt >= 0) acc = fmaf(DT<T>::ld(w + c * 7 + k), DT<T>::ld(x + (size_t)tt * C + c), acc);
    }
    DT<T>::st(y + i, acc + DT<T>::ld(b + c));
}

template <typename T>
__global__ void silu_mul_kernel(const T* gu, T* y, int rows, int I) {     // gu [rows][2I] = gate | up
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * I) return;
    const int r = (int)(i / I), c = (int)(i % I);
    const float g = DT<T>::ld(gu + (size_t)r * 2 * I + c), u = DT<T>::ld(gu + (size_t)r * 2 * I + I + c);
    DT<T>::st(y + i, DT<T>::rnd(g / (1.0f + expf(-g))) * u);
}

// RoPE on the q and k thirds of qkv [rows][3*QD], head_dim HD (rotate_half convention), position = row
template <typename T>
__global__ void rope_rows_kernel(T* qkv, const float* cos_tab, const float* sin_tab, int rows, int QD, int HD) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int half = HD / 2, per_row = 2 * (QD / HD) * half;
    if (i >= (size_t)rows * per_row) return;
    const int t = (int)(i / per_row), r = (int)(i % per_row);
    const int head = r / half, j = r % half;              // heads 0..QD/HD-1 = q, then k
    T* p = qkv + (size_t)t * 3 * QD + (size_t)head * HD;
    const float cs = cos_tab[(size_t)t * half + j], sn = sin_tab[(size_t)t * half + j];
    const float x0 = DT<T>::ld(p + j), x1 = DT<T>::ld(p + j + half);
    DT<T>::st(p + j, DT<T>::rnd(x0 * cs) + DT<T>::rnd(-x1 * sn));
    DT<T>::st(p + j + half, DT<T>::rnd(x1 * cs) + DT<T>::rnd(x0 * sn));
}

// causal sliding-window attention, one wave per (query, head); head_dim <= 128; fp32 math, one rounding
template <typename T>
__global__ __launch_bounds__(256) void swa_attn_kernel(const T* qkv, T* out, int Tn, int NH, int HD, int window, float scale) {
    const int q = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, lane = threadIdx.x & 63;
    if (q >= Tn) return;
    const int QD = NH * HD;
    const T* qp = qkv + (size_t)q * 3 * QD + (size_t)h * HD;
    const float q0 = lane < HD ? DT<T>::ld(qp + lane) : 0.f;
    const float q1 = lane + 64 < HD ? DT<T>::ld(qp + lane + 64) : 0.f;
    float m = -1e30f, l = 0.f, o0 = 0.f, o1 = 0.f;
    const int k_lo = max(0, q - window + 1);
    for (int k = k_lo; k <= q; ++k) {
        const T* kp = qkv + (size_t)k * 3 * QD + QD + (size_t)h * HD;
        const T* vp = kp + QD;
        float s = (lane < HD ? q0 * DT<T>::ld(kp + lane) : 0.f) + (lane + 64 < HD ? q1 * DT<T>::ld(kp + lane + 64) : 0.f);
        s = wave_sum(s) * scale;
        const float mn = fmaxf(m, s), al = __expf(m - mn), p = __expf(s - mn);
        l = l * al + p;
        o0 = o0 * al + (lane < HD ? p * DT<T>::ld(vp + lane) : 0.f);
        o1 = o1 * al + (lane + 64 < HD ? p * DT<T>::ld(vp + lane + 64) : 0.f);
        m = mn;
    }
    T* op = out + (size_t)q * QD + (size_t)h * HD;
    if (lane < HD) DT<T>::st(op + lane, o0 / l);
    if (lane + 64 < HD) DT<T>::st(op + lane + 64, o1 / l);
}

// final causal conv k=7, C -> 1, + clamp to [-1, 1]; fp32 PCM out.  One wave per 64 output samples.
template <typename T>
__global__ __launch_bounds__(256) void final_conv_kernel(const T* x, const T* w /*[7][C]*/, const T* b, float* pcm, int rows, int C) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= rows) return;
    float acc = 0.f;
    for (int k = 0; k < 7; ++k) {
        const int tt = t - 6 + k;
        if (tt < 0) continue;
        const T* xr = x + (size_t)tt * C;
        for (int c = 0; c < C; ++c) acc = fmaf(DT<T>::ld(w + k * C + c), DT<T>::ld(xr + c), acc);
    }
    float v = DT<T>::rnd(acc + DT<T>::ld(b));
    pcm[t] = fminf(1.f, fmaxf(-1.f, v));
}

}  // namespace fq3
