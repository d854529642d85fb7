// PROTOTYPE (harness only, not linked into libfq3hip.so): M = B token GEMV for batched decode -- B utterances in
// lock-step share one pass over the weights.  Same rounding points as gemv_kernel.
// Differences from the M = 1 kernel: the B input vectors are prepared (RMSNorm, T-rounded) cooperatively and staged
// in LDS as T[B][K] (16-48 KB), the weight rows stay in registers (unpacked once), every wave then walks the B tokens.
#pragma once
#include "../../faster-qwen3-tts_amd/csrc/decode_kernels.cuh"

namespace fq3 {

struct BatchGemvArgs {
    const void* W; int N; int K;
    const void* x; int x_stride;          // T[B][x_stride]
    const void* norm_w; float eps;
    void* y; int y_stride;                // T[B][y_stride]
    const void* res; int res_stride;      // T[B][res_stride] (EPI_RESIDUAL)
    int up_off;
};

template <typename T, int NCH, int PRO, int EPI, bool NT, int B, int R>
__global__ __launch_bounds__(256) void gemv_batch_kernel(BatchGemvArgs a) {
    constexpr int NR = (EPI == EPI_SWIGLU) ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);                     // [B][K], already normalised + rounded
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = a.K;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int row0 = (blockIdx.x * 4 + wave) * R;
    constexpr int TPW = (B + 3) / 4;                            // tokens prepared per wave

    // ---- 1. input-side loads: wave w prepares tokens w, w + 4, ... ----
    Raw8<T> xraw[TPW][NCH], nraw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int m = wave + 4 * t < B ? wave + 4 * t : B - 1;
            ldraw<false>(xraw[t][j], reinterpret_cast<const T*>(a.x) + (size_t)m * a.x_stride + offc);
        }
        if constexpr (PRO == PRO_NORM) ldraw<false>(nraw[j], reinterpret_cast<const T*>(a.norm_w) + offc);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. weight rows ----
    Raw8<T> raw[NR][R][NCH];
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = row0 + r < a.N ? row0 + r : a.N - 1;
            const T* wr = W + (size_t)(row + h * a.up_off) * K;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int off = j * 512 + lane * 8;
                ldraw<NT>(raw[h][r][j], wr + (off < K ? off : 0));
            }
        }
    float resv[B][R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r < a.N ? row0 + r : a.N - 1;
#pragma unroll
        for (int m = 0; m < B; ++m) {
            resv[m][r] = 0.f;
            if constexpr (EPI == EPI_RESIDUAL) resv[m][r] = DT<T>::ld(reinterpret_cast<const T*>(a.res) + (size_t)m * a.res_stride + row);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3. prepare the tokens (weights in flight), stage them in LDS ----
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int m = wave + 4 * t;
        float xr[NCH][8];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (j * 512 + lane * 8 >= K) zero(xraw[t][j]);
            unpack(xraw[t][j], xr[j]);
        }
        if constexpr (PRO == PRO_NORM) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) ss = fmaf(xr[j][i], xr[j][i], ss);
            ss = wave_sum(ss);
            const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                float nw[8];
                unpack(nraw[j], nw);
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    float u = xr[j][i] * rs, v = xr[j][i + 1] * rs;
                    DT<T>::rnd2(u, v);
                    u *= nw[i]; v *= nw[i + 1];
                    DT<T>::rnd2(u, v);
                    xr[j][i] = u; xr[j][i + 1] = v;
                }
            }
        }
        if (m < B) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int off = j * 512 + lane * 8;
                if (off < K) DT<T>::st8(xs + (size_t)m * K + off, xr[j]);
            }
        }
    }
    __syncthreads();
    // ---- 4. weights -> fp32 once; walk the B tokens ----
    float wf[NR][R][NCH][8];
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < NCH; ++j) unpack(raw[h][r][j], wf[h][r][j]);
    float acc[B][R][NR];
#pragma unroll
    for (int m = 0; m < B; ++m) {
        float xr[NCH][8];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
            Raw8<T> q;
            ldraw<false>(q, xs + (size_t)m * K + offc);           // LDS read (generic address space resolves to ds_read_b128)
            if (off >= K) zero(q);
            unpack(q, xr[j]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int h = 0; h < NR; ++h) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) s = fmaf(wf[h][r][j][i], xr[j][i], s);
                acc[m][r][h] = s;
            }
    }
#pragma unroll
    for (int m = 0; m < B; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int h = 0; h < NR; ++h) acc[m][r][h] = wave_sum(acc[m][r][h]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
#pragma unroll
        for (int m = 0; m < B; ++m) {
            float v;
            if constexpr (EPI == EPI_SWIGLU) {
                const float g = DT<T>::rnd(acc[m][r][0]);
                const float u = DT<T>::rnd(acc[m][r][NR - 1]);
                const float sg = DT<T>::rnd(g / (1.0f + expf(-g)));
                v = sg * u;
            } else {
                v = DT<T>::rnd(acc[m][r][0]);
                if constexpr (EPI == EPI_RESIDUAL) v = v + resv[m][r];
            }
            if (lane == 0 && row < a.N) DT<T>::st(reinterpret_cast<T*>(a.y) + (size_t)m * a.y_stride + row, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// PROTOTYPE 2: the same M = B GEMV on the matrix cores.  One workgroup = one tile of 16 weight rows (x NR for
// SwiGLU); the 4 waves split K; v_mfma_f32_16x16x32_bf16 with A = weights (lane: row = lane & 15, k group = lane >> 4,
// 16 bytes straight from global memory -- the MFMA operand layout IS a 16-byte-per-lane global load, no LDS hop),
// B = the B prepared tokens from LDS (lane: token = lane & 15, same k group), C[row = (lane >> 4) * 4 + reg][token].
// Layouts as verified in csrc/codec_kernels.cuh (conv_gemm_kernel).  bf16 only.
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 pbf16x8 __attribute__((ext_vector_type(8)));
typedef float pf32x4 __attribute__((ext_vector_type(4)));

template <int KSTEPS, int PRO, int EPI, bool NT, int B>       // KSTEPS = K / 128 (32-wide k steps per wave)
__global__ __launch_bounds__(256) void gemv_batch_mfma_kernel(BatchGemvArgs a) {
    typedef bf16_t T;
    constexpr int NR = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int K = KSTEPS * 128, KP = K + 8;                 // padded LDS row: 16 tokens x same k would share a bank
    constexpr int NCH = (K + 511) / 512;
    constexpr int TPW = (B + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);                     // [B][KP]
    float* red = reinterpret_cast<float*>(smem_raw + (size_t)B * KP * sizeof(T));      // [4 waves][NR][64 lanes][4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int row0 = blockIdx.x * 16;

    // ---- 1. token loads (wave w prepares tokens w, w + 4, ...) ----
    Raw8<T> xraw[TPW][NCH], nraw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const int m = wave + 4 * t < B ? wave + 4 * t : B - 1;
            ldraw<false>(xraw[t][j], reinterpret_cast<const T*>(a.x) + (size_t)m * a.x_stride + offc);
        }
        if constexpr (PRO == PRO_NORM) ldraw<false>(nraw[j], reinterpret_cast<const T*>(a.norm_w) + offc);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. this wave's quarter of the 16 (x NR) weight rows, already in MFMA A-operand layout ----
    Raw8<T> wreg[NR][KSTEPS];
    const int rowc = row0 + fr < a.N ? row0 + fr : a.N - 1;
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s)
            ldraw<NT>(wreg[h][s], W + (size_t)(rowc + h * a.up_off) * K + wave * (K / 4) + s * 32 + fq * 8);
    // residual for the epilogue lanes: token = fr, rows (fq * 4 .. + 3)
    float resv[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (EPI == EPI_RESIDUAL) {
        const int tok = fr < B ? fr : 0;
        const T* rp = reinterpret_cast<const T*>(a.res) + (size_t)tok * a.res_stride + row0 + fq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) resv[i] = DT<T>::ld(rp + (row0 + fq * 4 + i < a.N ? i : 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3. prepare tokens while the weights fly ----
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int m = wave + 4 * t;
        float xr[NCH][8];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            if (j * 512 + lane * 8 >= K) zero(xraw[t][j]);
            unpack(xraw[t][j], xr[j]);
        }
        if constexpr (PRO == PRO_NORM) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) ss = fmaf(xr[j][i], xr[j][i], ss);
            ss = wave_sum(ss);
            const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                float nw[8];
                unpack(nraw[j], nw);
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    float u = xr[j][i] * rs, v = xr[j][i + 1] * rs;
                    DT<T>::rnd2(u, v);
                    u *= nw[i]; v *= nw[i + 1];
                    DT<T>::rnd2(u, v);
                    xr[j][i] = u; xr[j][i + 1] = v;
                }
            }
        }
        if (m < B) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int off = j * 512 + lane * 8;
                if (off < K) DT<T>::st8(xs + (size_t)m * KP + off, xr[j]);
            }
        }
    }
    __syncthreads();
    // ---- 4. MFMA over this wave's K quarter ----
    pf32x4 acc[NR];
#pragma unroll
    for (int h = 0; h < NR; ++h) acc[h] = pf32x4{0.f, 0.f, 0.f, 0.f};
    const int tokc = fr < B ? fr : 0;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) {
        u32x4 bq = *reinterpret_cast<const u32x4*>(xs + (size_t)tokc * KP + wave * (K / 4) + s * 32 + fq * 8);
        if (fr >= B) bq = u32x4{0u, 0u, 0u, 0u};
        const pbf16x8 bfrag = __builtin_bit_cast(pbf16x8, bq);
#pragma unroll
        for (int h = 0; h < NR; ++h)
            acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(pbf16x8, wreg[h][s].v), bfrag, acc[h], 0, 0, 0);
    }
    // ---- 5. sum the four K quarters, epilogue on wave 0 ----
#pragma unroll
    for (int h = 0; h < NR; ++h) *reinterpret_cast<pf32x4*>(red + ((size_t)(wave * NR + h) * 64 + lane) * 4) = acc[h];
    __syncthreads();
    if (wave != 0) return;
    float tot[NR][4];
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        pf32x4 t = *reinterpret_cast<const pf32x4*>(red + ((size_t)(0 * NR + h) * 64 + lane) * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const pf32x4*>(red + ((size_t)(w * NR + h) * 64 + lane) * 4);
        tot[h][0] = t.x; tot[h][1] = t.y; tot[h][2] = t.z; tot[h][3] = t.w;
    }
    if (fr >= B) return;
    T* yp = reinterpret_cast<T*>(a.y) + (size_t)fr * a.y_stride + row0 + fq * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v;
        if constexpr (EPI == EPI_SWIGLU) {
            const float g = DT<T>::rnd(tot[0][i]);
            const float u = DT<T>::rnd(tot[NR - 1][i]);
            const float sg = DT<T>::rnd(g / (1.0f + expf(-g)));
            v = sg * u;
        } else {
            v = DT<T>::rnd(tot[0][i]);
            if constexpr (EPI == EPI_RESIDUAL) v = v + resv[i];
        }
        if (row0 + fq * 4 + i < a.N) DT<T>::st(yp + i, v);
    }
}

}  // namespace fq3
