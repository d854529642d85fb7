// Kernels of the 12 Hz RVQ codec decoder (vocoder).  Activations are channels-last [time][C] in the
// context dtype T, so every dense contraction (1x1 / k7 dilated / transposed convs, Linear layers) is
// ONE implicit-GEMM kernel on the matrix cores: A rows are time rows gathered at per-tap offsets,
// B is the pre-packed weight [N][taps*Cin].  bf16 uses v_mfma_f32_16x16x32_bf16, fp32 (parity mode)
// uses the exact v_mfma_f32_16x16x4_f32.  Everything else (RVQ gather, norms, depthwise conv, RoPE,
// sliding-window attention, output conv) is HBM/L2-bound row work.
//
// Round-2 structure (codec):
//   * every op works on a ROW RANGE [m_lo, M): the decoder is causal with a finite receptive field after the
//     transformer, so a streaming chunk only recomputes the rows its new samples depend on (fq3_codec.hip plans the
//     ranges); a row's arithmetic does not depend on where the range starts, so a tail decode is bit-identical to the
//     tail of a full decode;
//   * SnakeBeta is an EPILOGUE of the producing GEMM (second output), computed once per element from per-channel
//     constants prepared at bind time, instead of a separate elementwise pass in front of every conv;
//   * SwiGLU is an epilogue of the [gate|up] GEMM (weights interleaved in 16-row groups at pack time);
//   * the GEMM is tile-shape templated (128x128 ... 64x32) with double-buffered LDS (one barrier per K step) and the
//     host picks the shape per layer so that skinny or narrow layers still fill the chip.
// Rounding points follow the Torch module execution of the sibling implementation
// (transformers modeling_qwen3_omni_moe.py:3180-3263, :3542-3696): one rounding to T per op output.
#pragma once
#include "fq3_common.cuh"
#include "skinny_gemm.cuh"

namespace fq3 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kMaxTaps = 8;
struct GemmArgs {
    const void* A; int lda; int M; int a_rows;        // rows outside [0, a_rows) read as zeros (causal left pad)
    int m_lo;                                         // rows [m_lo, M) are computed
    int n_taps; int tap_off[kMaxTaps]; int Cin;       // K = n_taps * Cin; A row of (m, tap) = m + tap_off[tap]
    const void* W; int N;                             // packed weight [N][K]
    const void* bias; int bias_mod;                   // bias[n % bias_mod] (bias_mod = Cout for transposed convs)
    const void* scale;                                // optional per-n multiplier after the activation
    const void* res; int ldr;                         // optional residual [M][ldr]
    void* Y; int ldy;                                 // primary output (may be null when only Y2 is wanted)
    int act;                                          // 0 none, 1 exact GELU, 2 SwiGLU over 16-column [gate|up] groups, 3 SiLU,
                                                      // 4 ReLU, 5 ELU, 6 tanh(ReLU), 7 sigmoid, 8 log(max(v, 1e-5)) (reference-audio analysis)
    const void* sn_a; const void* sn_ib; void* Y2;    // optional second output: SnakeBeta(stored value), channel n % bias_mod
    int act2;                                         // second output's function: 0 SnakeBeta (sn_a / sn_ib), 1 ELU
    // split-K (single-tap GEMMs with few rows, e.g. the 200-token prefill's o_proj / down): workgroup z multiplies channel
    // slice z and stores fp32 partials to ws[z][m - m_lo][n]; splitk_reduce_kernel sums them in order and runs the epilogue
    float* ws; long ws_floats; int ksplit;
    // batched decode: n_seg > 1 independent problems of identical shape in one launch (grid.z / tile range = segment): segment g
    // reads A + g * a_seg, writes Y / Y2 + g * y_seg, residual res + g * r_seg (strides in ELEMENTS of the respective tensor);
    // rows, taps and zero padding are local to a segment, so utterance g never sees utterance g - 1's rows
    int n_seg; long a_seg, y_seg, r_seg;
    int glds_min_wgs;                                 // 0 = default: the LDS-DMA 128 x 64 tile from this many workgroups on (measurement hook of gemm_bench)
    int big_pair;                                     // 256-wide ring tiles: 0 = whole cache lines per row (two K steps per copy group: PAIR, round 6) where Cin % 64 == 0,
                                                      // -1 = the half-line ring of four K steps (measurement switch)
    const void* Wp;                                   // fragment-major copy of W for the weight-stationary kernel (skinny_gemm.cuh), or null
    const void* Wi;                                   // [gate | up] weights only: row-major copy with the halves interleaved in 16-row blocks (fq3_ctx.h kind 2), or null:
                                                      // gemm_swiglu_halves runs the 256-wide ring tile over it with SwiGLU in the epilogue (round 6)
    int no_skinny;                                    // measurement switch: 1 = keep the tiled / split-K kernels where skinny_gemm.cuh would serve
    int glds_cap8;                                    // measurement hook: bf16 x 2 outputs take the eight-wave LDS-DMA tiles up to this many 128 x 64 tiles (0 = default)
    int chain;                                        // 0 = the chain kernel for a handful of output tiles against a long K (round 6), -1 = never (measurement switch)
    int glds_waves;                                   // 0 = the LDS-DMA tile by eight waves where the grid is at most ~one round of tiles (round 6), -1 = always four (measurement switch)
    int xcd_map;                                      // 0 = XCD-aware tile order of the (N tiles x M tiles) grids (round 6), -1 = the plain blockIdx order (measurement switch)
    int epi_legacy;                                   // measurement switch: 1 = the register-layout epilogue (32-byte runs per row) instead of the
                                                      // LDS-parked one (whole tile rows per store instruction); 0 in the product
};

// XCD-aware tile order of a (gx = N tiles) x (gy = M tiles) grid (round 6).  The dispatcher deals workgroups to the 8 XCDs round robin
// in linear order (x fastest), so with the plain order the N tiles of one M tile -- which read the same A rows -- sit on different
// XCDs and every A line crosses the fabric once per XCD that asks: M = 2000, N = 1024, K = 2048 on 128 x 64 tiles moved 8 x 8 MB of A
// through 4-MB L2s for 12 MB of operands, at the SAME 0.9 us per K step whatever the ring depth (profiles/r06_glds_depth.txt).  Here
// XCD x takes a contiguous range of tile ids (bijective also when the count is not a multiple of 8) and ids walk PANELS of four M
// tiles, M fastest: the ~32 workgroups an XCD runs at a time cover 4 M tiles x 8 N tiles -- every A line serves 8 CUs, every W line 4,
// and the operands of the range fit the L2.  Pure relabelling of which workgroup computes which tile: results are unchanged.
__device__ __forceinline__ void xcd_tile(int bx, int by, int gx, int gy, bool plain, int& tn, int& tm) {
    if (plain) { tn = bx; tm = by; return; }
    const int orig = bx + gx * by, total = gx * gy;
    const int xcd = orig & 7, q = total >> 3, r8 = total & 7;
    const int id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    constexpr int PH = 4;
    const int p = id / (PH * gx), rem = id - p * (PH * gx);
    const int left = gy - p * PH, h = left < PH ? left : PH;
    tn = rem / h;
    tm = p * PH + (rem - tn * h);
}

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float elu1(float x) { return x > 0.f ? x : expm1f(x); }
// the activations only the reference-audio analysers use (act >= 4), kept out of line of the codec's epilogue
__device__ __forceinline__ float act_extra(int act, float v) {
    switch (act) {
        case 4: return fmaxf(v, 0.f);
        case 5: return elu1(v);
        case 6: return tanhf(fmaxf(v, 0.f));
        case 7: return 1.0f / (1.0f + expf(-v));
        default: return logf(fmaxf(v, 1e-5f));
    }
}

// SnakeBeta on an already T-rounded value, each Torch op rounded to T (modeling :3566-3580):
// x + (1 / (exp(beta) + 1e-9)) * sin(x * exp(alpha))^2; a = rnd(exp(alpha)), ib = rnd(1 / rnd(rnd(exp(beta)) + 1e-9))
// The sine: fp32 contexts use the accurate sinf (the fp32 waveform is compared with the oracle at 1e-6); in bf16 the value is rounded
// to 8 mantissa bits straight away, so the hardware sine (v_sin_f32 on x / 2 pi, absolute error ~1e-6 at the arguments of a
// vocoder, |x a| up to a few hundred) gives the same bf16 value except within ~1e-6 of a rounding boundary (about one element in
// 3000, one bf16 ulp of the sine) -- at ~40 fewer VALU instructions per element.  The SnakeBeta epilogues evaluate it for
// 8 x 10^8 elements per 300 decoded frames: it was a quarter of the decoder's time.
template <typename T>
__device__ __forceinline__ float snake_sin(float x) {
    if constexpr (sizeof(T) == 2) return __sinf(x);
    else return sinf(x);                 // fp32 and the bf16 x 2 mode: both are compared with the fp32 oracle
}
template <typename T>
__device__ __forceinline__ float snake_apply(float v, float a, float ib) {
    const float s = DT<T>::rnd(snake_sin<T>(DT<T>::rnd(v * a)));
    return v + DT<T>::rnd(ib * DT<T>::rnd(s * s));
}
template <typename T>
__global__ void snake_consts_kernel(const T* alpha, const T* beta, T* a_out, T* ib_out, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float a = DT<T>::rnd(expf(DT<T>::ld(alpha + c)));
    const float b = DT<T>::rnd(expf(DT<T>::ld(beta + c)));
    DT<T>::st(a_out + c, a);
    DT<T>::st(ib_out + c, DT<T>::rnd(1.0f / DT<T>::rnd(b + 1e-9f)));
}

// Epilogue shared by the GEMM kernels.  C/D layout of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg.
template <typename T, int BM, int BN, int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x4_t (&acc)[TM][TN], int m0, int n0, int wr, int wc, int lane) {
    if (a.ksplit > 1) {
        float* ws = a.ws + (size_t)blockIdx.z * (size_t)(a.M - a.m_lo) * a.N;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wc * (BN / 2) + j * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wr * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
                    if (n < a.N && m < a.M) ws[(size_t)(m - a.m_lo) * a.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    T* Y = reinterpret_cast<T*>(a.Y);
    T* Y2 = reinterpret_cast<T*>(a.Y2);
    if (a.act == 2) {
        // SwiGLU: tile j even = gate columns, j odd = the matching up columns (16-row interleave of the packed weight)
        if constexpr (TN >= 2) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j + 1 < TN; j += 2) {       // (an odd TN -- the 96-wide tile -- never runs SwiGLU; never index acc[i][TN])
                    const int np = n0 + wc * (BN / 2) + j * 16;             // physical column of the gate tile
                    if (np >= a.N) continue;
                    const int no = np / 2 + (lane & 15);                    // logical output column
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = m0 + wr * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
                        if (m >= a.M) continue;
                        const float g = DT<T>::rnd(acc[i][j][r]), u = DT<T>::rnd(acc[i][j + 1][r]);
                        DT<T>::st(Y + (size_t)m * a.ldy + no, DT<T>::rnd(g / (1.0f + expf(-g))) * u);
                    }
                }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wc * (BN / 2) + j * 16 + (lane & 15);
            if (n >= a.N) continue;
            const int ch = n % a.bias_mod;
            const float b = a.bias ? DT<T>::ld(reinterpret_cast<const T*>(a.bias) + ch) : 0.f;
            const float sc = a.scale ? DT<T>::ld(reinterpret_cast<const T*>(a.scale) + n) : 1.f;
            float sa = 0.f, sib = 0.f;
            if (Y2 && a.act2 == 0) { sa = DT<T>::ld(reinterpret_cast<const T*>(a.sn_a) + ch); sib = DT<T>::ld(reinterpret_cast<const T*>(a.sn_ib) + ch); }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wr * (BM / 2) + i * 16 + (lane >> 4) * 4 + r;
                if (m >= a.M) continue;
                float v = DT<T>::rnd(acc[i][j][r] + b);
                if (a.act == 1) v = DT<T>::rnd(gelu_exact(v));
                if (a.act == 3) v = DT<T>::rnd(v / (1.0f + expf(-v)));
                if (a.act >= 4) v = DT<T>::rnd(act_extra(a.act, v));
                if (a.scale) v = DT<T>::rnd(sc * v);
                if (a.res) v = v + DT<T>::ld(reinterpret_cast<const T*>(a.res) + (size_t)m * a.ldr + n);
                v = DT<T>::rnd(v);
                if (Y) DT<T>::st(Y + (size_t)m * a.ldy + n, v);
                if (Y2) DT<T>::st(Y2 + (size_t)m * a.ldy + n, a.act2 ? DT<T>::rnd(elu1(v)) : snake_apply<T>(v, sa, sib));
            }
        }
}

// The same per-element epilogue (bias, activation, scale, residual, one rounding, second output) for a kernel that walks its
// outputs from LDS instead of from the MFMA register layout (big_gemm_kernel).  Not for act == 2 (SwiGLU pairs columns).
template <typename T>
__device__ __forceinline__ void epilogue_elem(const GemmArgs& a, float accv, int m, int n) {
    const int ch = n % a.bias_mod;
    const float b = a.bias ? DT<T>::ld(reinterpret_cast<const T*>(a.bias) + ch) : 0.f;
    float v = DT<T>::rnd(accv + b);
    if (a.act == 1) v = DT<T>::rnd(gelu_exact(v));
    if (a.act == 3) v = DT<T>::rnd(v / (1.0f + expf(-v)));
    if (a.act >= 4) v = DT<T>::rnd(act_extra(a.act, v));
    if (a.scale) v = DT<T>::rnd(DT<T>::ld(reinterpret_cast<const T*>(a.scale) + n) * v);
    if (a.res) v = v + DT<T>::ld(reinterpret_cast<const T*>(a.res) + (size_t)m * a.ldr + n);
    v = DT<T>::rnd(v);
    if (a.Y) DT<T>::st(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + n, v);
    if (a.Y2) {
        float y2;
        if (a.act2) y2 = DT<T>::rnd(elu1(v));
        else y2 = snake_apply<T>(v, DT<T>::ld(reinterpret_cast<const T*>(a.sn_a) + ch), DT<T>::ld(reinterpret_cast<const T*>(a.sn_ib) + ch));
        DT<T>::st(reinterpret_cast<T*>(a.Y2) + (size_t)m * a.ldy + n, y2);
    }
}

// Four consecutive columns (n .. n + 3, n % 4 == 0) of output row m: the same per-element arithmetic as gemm_epilogue, with
// 8-byte (bf16) / 16-byte (fp32) accesses.  Caller guarantees m < M, n + 3 < N and 4-element alignment of ldy / ldr / bias_mod.
struct EpiQ {
    const void* bias; int bias_mod; const void* scale; const void* res; int ldr; void* Y; int ldy; void* Y2;
    const void* sn_a; const void* sn_ib; int act, act2;
};
__device__ __forceinline__ EpiQ epiq_of(const GemmArgs& a) {
    return EpiQ{a.bias, a.bias_mod, a.scale, a.res, a.ldr, a.Y, a.ldy, a.Y2, a.sn_a, a.sn_ib, a.act, a.act2};
}
template <typename T>
__device__ __forceinline__ void epi_quad(const EpiQ& a, f32x4_t acc, int m, int n) {
    const int ch = n % a.bias_mod;
    // four consecutive values of a per-column vector / of a row, one 8-byte (bf16) or 16-byte (fp32) load each
    auto ld4 = [](const void* base, size_t off, float (&f)[4]) {
        const T* p = reinterpret_cast<const T*>(base) + off;
        if constexpr (sizeof(T) == 2) {
            const uint2 rr = *reinterpret_cast<const uint2*>(p);
            f[0] = __uint_as_float(rr.x << 16); f[1] = __uint_as_float(rr.x & 0xFFFF0000u);
            f[2] = __uint_as_float(rr.y << 16); f[3] = __uint_as_float(rr.y & 0xFFFF0000u);
        } else if constexpr (std::is_same<T, bfs_t>::value) {
            const u32x4 rr = *reinterpret_cast<const u32x4*>(p);
            f[0] = bfs_to_f(rr.x); f[1] = bfs_to_f(rr.y); f[2] = bfs_to_f(rr.z); f[3] = bfs_to_f(rr.w);
        } else {
            const f32x4_t rr = *reinterpret_cast<const f32x4_t*>(p);
            f[0] = rr[0]; f[1] = rr[1]; f[2] = rr[2]; f[3] = rr[3];
        }
    };
    float rv[4] = {0.f, 0.f, 0.f, 0.f}, bv[4] = {0.f, 0.f, 0.f, 0.f}, sv[4] = {1.f, 1.f, 1.f, 1.f}, sa[4] = {0.f, 0.f, 0.f, 0.f}, sib[4] = {0.f, 0.f, 0.f, 0.f}, v[4], v2[4];
    if (a.res) ld4(a.res, (size_t)m * a.ldr + n, rv);
    if (a.bias) ld4(a.bias, ch, bv);
    if (a.scale) ld4(a.scale, n, sv);
    if (a.Y2 && !a.act2) { ld4(a.sn_a, ch, sa); ld4(a.sn_ib, ch, sib); }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float x = DT<T>::rnd(acc[c] + bv[c]);
        if (a.act == 1) x = DT<T>::rnd(gelu_exact(x));
        if (a.act == 3) x = DT<T>::rnd(x / (1.0f + expf(-x)));
        if (a.act >= 4) x = DT<T>::rnd(act_extra(a.act, x));
        if (a.scale) x = DT<T>::rnd(sv[c] * x);
        if (a.res) x = x + rv[c];
        v[c] = DT<T>::rnd(x);
        v2[c] = 0.f;
        if (a.Y2) v2[c] = a.act2 ? DT<T>::rnd(elu1(v[c])) : snake_apply<T>(v[c], sa[c], sib[c]);
    }
    if constexpr (sizeof(T) == 2) {
        if (a.Y) *reinterpret_cast<uint2*>(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + n) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        if (a.Y2) *reinterpret_cast<uint2*>(reinterpret_cast<T*>(a.Y2) + (size_t)m * a.ldy + n) = uint2{pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3])};
    } else if constexpr (std::is_same<T, bfs_t>::value) {
        if (a.Y) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + n) = u32x4{f_to_bfs(v[0]), f_to_bfs(v[1]), f_to_bfs(v[2]), f_to_bfs(v[3])};
        if (a.Y2) *reinterpret_cast<u32x4*>(reinterpret_cast<T*>(a.Y2) + (size_t)m * a.ldy + n) = u32x4{f_to_bfs(v2[0]), f_to_bfs(v2[1]), f_to_bfs(v2[2]), f_to_bfs(v2[3])};
    } else {
        if (a.Y) *reinterpret_cast<f32x4_t*>(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + n) = f32x4_t{v[0], v[1], v[2], v[3]};
        if (a.Y2) *reinterpret_cast<f32x4_t*>(reinterpret_cast<T*>(a.Y2) + (size_t)m * a.ldy + n) = f32x4_t{v2[0], v2[1], v2[2], v2[3]};
    }
}
// the arguments of segment `seg` of a batched launch (TA: element type of A, TY: of Y / Y2 / res)
template <typename TA, typename TY>
__device__ __forceinline__ GemmArgs gemm_segment(const GemmArgs& a, int seg) {
    GemmArgs b = a;
    if (a.n_seg > 1) {
        b.A = reinterpret_cast<const TA*>(a.A) + (size_t)seg * a.a_seg;
        if (a.Y) b.Y = reinterpret_cast<TY*>(a.Y) + (size_t)seg * a.y_seg;
        if (a.Y2) b.Y2 = reinterpret_cast<TY*>(a.Y2) + (size_t)seg * a.y_seg;
        if (a.res) b.res = reinterpret_cast<const TY*>(a.res) + (size_t)seg * a.r_seg;
    }
    return b;
}
// may the LDS-parked epilogue serve this GEMM?  (no split-K partials, no SwiGLU column pairing)
__device__ __forceinline__ bool epi_can_park(const GemmArgs& a) { return !a.epi_legacy && a.ksplit <= 1 && a.act != 2; }
// are 4-column accesses legal for this GEMM's outputs?
__device__ __forceinline__ bool epi_quads_ok(const GemmArgs& a) {
    return (a.N & 3) == 0 && (a.ldy & 3) == 0 && (a.bias_mod & 3) == 0 && (!a.res || (a.ldr & 3) == 0);
}
// Row-contiguous epilogue of the 4-wave (2 x 2) GEMM kernels: the accumulators -- in the MFMA C layout a lane holds one COLUMN of
// four rows, so a store instruction of the register-layout epilogue touches 32 bytes of each of 64 rows -- are parked in LDS
// (the operand stages are dead by now) and walked with four consecutive columns per lane: one instruction stores whole rows
// of the tile (128 B for BN = 64), the residual is read the same way.  PARK_FLOATS = floats of LDS available; when the tile
// does not fit it goes in two halves (the rows of wave row 0, then of wave row 1).  Padded row of BN + 4 floats: the four
// 16-lane groups of a C-layout write are 4 rows apart, 4 x (BN + 4) = 16 mod 64 banks, so the 64 lanes hit 64 distinct banks.
// Same arithmetic per element as gemm_epilogue: bit-identical.
// WCOLS = waves across the tile's columns (2 for the 4-wave kernels, 4 for the 8-wave 256-row tile), NTHREADS = workgroup size.
template <typename T, int BM, int BN, int TM, int TN, int PARK_FLOATS, int WCOLS = 2, int NTHREADS = 256>
__device__ __forceinline__ void gemm_epilogue_parked(const GemmArgs& a, f32x4_t (&acc)[TM][TN], int m0, int n0, int wr, int wc, int lane, float* park) {
    constexpr int LDP = BN + 4;
    constexpr int HALVES = (BM * LDP <= PARK_FLOATS) ? 1 : 2;
    static_assert((BM / HALVES) * LDP <= PARK_FLOATS, "LDS too small to park half a tile");
    constexpr int ROWS = BM / HALVES, QPR = BN / 4;
    const int fr = lane & 15, fq = lane >> 4;
#pragma unroll
    for (int half = 0; half < HALVES; ++half) {
        if (HALVES == 1 || wr == half) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = (HALVES == 1 ? wr * (BM / 2) : 0) + i * 16 + fq * 4 + r;
                        park[row * LDP + wc * (BN / WCOLS) + j * 16 + fr] = acc[i][j][r];
                    }
        }
        __syncthreads();
        const EpiQ e = epiq_of(a);
        const bool quads = epi_quads_ok(a);
        for (int q = threadIdx.x; q < ROWS * QPR; q += NTHREADS) {
            const int row = q / QPR, c4 = (q - row * QPR) * 4;
            const int m = m0 + half * ROWS + row, n = n0 + c4;
            if (m >= a.M || n >= a.N) continue;
            if (quads) epi_quad<T>(e, *reinterpret_cast<const f32x4_t*>(park + row * LDP + c4), m, n);
            else for (int c = 0; c < 4 && n + c < a.N; ++c) epilogue_elem<T>(a, park[row * LDP + c4 + c], m, n + c);      // unaligned leading dimensions
        }
        if (half + 1 < HALVES) __syncthreads();
    }
}

// BM x BN output tile per workgroup, 4 waves as 2 x 2, each wave (BM/2) x (BN/2) = TM x TN MFMA tiles, K step 32,
// double-buffered LDS (one barrier per step) fed by a PF-deep REGISTER prefetch: the global loads of step s + PF are
// issued while step s is multiplied, so PF tiles (not one) are in flight per workgroup.  The streaming chunks make most
// of these GEMMs skinny (M = 30..400 rows against K up to 7168): few workgroups, long K loops, i.e. latency-bound --
// with one tile in flight a step cost a full memory round trip (~0.8 us measured), PF = 8 hides it.
// Every load is unconditional on a clamped address (out-of-range rows / the padded tail steps are zeroed when the tile
// is staged) so that the compiler's s_waitcnt vmcnt(N) stays exact and the loads really stay in flight.
// grid = (ceil(N / BN), ceil((M - m_lo) / BM)).
// the K loop of conv_gemm_kernel: accumulates the BM x BN tile at (m0, n0) into acc; smem holds 2 * (BM + BN) * LD elements.
// Ends with a workgroup barrier: the operand stages are dead afterwards.
// `taps`: the kernel argument itself -- the tap table is indexed dynamically, which must go to the kernarg segment; `a` (a segment's
// view: shifted pointers) is a register copy that is never indexed
template <typename T, int BM, int BN, int PF>
__device__ __forceinline__ void conv_gemm_mainloop(const GemmArgs& a, const GemmArgs& taps, T* smem, f32x4_t (&acc)[BM / 32][BN / 32], int m0, int n0) {
    constexpr int BK = 32;
    constexpr int LD = BK + (sizeof(T) == 2 ? 8 : 4);          // padded LDS row (elements): breaks the 64/128-byte stride
    constexpr int TM = BM / 32, TN = BN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile too small");
    T (*As)[BM * LD] = reinterpret_cast<T (*)[BM * LD]>(smem);
    T (*Bs)[BN * LD] = reinterpret_cast<T (*)[BN * LD]>(smem + 2 * BM * LD);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int ldw = a.n_taps * a.Cin;                   // weight row stride
    const T* A = reinterpret_cast<const T*>(a.A);
    const T* W = reinterpret_cast<const T*>(a.W);
    int Cin = a.Cin;
    if (a.ksplit > 1) { Cin = a.Cin / a.ksplit; A += blockIdx.z * Cin; W += blockIdx.z * Cin; }     // single-tap only: a channel slice
    const int K = a.n_taps * Cin;
    const int nsteps = K / BK;
    constexpr int EPT = 16 / sizeof(T);                 // elements per 16-byte access
    constexpr int TPR = BK / EPT;                       // threads per tile row
    constexpr int RPP = 256 / TPR;                      // rows per pass
    constexpr int NPA = (BM + RPP - 1) / RPP, NPB = (BN + RPP - 1) / RPP;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int lr = tid / TPR, lc = (tid % TPR) * EPT;
    // B rows of this thread (clamped: columns >= N are computed on a duplicate row and never stored)
    const T* wrow[NPB];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
        int n = n0 + lr + p * RPP;
        n = n < a.N ? n : a.N - 1;
        wrow[p] = W + (size_t)n * ldw + lc;
    }
    u32x4 areg[PF][NPA], breg[PF][NPB];
    // issue cursor (step being loaded) and stage cursor (step being written to LDS): (tap, channel offset) walk K in order
    int i_step = 0, i_tap = 0, i_ci = 0, s_step = 0, s_tap = 0, s_ci = 0;
    auto issue = [&](int slot) {                       // loads of step i_step (clamped to the last real step) into `slot`
        const int toff = taps.tap_off[i_tap];
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            int ar = m0 + lr + p * RPP + toff;
            ar = ar < 0 ? 0 : (ar >= a.a_rows ? a.a_rows - 1 : ar);
            areg[slot][p] = *reinterpret_cast<const u32x4*>(A + (size_t)ar * a.lda + i_ci + lc);
        }
#pragma unroll
        for (int p = 0; p < NPB; ++p) breg[slot][p] = *reinterpret_cast<const u32x4*>(wrow[p] + (size_t)i_tap * Cin + i_ci);
        if (i_step + 1 < nsteps) { ++i_step; i_ci += BK; if (i_ci >= Cin) { i_ci = 0; ++i_tap; } }
        else i_step = nsteps;                           // further issues re-read the last tile; staged as zeros
    };
    auto stage = [&](int slot, int buf) {               // slot -> LDS[buf]; rows outside the problem and tail steps become zeros
        const int toff = taps.tap_off[s_tap];
        const bool real = s_step < nsteps;
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            const int r = lr + p * RPP, m = m0 + r, ar = m + toff;
            const bool ok = real && m < a.M && ar >= 0 && ar < a.a_rows;
            if (r < BM) *reinterpret_cast<u32x4*>(&As[buf][r * LD + lc]) = ok ? areg[slot][p] : u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int p = 0; p < NPB; ++p) {
            const int r = lr + p * RPP;
            if (r < BN) *reinterpret_cast<u32x4*>(&Bs[buf][r * LD + lc]) = breg[slot][p];
        }
        ++s_step; s_ci += BK; if (s_ci >= Cin) { s_ci = 0; if (s_tap + 1 < a.n_taps) ++s_tap; }
    };
#pragma unroll
    for (int d = 0; d < PF; ++d) issue(d);
    stage(0, 0);
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    const int ngroups = (nsteps + PF - 1) / PF;         // the K loop runs ngroups * PF steps; the tail steps multiply zeros
    for (int g = 0; g < ngroups; ++g) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int buf = d & 1;                      // PF is even: step parity = d parity
            issue(d);                                   // slot d was staged one step ago: free again, refill with step + PF
            const T* as = As[buf];
            const T* bs = Bs[buf];
            if constexpr (sizeof(T) == 2) {
                bf16x8_t af[TM], bfr[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(as + (wr * (BM / 2) + i * 16 + fr) * LD + fq * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8_t*>(bs + (wc * (BN / 2) + j * 16 + fr) * LD + fq * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int kk = 0; kk < BK; kk += 4) {
                    float af[TM], bfr[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = reinterpret_cast<const float*>(as)[(wr * (BM / 2) + i * 16 + fr) * LD + kk + fq];
#pragma unroll
                    for (int j = 0; j < TN; ++j) bfr[j] = reinterpret_cast<const float*>(bs)[(wc * (BN / 2) + j * 16 + fr) * LD + kk + fq];
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
            }
            stage((d + 1) % PF, buf ^ 1);               // next step's tile (loaded PF - 1 steps ago) -> the other buffer
            __syncthreads();
        }
    }
}

// T = operand type (bf16 | fp32), TE = storage type of the outputs / residual / per-column vectors (T itself, or bfs_t over bf16
// operands: the bf16 x 2 mode, whose A operand is the [rows][2 Cin] bf16 image of a bfs_t tensor).
template <typename T, int BM, int BN, int PF, typename TE = T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((sizeof(T) == 2 && BM == 128 && BN == 96) ? 3 : 1, 8))) void conv_gemm_kernel(GemmArgs a_in) {
    constexpr int BK = 32;
    constexpr int LD = BK + (sizeof(T) == 2 ? 8 : 4);
    constexpr int TM = BM / 32, TN = BN / 32;
    __shared__ __attribute__((aligned(16))) T smem[2 * (BM + BN) * LD];          // operand stages; the epilogue parks the tile here
    const GemmArgs a = gemm_segment<T, TE>(a_in, a_in.n_seg > 1 ? (int)blockIdx.z : 0);      // (blockIdx.z is the K slice of a split-K launch: n_seg <= 1 there)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int tile_n, tile_m;
    xcd_tile((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, a_in.xcd_map < 0, tile_n, tile_m);
    const int m0 = a.m_lo + tile_m * BM, n0 = tile_n * BN;
    const int wr = wave >> 1, wc = wave & 1;
    f32x4_t acc[TM][TN];
    conv_gemm_mainloop<T, BM, BN, PF>(a, a_in, smem, acc, m0, n0);
    constexpr int kParkFloats = (int)(sizeof(T) * 2 * (BM + BN) * LD / sizeof(float));
    if constexpr (BN % 64 != 0 && BN != 32) {
        // the 96-wide tile never serves split-K or SwiGLU GEMMs (gemm_launch), and a third register-layout epilogue copy for
        // its 12 accumulator tiles is more than the unroller takes (the accumulators would move to scratch)
        gemm_epilogue_parked<TE, BM, BN, TM, TN, kParkFloats>(a, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(smem));
    } else {
        if (epi_can_park(a)) gemm_epilogue_parked<TE, BM, BN, TM, TN, kParkFloats>(a, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(smem));
        else gemm_epilogue<TE, BM, BN, TM, TN>(a, acc, m0, n0, wr, wc, lane);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Fused residual unit of the decoder blocks with C = 96 / 192 channels (blocks 4 and 3: 0.7M / 0.24M rows for 370 frames):
//   mid = SnakeBeta_act2(conv1_k7(S) + b1);   h' = (conv2_1x1(mid) + b2) + h;   S' = SnakeBeta_next(h')
// in ONE kernel per 128-row tile: conv1 is conv_gemm_kernel's K loop over a 128 x C tile (BN = C: the whole channel width, so the
// tile holds complete rows of `mid`), its SnakeBeta image goes to LDS as bf16 [128][C + 8] -- exactly the A operand of the 1x1
// conv -- and the second GEMM (K = C: 3 or 6 MFMA steps) reads its weight fragments straight from global memory in operand layout
// (double-buffered per K step).  The `mid` tensor never goes to HBM (one write + one read of rows x C less per unit) and the unit
// is one launch instead of two.  Every value follows the same chain as the two-kernel path (same MFMA step order, same roundings,
// mid rounded to T exactly where it was stored before): bit-identical (tools/microbench/gemm_bench.hip codec).
// c1: the conv1 GemmArgs (A = S, 7 taps, W1, bias b1, sn_a / sn_ib of act2; Y / Y2 unused); c2: the conv2 GemmArgs (W = W2 [C][C],
// bias b2, res = h, Y = h' (may be null), Y2 = S', sn_a / sn_ib of the next activation; A unused).
// ---------------------------------------------------------------------------------------------------------------------
struct ResUnitArgs { GemmArgs c1, c2; };

// TE = storage type of the activations: bf16_t, or bfs_t (the bf16 x 2 mode, round 5): conv1 then walks the [rows][2 C] bf16 image of
// its bfs_t input against the K-duplicated weight (the host doubles lda / Cin as gemm_launch<bfs_t> does), `mid` is parked in LDS as
// 32-bit (hi | lo) words -- i.e. as the [128][2 C] bf16 image the 1x1 conv's A fragments are read from -- and W2 is the K-duplicated
// [C][2 C] weight.  Same chains and roundings as the two-GEMM bf16 x 2 path: bit-identical.  C = 96 only (the 192-channel image of
// `mid` would need 100 KB of static LDS).
template <typename T, int C, typename TE = T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(C == 96 ? 3 : 1, 4))) void resunit_kernel(ResUnitArgs u_in) {
    constexpr bool kSplit = std::is_same<TE, bfs_t>::value;
    constexpr int KM = kSplit ? 2 : 1;                                    // bf16 columns per activation element
    static_assert(sizeof(T) == 2 && (C == 96 || C == 192) && (!kSplit || C == 96), "bf16 operands, 96 or 192 channels (bf16 x 2: 96)");
    ResUnitArgs u;
    u.c1 = gemm_segment<T, TE>(u_in.c1, (int)blockIdx.y);                    // batched decode: one utterance per blockIdx.y
    u.c2 = gemm_segment<T, TE>(u_in.c2, (int)blockIdx.y);
    constexpr int BM = 128, BN = C, BK = 32, LD = BK + 8, TM = BM / 32, TN = BN / 32, KS = KM * C / 32;
    constexpr int MLD = KM * C + 8;                                       // mid tile row (bf16 elements): 16-byte rows, bank-spread
    constexpr int kOperandElems = 2 * (BM + BN) * LD, kMidElems = BM * MLD;
    constexpr int kSmemElems = kOperandElems > kMidElems ? kOperandElems : kMidElems;
    __shared__ __attribute__((aligned(16))) T smem[kSmemElems];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = u.c1.m_lo + blockIdx.x * BM;
    f32x4_t acc[TM][TN];
    conv_gemm_mainloop<T, BM, BN, 2>(u.c1, u_in.c1, smem, acc, m0, 0);   // ends with a barrier: the operand stages are dead
    // ---- mid = SnakeBeta(rnd(acc + b1)) -> LDS, row-major (the C layout holds column fr of rows fq * 4 + r) ----
    if constexpr (kSplit) {
        // bf16 x 2: 48 accurate-sine SnakeBeta evaluations per lane, unrolled over the accumulator registers, are more than the unroller
        // takes (the accumulators went to scratch).  The raw fp32 tile is parked in LDS instead -- row pitch C + 4 floats, which IS the
        // pitch of the [128][2 C + 8] bf16 image -- and converted IN PLACE by a rolled loop, four columns per lane.
        float* park = reinterpret_cast<float*>(smem);
        constexpr int LDP = MLD / 2;
        static_assert(LDP == C + 4, "pitch");
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    park[(wr * (BM / 2) + i * 16 + fq * 4 + r) * LDP + wc * (BN / 2) + j * 16 + fr] = acc[i][j][r];
        __syncthreads();
        const TE* b1 = reinterpret_cast<const TE*>(u.c1.bias);
        const TE* sa = reinterpret_cast<const TE*>(u.c1.sn_a);
        const TE* sib = reinterpret_cast<const TE*>(u.c1.sn_ib);
        for (int q = tid; q < BM * (C / 4); q += 256) {
            const int row = q / (C / 4), c4 = (q - row * (C / 4)) * 4;
            float* pp = park + row * LDP + c4;
            const f32x4_t av4 = *reinterpret_cast<const f32x4_t*>(pp);
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = c4 + e;
                const float bv = b1 ? DT<TE>::ld(b1 + n) : 0.f;
                const float v = DT<TE>::rnd(DT<TE>::rnd(av4[e] + bv));                   // the unfused epilogue rounds twice (idempotent)
                o[e] = f_to_bfs(snake_apply<TE>(v, DT<TE>::ld(sa + n), DT<TE>::ld(sib + n)));
            }
            *reinterpret_cast<u32x4*>(pp) = o;
        }
    } else {
        const TE* b1 = reinterpret_cast<const TE*>(u.c1.bias);
        const TE* sa = reinterpret_cast<const TE*>(u.c1.sn_a);
        const TE* sib = reinterpret_cast<const TE*>(u.c1.sn_ib);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = wc * (BN / 2) + j * 16 + fr;
            const float bv = b1 ? DT<TE>::ld(b1 + n) : 0.f, av = DT<TE>::ld(sa + n), iv = DT<TE>::ld(sib + n);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wr * (BM / 2) + i * 16 + fq * 4 + r;
                    const float v = DT<TE>::rnd(DT<TE>::rnd(acc[i][j][r] + bv));         // the unfused epilogue rounds twice (idempotent)
                    DT<T>::st(smem + row * MLD + n, snake_apply<TE>(v, av, iv));
                }
        }
    }
    __syncthreads();
    // ---- conv2: acc2 = mid (LDS, A operand) x W2^T (B operand from global: row n = fr, k = ks * 32 + fq * 8) ----
    const T* W2 = reinterpret_cast<const T*>(u.c2.W);
    const T* wp = W2 + (size_t)(wc * (BN / 2) + fr) * (KM * C) + fq * 8;
    bf16x8_t bfr[2][TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)j * 16 * (KM * C));
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        if (ks + 1 < KS) {
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[(ks + 1) & 1][j] = *reinterpret_cast<const bf16x8_t*>(wp + (size_t)j * 16 * (KM * C) + (ks + 1) * 32);
        }
        bf16x8_t af[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(smem + (wr * (BM / 2) + i * 16 + fr) * MLD + ks * 32 + fq * 8);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[ks & 1][j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();                                                      // everybody is done reading mid: the epilogue parks there
    constexpr int kParkFloats = (int)(sizeof(T) * kSmemElems / sizeof(float));
    gemm_epilogue_parked<TE, BM, BN, TM, TN, kParkFloats>(u.c2, acc, m0, 0, wr, wc, lane, reinterpret_cast<float*>(smem));
}

// can this (conv1, conv2) pair run as one resunit_kernel launch?  (arguments in the storage type's own units)
template <typename T>
inline bool resunit_ok(const GemmArgs& c1, const GemmArgs& c2) {
    const int C = c1.N;
    const bool shape = std::is_same<T, bfs_t>::value ? C == 96 : (sizeof(T) == 2 && (C == 96 || C == 192));
    return shape && c1.n_taps == 7 && c1.Cin == C && c2.N == C && c2.Cin == C && c2.n_taps == 1 && c1.sn_a &&
           c1.sn_ib && c2.Y2 && c2.sn_a && !c2.act2 && !c1.act && !c2.act && !c1.scale && !c2.scale && !c1.res && c2.res && c2.ldr == C &&
           c2.ldy == C && c2.bias_mod == C && c1.bias_mod == C && c1.lda == C;
}
template <typename T>
inline bool resunit_launch(const GemmArgs& c1, const GemmArgs& c2, hipStream_t s) {
    if constexpr (sizeof(T) != 2 && !std::is_same<T, bfs_t>::value) return false;
    else {
        const int rows = c1.M - c1.m_lo, C = c1.N;
        if (!resunit_ok<T>(c1, c2)) return false;
        if (rows <= 0) return true;
        ResUnitArgs u{c1, c2};
        u.c2.M = c1.M; u.c2.m_lo = c1.m_lo;
        const dim3 grid((rows + 127) / 128, c1.n_seg > 1 ? c1.n_seg : 1);
        if constexpr (std::is_same<T, bfs_t>::value) {
            // the [rows][2 C] bf16 image of the bfs_t input against the K-duplicated weights (what gemm_launch<bfs_t> does for a GEMM)
            u.c1.lda = 2 * c1.lda; u.c1.Cin = 2 * c1.Cin; u.c1.a_seg = 2 * c1.a_seg; u.c1.ws = nullptr;
            u.c2.Cin = 2 * c2.Cin; u.c2.ws = nullptr;
            hipLaunchKernelGGL((resunit_kernel<bf16_t, 96, bfs_t>), grid, dim3(256), 0, s, u);
        } else {
            if (C == 96) hipLaunchKernelGGL((resunit_kernel<T, 96>), grid, dim3(256), 0, s, u);
            else if (C == 192) hipLaunchKernelGGL((resunit_kernel<T, 192>), grid, dim3(256), 0, s, u);
            else return false;
        }
        return true;
    }
}

// Large-M variant (bf16): 128 x 64 tile, K step 64, operands copied global -> LDS by the DMA path (global_load_lds, 16 bytes
// per lane: no staging registers, no ds_write pass), two LDS stages, one barrier per step.  The LDS image of a DMA copy is
// lane-linear (wave-uniform base + lane * 16), so rows are exactly 128 bytes with no padding; bank conflicts are avoided by
// swizzling the SOURCE address: LDS chunk c of row r holds global chunk c ^ ((r >> 1) & 7), which spreads the 16 rows of
// an MFMA fragment read over all 64 banks.  Rows that must read as zeros (causal left pad, rows past M) are pointed at a
// zero page.  The MFMA sequence per output element is the same ascending-K chain of 32-wide products as in
// conv_gemm_kernel, so both kernels produce bit-identical results (tail decodes stay identical to full decodes).
// Requires Cin % 64 == 0 (a K step never straddles a tap).
static __device__ u32x4 g_gemm_zero_page[8];      // 128 zero bytes (one copy per translation unit)

// s_waitcnt immediate that waits for vmcnt <= n only (gfx9 encoding: vmcnt = bits 3:0 and 15:14, expcnt 6:4, lgkmcnt 11:8)
constexpr int vmcnt_imm(int n) { return (n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14); }

// NT = 512 (round 6): the same tile by EIGHT waves as 2 (M) x 4 (N), each a 64 x (BN / 4) block.  A wave takes delivery of ~3.6 B/clk out
// of the L2 however many loads it keeps in flight and whichever way they travel (tools/microbench/dma_rate_bench.hip,
// profiles/r06_dma_rate.txt: one workgroup of 4 / 8 / 16 waves per CU: 35 / 64 / 78 GB/s), so a grid that gives every CU ONE tile -- the
// packed prefill's o_proj / down: M = 2000, N = 1024 -- moves its operands at the rate of four waves per CU and spends 0.9 us per K
// step of 64 whatever the ring depth (profiles/r06_glds_depth.txt).  Eight waves share the same copies (half as many per thread); the
// MFMA chain per output element is unchanged (bit-identical); only the LDS-parked epilogue is served (gemm_launch checks).
// DBG (measurement builds of tools/microbench/gemm_bench.hip only; 0 in the library): 1 = no MFMAs (the fragment reads stay), 2 = no
// fragment reads either (copies + barriers only), 3 = copies only (no barrier): where does a K step's time go?
template <int BN, int STAGES, typename TE = bf16_t, int NT = 256, int BM = 128, int DBG = 0>
__global__ __launch_bounds__(NT) void glds_gemm_kernel(GemmArgs a_in) {
    static_assert(STAGES >= 2 && STAGES <= 6, "two to six LDS stages");
    static_assert(NT == 256 || NT == 512, "four or eight waves");
    typedef bf16_t T;
    const GemmArgs a = gemm_segment<T, TE>(a_in, (int)blockIdx.z);
    constexpr int WCOLS = NT == 256 ? 2 : 4;                                  // waves across the tile's columns (two wave rows of 64 rows each)
    constexpr int BK = 64, TM = BM / 32, TN = BN / (16 * WCOLS);            // BM = 64 (eight waves only): half-height tiles where 128-row tiles would leave most CUs idle
    static_assert(BM == 128 || (BM == 64 && NT == 512), "128-row tiles, or 64-row tiles by eight waves");
    constexpr int NPA = BM * 8 / NT, NPB = BN * 8 / NT;                       // 16-byte copy slots per thread and stage: A tile, B tile
    static_assert(TN >= 1 && NPB >= 1, "tile too narrow for the wave layout");
    extern __shared__ __attribute__((aligned(128))) unsigned char glds_smem[];
    T* As = reinterpret_cast<T*>(glds_smem);                                  // [STAGES][BM * BK]
    T* Bs = As + STAGES * BM * BK;                                            // [STAGES][BN * BK]
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tile_n, tile_m;
    xcd_tile((int)blockIdx.x, (int)blockIdx.y, (int)gridDim.x, (int)gridDim.y, a_in.xcd_map < 0, tile_n, tile_m);
    const int m0 = a.m_lo + tile_m * BM, n0 = tile_n * BN;
    const int wr = NT == 256 ? wave >> 1 : wave >> 2, wc = NT == 256 ? wave & 1 : wave & 3;
    const int K = a.n_taps * a.Cin, nsteps = K / BK;
    const T* A = reinterpret_cast<const T*>(a.A);
    const T* W = reinterpret_cast<const T*>(a.W);
    const T* zero = reinterpret_cast<const T*>(g_gemm_zero_page);
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // this thread's copy slots: slot = p * NT + tid -> row = slot >> 3, LDS chunk = slot & 7, source chunk = chunk ^ swz(row)
    int arow[NPA], asrc[NPA];
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
        const int slot = p * NT + tid, r = slot >> 3, c = slot & 7;
        arow[p] = m0 + r;
        asrc[p] = (c ^ ((r >> 1) & 7)) * 8;
    }
    const T* wsrc[NPB];
#pragma unroll
    for (int p = 0; p < NPB; ++p) {
        const int slot = p * NT + tid, r = slot >> 3, c = slot & 7;
        int n = n0 + r;
        n = n < a.N ? n : a.N - 1;
        wsrc[p] = W + (size_t)n * K + (c ^ ((r >> 1) & 7)) * 8;
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    auto issue = [&](int step, int buf) {
        const int k0 = step * BK;
        const int tap = k0 / a.Cin, ci = k0 - tap * a.Cin;
        const int toff = a_in.tap_off[tap];
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            const int ar = arow[p] + toff;
            const bool ok = arow[p] < a.M && ar >= 0 && ar < a.a_rows;
            const T* src = ok ? A + (size_t)ar * a.lda + ci + asrc[p] : zero;
            __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(As + buf * BM * BK + (p * NT + wave * 64) * 8), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < NPB; ++p)
            __builtin_amdgcn_global_load_lds((gbl_ptr)(wsrc[p] + k0), (lds_ptr)(Bs + buf * BN * BK + (p * NT + wave * 64) * 8), 16, 0, 0);
    };
    issue(0, 0);
#pragma unroll
    for (int d = 1; d < STAGES - 1; ++d)
        if (d < nsteps) issue(d, d);
    int buf = 0;
    constexpr int CPS = NPA + NPB;                       // copies per thread and stage
    for (int s = 0; s < nsteps; ++s) {
        // this wave's copies of stage s have landed; the copies of the (up to STAGES - 2) stages behind it may stay in flight
        const int rem = nsteps - 1 - s;
        if (STAGES >= 6 && rem >= 4) __builtin_amdgcn_s_waitcnt(vmcnt_imm(4 * CPS));
        else if (STAGES >= 5 && rem >= 3) __builtin_amdgcn_s_waitcnt(vmcnt_imm(3 * CPS));
        else if (STAGES >= 4 && rem >= 2) __builtin_amdgcn_s_waitcnt(vmcnt_imm(2 * CPS));
        else if (STAGES >= 3 && rem >= 1) __builtin_amdgcn_s_waitcnt(vmcnt_imm(CPS));
        else __builtin_amdgcn_s_waitcnt(vmcnt_imm(0));
        if constexpr (DBG != 3) __builtin_amdgcn_s_barrier();                    // ... and everybody's; everybody is also done reading the stage refilled next
        if (s + STAGES - 1 < nsteps) issue(s + STAGES - 1, (buf + STAGES - 1) % STAGES);       // flies under this step's MFMAs
        const T* as = As + buf * BM * BK;
        const T* bs = Bs + buf * BN * BK;
#pragma unroll
        for (int ks = 0; ks < (DBG >= 2 ? 0 : 2); ++ks) {
            bf16x8_t af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = wr * (BM / 2) + i * 16 + fr;
                af[i] = *reinterpret_cast<const bf16x8_t*>(as + r * BK + (((ks * 4 + fq) ^ ((r >> 1) & 7)) * 8));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wc * (BN / WCOLS) + j * 16 + fr;
                bfr[j] = *reinterpret_cast<const bf16x8_t*>(bs + r * BK + (((ks * 4 + fq) ^ ((r >> 1) & 7)) * 8));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if constexpr (DBG == 1) { acc[i][j][0] += (float)af[i][0] + (float)bfr[j][0]; }      // (keeps the reads alive)
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
                }
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
    }
    constexpr int kParkFloats = STAGES * (BM + BN) * BK * 2 / 4;
    if (NT == 512 || epi_can_park(a)) {
        __syncthreads();                                 // everybody is done reading the last stage
        gemm_epilogue_parked<TE, BM, BN, TM, TN, kParkFloats, WCOLS, NT>(a, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(glds_smem));
    } else {
        if constexpr (NT == 256) gemm_epilogue<TE, BM, BN, TM, TN>(a, acc, m0, n0, wr, wc, lane);
    }
}

template <int BN, int STAGES, typename TE = bf16_t, int NT = 256, int BM = 128, int DBG = 0>
inline void glds_go(const GemmArgs& a, hipStream_t s) {
    const int rows = a.M - a.m_lo;
    const size_t shm = (size_t)STAGES * (BM + BN) * 64 * 2;
    auto kern = glds_gemm_kernel<BN, STAGES, TE, NT, BM, DBG>;
    if (shm > 48 * 1024) {
        // the LDS limit of a kernel is a per-device setting: raised once per instantiation AND device (a process may drive several)
        static bool raised[16] = {};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) dev = -1;
        if (dev < 0 || !raised[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
            if (dev >= 0) raised[dev] = true;
        }
    }
    hipLaunchKernelGGL(kern, dim3((a.N + BN - 1) / BN, (rows + BM - 1) / BM, a.n_seg > 1 ? a.n_seg : 1), dim3(NT), shm, s, a);
}

// Chain GEMM (round 6): FEW rows against a LONG K -- the convs of a streaming chunk (M = 52 .. 416 rows, K = 5376 .. 14336) -- bit-identical
// to the tile families.  Those GEMMs have a handful of output tiles (19 workgroups on average for a 25 + 8-frame chunk), and splitting K
// over workgroups is not open to them: a tail decode must equal the full decode bit for bit, so every output element keeps ONE ascending
// chain of 32-wide MFMA products.  What bounds them is the rate at which a WAVE takes delivery of its operands (~3.6 B/clk,
// profiles/r06_dma_rate.txt), not the matrix cores.  So: a workgroup of NW waves owns ONE 32 x 32 output tile, the K steps are dealt to the
// waves in chunks of CH (chunk c belongs to wave c % NW), every wave keeps its next chunk's operand fragments in flight -- straight from
// global memory in operand layout, no LDS staging -- and the ACCUMULATORS travel: the owner of chunk c takes the tile's partial sums from
// LDS, appends its CH x 4 products in ascending K order, puts them back, and the workgroup meets at a barrier (lgkmcnt only: the loads stay
// in flight).  NW waves deliver in parallel; the arithmetic is the serial chain it always was (4 MFMAs per K step: ~1 % of the time).
// The LDS image of the accumulators is the parked tile of gemm_epilogue_parked: the epilogue walks it with four columns per lane.
template <int NW, int CH, typename TE = bf16_t>
__global__ __launch_bounds__(NW * 64) void chain_gemm_kernel(GemmArgs a_in) {
    typedef bf16_t T;
    constexpr int BM = 32, BN = 32, TM = 2, TN = 2, LDP = BN + 4, NT = NW * 64;
    __shared__ __attribute__((aligned(16))) float park[BM * LDP];
    const GemmArgs a = gemm_segment<T, TE>(a_in, (int)blockIdx.z);
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m0 = a.m_lo + blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int K = a.n_taps * a.Cin, nsteps = K / 32, nchunks = (nsteps + CH - 1) / CH;
    const T* A = reinterpret_cast<const T*>(a.A);
    const T* zero = reinterpret_cast<const T*>(g_gemm_zero_page) + fq * 8;
    int am[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int m = m0 + i * 16 + fr; am[i] = m < a.M ? m : -(1 << 28); }      // rows past M: out of every tap's range
    const T* wrow[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        int n = n0 + j * 16 + fr;
        n = n < a.N ? n : a.N - 1;                                   // columns past N are computed on row N - 1 and never stored
        wrow[j] = reinterpret_cast<const T*>(a.W) + (size_t)n * K + fq * 8;
    }
    u32x4 ar[CH][TM], wr[CH][TN];
    auto load_chunk = [&](int c) {
#pragma unroll
        for (int s = 0; s < CH; ++s) {
            int t = c * CH + s;
            t = t < nsteps ? t : nsteps - 1;                         // the last chunk's spare steps reload the last step (never multiplied)
            const int k0 = t * 32, tap = k0 / a.Cin, ci = k0 - tap * a.Cin;
            const int toff = a_in.tap_off[tap];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = am[i] + toff;
                const T* src = (r >= 0 && r < a.a_rows) ? A + (size_t)r * a.lda + ci + fq * 8 : zero;
                ar[s][i] = *reinterpret_cast<const u32x4*>(src);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) wr[s][j] = *reinterpret_cast<const u32x4*>(wrow[j] + k0);
        }
    };
    if (wave < nchunks) load_chunk(wave);
    for (int c = 0; c < nchunks; ++c) {
        if (wave == (c % NW)) {
            f32x4_t acc[TM][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = c ? park[(i * 16 + fq * 4 + r) * LDP + j * 16 + fr] : 0.f;
#pragma unroll
            for (int s = 0; s < CH; ++s) {
                if (c * CH + s < nsteps) {
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ar[s][i]), __builtin_bit_cast(bf16x8_t, wr[s][j]), acc[i][j], 0, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) park[(i * 16 + fq * 4 + r) * LDP + j * 16 + fr] = acc[i][j][r];
            if (c + NW < nchunks) load_chunk(c + NW);                // in flight over the next NW - 1 chunks of the chain
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): the partial sums are in LDS (the operand loads stay in flight)
        __builtin_amdgcn_s_barrier();
    }
    const EpiQ e = epiq_of(a);
    const bool quads = epi_quads_ok(a);
    for (int q = tid; q < BM * (BN / 4); q += NT) {
        const int row = q / (BN / 4), c4 = (q - row * (BN / 4)) * 4;
        const int m = m0 + row, n = n0 + c4;
        if (m >= a.M || n >= a.N) continue;
        if (quads) epi_quad<TE>(e, *reinterpret_cast<const f32x4_t*>(park + row * LDP + c4), m, n);
        else for (int cc = 0; cc < 4 && n + cc < a.N; ++cc) epilogue_elem<TE>(a, park[row * LDP + c4 + cc], m, n + cc);
    }
}
template <int NW, int CH, typename TE = bf16_t>
inline void chain_go(const GemmArgs& a, hipStream_t s) {
    const int rows = a.M - a.m_lo;
    hipLaunchKernelGGL((chain_gemm_kernel<NW, CH, TE>), dim3((a.N + 31) / 32, (rows + 31) / 32, a.n_seg > 1 ? a.n_seg : 1), dim3(NW * 64), 0, s, a);
}
// shapes the chain kernel serves: a K step never straddles a tap, the per-element epilogue is the parked one (no split-K, no SwiGLU pairing)
inline bool chain_ok(const GemmArgs& a) {
    return a.Cin % 32 == 0 && a.act != 2 && a.ksplit <= 1 && !a.epi_legacy && a.lda % 8 == 0;
}

// Largest-M variant (bf16): 256 x 256 tile, 8 waves as 2 (M) x 4 (N), each wave a 128 x 64 block of 8 x 4 MFMA tiles
// (one LDS byte feeds 2.7x the arithmetic of the 128 x 64 tile), K step 32, operands by global_load_lds into a RING OF FOUR
// 32 KB stages: the copies of K steps t + 1 .. t + 3 are in flight while step t is multiplied, the wait before a step is
// a counted vmcnt(8) (never 0 in the steady state) and there is ONE barrier per step -- it both publishes stage t and
// proves that stage t - 1 (refilled right after it) has been read by everybody.  One workgroup per CU (128 KB LDS), two
// waves per SIMD.  LDS rows are 64 B (four 16-byte chunks); LDS chunk c of row r holds global chunk c ^ ((-(r >> 2)) & 3),
// which makes the 16-lane groups of a ds_read_b128 fragment read hit 16 distinct bank quads.  blockIdx is remapped so that
// the workgroups of one XCD (every 8th id) cover a compact range of tiles (shared A / W panels stay in that XCD's L2).
// Per output element the same ascending chain of 32-wide MFMA products as conv_gemm_kernel: bit-identical results.
constexpr int kBigBM = 256, kBigBN = 256, kBigBK = 32, kBigStages = 4;
// workgroups from which gemm_launch takes the LDS-DMA 128 x 64 tile (below: the register-prefetch tiles).  512 until round 5; measured in
// round 6 (profiles/r06_gemm_shapes.txt): from ~150 workgroups on the LDS-DMA tile (whole 128-byte lines per row, K steps of 64) is never
// slower than the 64 x 64 register-prefetch tile (half lines) and wins 8-10 % on the packed prefill's o_proj / down (M = 1200-2000, N = 1024)
// and 25 % on the codec's dec.0 (M = 1480, N = 1536, K = 7168).  All tile families agree bit for bit.
constexpr int kGldsMinWgs = 150;
// Batched decode: the grid holds n_seg x (tiles of one segment) workgroups; the XCD-aware remap runs over ALL of them (so that the
// tiles of one utterance that share A / W panels stay on one XCD), then tile / tiles_seg selects the segment.
// PAIR (round 6): the operand copies move WHOLE 128-byte lines -- 8 lanes per row, two K steps (64 columns) per copy group -- into two
// buffers of 64-column rows (the same LDS bytes as the ring of four 32-column stages).  A copy of 16 rows x 64 B (half lines) is served at
// 40 GB/s per CU out of the L2, whole lines at 148 (tools/microbench/l2_rate_bench.hip, profiles/r06_l2_rate_by_pattern.txt): at 32 KB per
// K step that is 0.8 us against 0.43 us of MFMA time for the 256 x 256 tile -- the ring was memory-bound at ~half the matrix-core peak.
// Buffer d & 1 holds K steps 2 d and 2 d + 1; pair d + 1 is copied while pair d is multiplied (one barrier per pair).  LDS chunk c of row
// r holds global chunk c ^ ((r >> 1) & 7) of the row's 128 B: the 16 rows of a fragment read cover all 64 banks.  Per output element the
// MFMA products are issued in the same ascending order over the K steps: bit-identical to the ring of four (and to conv_gemm_kernel).
template <int ST, bool TR, int BN, typename TE = bf16_t, bool PAIR = false>
__global__ __launch_bounds__(512) void big_gemm_kernel(GemmArgs a_in, int tiles_x, int tiles_seg, int tiles_total) {
    typedef bf16_t T;
    static_assert(!TR || std::is_same<TE, bf16_t>::value, "the register-layout (TR) epilogue stores bf16");
    // BN = 256: wave block 128 x 64 (8 x 4 tiles).  BN = 128 (TR only; N = 2048 at M = 4096 would fill half the CUs with 256-wide
    // tiles): wave block 128 x 32 (8 x 2 tiles), 3 copies per thread and step.
    constexpr int BM = kBigBM, BK = kBigBK, TM = 8, TN = BN / 64, WN = BN / 4, NPB = BN / 128, CP = 2 + NPB;
    static_assert(ST == 4, "ring of four stages");
    static_assert(BN == 256 || BN == 128, "256- or 128-wide tiles");
    extern __shared__ __attribute__((aligned(128))) unsigned char big_smem[];
    T* As = reinterpret_cast<T*>(big_smem);                                   // [ST][BM * BK]
    T* Bs = As + ST * BM * BK;                                                // [ST][BN * BK]
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware tile id (bijective also when the tile count is not a multiple of 8)
    const int orig = blockIdx.x, xcd = orig & 7, q = tiles_total >> 3, r8 = tiles_total & 7;
    const int tile_all = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (orig >> 3);
    const int seg = tile_all / tiles_seg, tile = tile_all - seg * tiles_seg;
    const GemmArgs a = gemm_segment<T, TE>(a_in, seg);
    const int m0 = a.m_lo + (tile / tiles_x) * BM, n0 = (tile % tiles_x) * BN;
    const int wr = wave >> 2, wc = wave & 3;
    const int K = a.n_taps * a.Cin, nsteps = K / BK;
    const T* A = reinterpret_cast<const T*>(a.A);
    const T* W = reinterpret_cast<const T*>(a.W);
    const T* zero = reinterpret_cast<const T*>(g_gemm_zero_page);
    f32x4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    // copy slots: slot = p * 512 + tid -> row = slot >> 2, LDS chunk = slot & 3, source chunk = chunk ^ swz(row)
    // (PAIR: row = slot >> 3, chunk = slot & 7 of the row's 128 bytes, swz(row) = (row >> 1) & 7; twice the slots per copy group)
    constexpr int NPA = PAIR ? 4 : 2, NPW = PAIR ? 2 * NPB : NPB;
    int arow[NPA], asrc[NPA];
    const T* wsrc[NPW];
#pragma unroll
    for (int p = 0; p < NPA; ++p) {
        const int slot = p * 512 + tid, r = PAIR ? slot >> 3 : slot >> 2, c = PAIR ? slot & 7 : slot & 3;
        const int sc = PAIR ? (c ^ ((r >> 1) & 7)) * 8 : (c ^ ((0 - (r >> 2)) & 3)) * 8;
        arow[p] = m0 + r;
        asrc[p] = sc;
        if (p < NPW) {
            int n = n0 + r;
            n = n < a.N ? n : a.N - 1;
            wsrc[p] = W + (size_t)n * K + sc;
        }
    }
    typedef __attribute__((address_space(3))) void* lds_ptr;
    typedef const __attribute__((address_space(1))) void* gbl_ptr;
    // (PAIR: `step` counts copy groups of 64 columns, a buffer holds BM x 64 | BN x 64 elements; a group never straddles a tap: Cin % 64 == 0)
    constexpr int GK = PAIR ? 2 * BK : BK;                       // columns per copy group
    auto issue = [&](int step, int buf) {
        const int k0 = step * GK;
        const int tap = k0 / a.Cin, ci = k0 - tap * a.Cin;
        const int toff = a_in.tap_off[tap];
#pragma unroll
        for (int p = 0; p < NPA; ++p) {
            const int ar = arow[p] + toff;
            const bool ok = arow[p] < a.M && ar >= 0 && ar < a.a_rows;
            const T* src = ok ? A + (size_t)ar * a.lda + ci + asrc[p] : zero;
            __builtin_amdgcn_global_load_lds((gbl_ptr)src, (lds_ptr)(As + buf * BM * GK + (p * 512 + wave * 64) * 8), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < NPW; ++p)
            __builtin_amdgcn_global_load_lds((gbl_ptr)(wsrc[p] + k0), (lds_ptr)(Bs + buf * BN * GK + (p * 512 + wave * 64) * 8), 16, 0, 0);
    };
    if constexpr (PAIR) {
        // ---- two buffers of 64-column rows: pair d = K steps 2 d, 2 d + 1 ----
        const int npairs = nsteps >> 1;                              // (K % 64 == 0: the host's condition for this form)
        const int swz8 = (fr >> 1) & 7;
        const int a_off8 = (wr * 128 + fr) * GK, b_off8 = (wc * WN + fr) * GK;
        bf16x8_t fa0[TM], fb0[TN], fa1[TM], fb1[TN];
        auto load_frags8 = [&](bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN], int buf, int sub) {
            const int ch = ((sub * 4 + fq) ^ swz8) * 8;
            const T* as = As + buf * BM * GK + a_off8 + ch;
            const T* bs = Bs + buf * BN * GK + b_off8 + ch;
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bs + j * 16 * GK);
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(as + i * 16 * GK);
        };
        auto mfmas = [&](bf16x8_t (&ca)[TM], bf16x8_t (&cb)[TN]) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
        };
        issue(0, 0);
        for (int d = 0; d < npairs; ++d) {
            __builtin_amdgcn_s_waitcnt(0x0f70);                      // vmcnt(0): my copies of pair d (nothing else is in flight)
            __builtin_amdgcn_s_barrier();                            // pair d is published; everybody is done reading pair d - 1's buffer
            issue(d + 1 < npairs ? d + 1 : npairs - 1, (d + 1) & 1); // (past the end: a redundant re-copy into the free buffer, never read)
            load_frags8(fa0, fb0, d & 1, 0);
            load_frags8(fa1, fb1, d & 1, 1);                         // the second step's fragments are read under the first step's MFMAs
            mfmas(fa0, fb0);
            mfmas(fa1, fb1);
            constexpr int PER = (TM * TN) / (TM + TN);
            __builtin_amdgcn_sched_group_barrier(0x020, NPA + NPW, 0);               // the copies first
            __builtin_amdgcn_sched_group_barrier(0x100, TM + TN, 0);                 // the first step's fragment reads
#pragma unroll
            for (int k = 0; k < TM + TN; ++k) {                                      // the second step's reads between the first step's MFMAs
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 2 * TM * TN - PER * (TM + TN), 0);
        }
        __builtin_amdgcn_s_waitcnt(0x0f70);                          // the tail's redundant copy must not outlive the workgroup's LDS
    } else {
#pragma unroll
    for (int d = 0; d < ST - 1; ++d) issue(d < nsteps ? d : nsteps - 1, d);          // a short K re-copies its last step (never read)
    const int swz = ((0 - (fr >> 2)) & 3);
    const int a_off = (wr * 128 + fr) * BK + ((fq ^ swz) * 8);                       // + i * 16 * BK
    const int b_off = (wc * WN + fr) * BK + ((fq ^ swz) * 8);                        // + j * 16 * BK
    // Fragments are double-buffered in registers: while step s is multiplied from `cur`, the fragments of step s + 1 are
    // read into `nxt`, so no MFMA waits for LDS.  Iteration s: vmcnt(4) = my copies of step s + 1 have landed (only step
    // s + 2's four may stay in flight); the barrier publishes them and proves that everybody's reads of step s - 1 are done
    // (they fed the MFMAs of iteration s - 1); then step s + 3 is copied into the stage step s - 1 used.
    bf16x8_t fa0[TM], fb0[TN], fa1[TM], fb1[TN];
    auto load_frags = [&](bf16x8_t (&fa)[TM], bf16x8_t (&fb)[TN], int stage) {
        const T* as = As + stage * BM * BK + a_off;
        const T* bs = Bs + stage * BN * BK + b_off;
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const bf16x8_t*>(bs + j * 16 * BK);
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(as + i * 16 * BK);
    };
    auto body = [&](bf16x8_t (&ca)[TM], bf16x8_t (&cb)[TN], bf16x8_t (&na)[TM], bf16x8_t (&nb)[TN], int s) {
        if constexpr (CP == 4) __builtin_amdgcn_s_waitcnt(0x0f74); else __builtin_amdgcn_s_waitcnt(0x0f73);      // vmcnt(CP): one step's copies may stay in flight
        __builtin_amdgcn_s_barrier();
        {
            const int nx = s + ST - 1;
            issue(nx < nsteps ? nx : nsteps - 1, (s + ST - 1) & (ST - 1));           // redundant re-copy past the end: never read
        }
        load_frags(na, nb, (s + 1) & (ST - 1));                                      // step s + 1 (garbage after the last step: unused)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = TR ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(cb[j], ca[i], acc[i][j], 0, 0, 0)      // operands swapped: C^T tile
                               : __builtin_amdgcn_mfma_f32_16x16x32_bf16(ca[i], cb[j], acc[i][j], 0, 0, 0);
        // schedule: the 4 copies first, then one fragment read between every two MFMAs (left alone the compiler sinks all 12
        // reads behind the 32 MFMAs and the next step starts by waiting for them)
        constexpr int PER = (TM * TN) / (TM + TN);                                   // MFMAs per fragment read: 2 (BN = 256) or 1
        __builtin_amdgcn_sched_group_barrier(0x020, CP, 0);                          // VMEM (the LDS-DMA copies)
#pragma unroll
        for (int k = 0; k < TM + TN; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       // one DS read
            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);                     // PER MFMAs
        }
        __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - PER * (TM + TN), 0);
    };
    if constexpr (CP == 4) __builtin_amdgcn_s_waitcnt(0x0f78); else __builtin_amdgcn_s_waitcnt(0x0f76);          // vmcnt(2 CP): my copies of step 0
    __builtin_amdgcn_s_barrier();
    load_frags(fa0, fb0, 0);
    for (int s = 0; s < nsteps; s += 2) {
        body(fa0, fb0, fa1, fb1, s);
        if (s + 1 < nsteps) body(fa1, fb1, fa0, fb0, s + 1);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);                  // the tail's redundant copies must not outlive the workgroup's LDS
    }
    if constexpr (TR) {
    // Variant TR (wide, PLAIN outputs only -- bias at most; the prefill's qkv / gate_up / o / down: gemm_launch never sends anything
    // else here): the MFMAs ran with swapped operands, so each accumulator tile is the TRANSPOSE of the usual layout: a lane holds 4
    // consecutive output COLUMNS (n = j * 16 + fq * 4 + r) of one row (m = i * 16 + fr) -> 8-byte stores straight from the registers.
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int m = m0 + wr * 128 + i * 16 + fr, n = n0 + wc * WN + j * 16 + fq * 4;
            if (m >= a.M || n >= a.N) continue;
            if (a.act == 2) {
                // SwiGLU over a 16-row-interleaved [gate | up] weight (GemmArgs::Wi): tile j even = gate columns, j + 1 = the matching up
                // columns, in the same lane and register.  y = rnd(rnd(silu(rnd(g))) * rnd(u)): the values of the GEMM + silu_mul_kernel pair
                if constexpr (TN >= 2) {
                    if ((j & 1) == 0) {
                        const int no = (n0 + wc * WN + j * 16) / 2 + fq * 4;      // logical output column of this lane's four values
                        float y[4];
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            const float g = DT<T>::rnd(acc[i][j][c]), u = DT<T>::rnd(acc[i][j + 1 < TN ? j + 1 : j][c]);
                            y[c] = DT<T>::rnd(g / (1.0f + expf(-g))) * u;
                        }
                        *reinterpret_cast<uint2*>(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + no) = uint2{pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3])};
                    }
                }
                continue;
            }
            f32x4_t t = acc[i][j];
            const int ch = n % a.bias_mod;
            if (n + 3 < a.N && (a.ldy & 3) == 0 && (a.bias_mod & 3) == 0) {
                if (a.bias) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) t[c] += DT<T>::ld(reinterpret_cast<const T*>(a.bias) + ch + c);
                }
                *reinterpret_cast<uint2*>(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + n) = uint2{pack_bf16x2(t[0], t[1]), pack_bf16x2(t[2], t[3])};
            } else {
                for (int c = 0; c < 4 && n + c < a.N; ++c)          // ragged right edge / unaligned leading dimension
                    DT<T>::st(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + n + c,
                              t[c] + (a.bias ? DT<T>::ld(reinterpret_cast<const T*>(a.bias) + (n + c) % a.bias_mod) : 0.f));
            }
        }
    } else if constexpr (BN == 256) {
    // Variant !TR (tall outputs with the full epilogue: the codec's convs): through LDS, in two halves of 4 x 4 MFMA tiles.
    // Each wave parks 64 x 64 fp32 values in its own 16 KB region (static register indices only), then walks them with 4
    // columns per lane and 4 rows per wave instruction: 128 contiguous bytes per output row, per-column constants hoisted.
    __syncthreads();
    float* park = reinterpret_cast<float*>(big_smem) + wave * (64 * 64);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    park[(i * 16 + fq * 4 + r) * 64 + j * 16 + fr] = acc[half * 4 + i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        const int mb = m0 + wr * 128 + half * 64;
        const int nq = n0 + wc * 64 + (lane & 15) * 4;                   // this lane's 4 consecutive columns
        const bool vec_ok = (a.ldy & 3) == 0 && (!a.res || (a.ldr & 3) == 0) && nq + 3 < a.N && a.bias_mod >= 4 && (a.bias_mod & 3) == 0;
        if (vec_ok && !std::is_same<TE, bf16_t>::value) {
            // other storage types (the bf16 x 2 mode): the shared 4-column epilogue, one row of 16 lanes x 4 columns per step
            const EpiQ e = epiq_of(a);
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4), m = mb + row;
                if (m >= a.M) break;
                epi_quad<TE>(e, *reinterpret_cast<const f32x4_t*>(park + row * 64 + (lane & 15) * 4), m, nq);
            }
        } else if (vec_ok) {
            const int ch = nq % a.bias_mod;
            float b[4], sc[4], sa[4], sib[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                b[c] = a.bias ? DT<T>::ld(reinterpret_cast<const T*>(a.bias) + ch + c) : 0.f;
                sc[c] = a.scale ? DT<T>::ld(reinterpret_cast<const T*>(a.scale) + nq + c) : 1.f;
                sa[c] = (a.Y2 && !a.act2) ? DT<T>::ld(reinterpret_cast<const T*>(a.sn_a) + ch + c) : 0.f;
                sib[c] = (a.Y2 && !a.act2) ? DT<T>::ld(reinterpret_cast<const T*>(a.sn_ib) + ch + c) : 0.f;
            }
            for (int it = 0; it < 16; ++it) {
                const int row = it * 4 + (lane >> 4), m = mb + row;
                if (m >= a.M) break;                                     // rows ascend with `it` for every lane group
                const f32x4_t pv = *reinterpret_cast<const f32x4_t*>(park + row * 64 + (lane & 15) * 4);
                float rv[4] = {0.f, 0.f, 0.f, 0.f};
                if (a.res) {
                    const uint2 rr = *reinterpret_cast<const uint2*>(reinterpret_cast<const T*>(a.res) + (size_t)m * a.ldr + nq);
                    rv[0] = __uint_as_float(rr.x << 16); rv[1] = __uint_as_float(rr.x & 0xFFFF0000u);
                    rv[2] = __uint_as_float(rr.y << 16); rv[3] = __uint_as_float(rr.y & 0xFFFF0000u);
                }
                float v[4], v2[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float x = DT<T>::rnd(pv[c] + b[c]);
                    if (a.act == 1) x = DT<T>::rnd(gelu_exact(x));
                    if (a.act == 3) x = DT<T>::rnd(x / (1.0f + expf(-x)));
                    if (a.act >= 4) x = DT<T>::rnd(act_extra(a.act, x));
                    if (a.scale) x = DT<T>::rnd(sc[c] * x);
                    if (a.res) x = x + rv[c];
                    v[c] = DT<T>::rnd(x);
                    v2[c] = 0.f;
                    if (a.Y2) v2[c] = a.act2 ? DT<T>::rnd(elu1(v[c])) : snake_apply<T>(v[c], sa[c], sib[c]);
                }
                if (a.Y) *reinterpret_cast<uint2*>(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + nq) = uint2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                if (a.Y2) *reinterpret_cast<uint2*>(reinterpret_cast<T*>(a.Y2) + (size_t)m * a.ldy + nq) = uint2{pack_bf16x2(v2[0], v2[1]), pack_bf16x2(v2[2], v2[3])};
            }
        } else {
            const int n = n0 + wc * 64 + lane;
            if (n < a.N)
                for (int row = 0; row < 64; ++row) {
                    const int m = mb + row;
                    if (m >= a.M) break;
                    epilogue_elem<TE>(a, park[row * 64 + lane], m, n);
                }
        }
        __builtin_amdgcn_wave_barrier();                 // before the second half overwrites the region
    }
    } else {
    // Variant !TR, BN = 128 (the codec's N = 384 / 768 convs with bias / residual / SnakeBeta second output): the whole 256 x 128
    // tile is parked in the (dead) operand ring in two halves of 128 rows and walked by all 512 threads, 256 contiguous bytes per
    // output row (the register epilogue of the TR variant touches 32-byte runs and calls out of line per 4 columns).
    __syncthreads();
    constexpr int kParkFloats = ST * (BM + BN) * BK * 2 / 4;
    gemm_epilogue_parked<TE, BM, BN, TM, TN, kParkFloats, 4, 512>(a, acc, m0, n0, wr, wc, lane, reinterpret_cast<float*>(big_smem));
    }
}

template <bool TR, int BN, typename TE = bf16_t>
inline void big_go_t(const GemmArgs& a, hipStream_t s) {
    const int rows = a.M - a.m_lo, nseg = a.n_seg > 1 ? a.n_seg : 1;
    const int tx = (a.N + BN - 1) / BN, ty = (rows + kBigBM - 1) / kBigBM;
    const size_t shm = (size_t)kBigStages * (kBigBM + BN) * kBigBK * 2;
    // whole-line copies (PAIR) wherever a group of 64 columns stays inside one tap; the half-line ring of four otherwise (Cin = 96, 160, ...)
    if (a.big_pair >= 0 && a.Cin % 64 == 0) {
        static bool attr_p = false;
        auto kern = big_gemm_kernel<kBigStages, TR, BN, TE, true>;
        if (!attr_p) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr_p = true; }
        hipLaunchKernelGGL(kern, dim3(tx * ty * nseg), dim3(512), shm, s, a, tx, tx * ty, tx * ty * nseg);
        return;
    }
    static bool attr = false;
    auto kern = big_gemm_kernel<kBigStages, TR, BN, TE>;
    if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = true; }
    hipLaunchKernelGGL(kern, dim3(tx * ty * nseg), dim3(512), shm, s, a, tx, tx * ty, tx * ty * nseg);
}
// wide plain outputs (prefill qkv / gate_up: 980 vs 872 TFLOP/s) store straight from the transposed accumulators; tall outputs
// with the full epilogue (codec convs: 661 vs 584) go through the LDS walk with 128-byte row stores
template <typename TE = bf16_t>
inline void big_go(const GemmArgs& a, hipStream_t s) {
    const bool plain = a.act == 0 && !a.scale && !a.res && !a.Y2;
    if constexpr (std::is_same<TE, bf16_t>::value) {
        if (plain && a.N >= 1024 && a.N % kBigBN == 0) { big_go_t<true, kBigBN>(a, s); return; }
    }
    big_go_t<false, kBigBN, TE>(a, s);
}
// does a tiling into 256 x bn tiles use the chip well?  (fill of the tiles) x (fill of the last wave of workgroups); nseg
// independent problems of this shape share the launch
inline bool big_tiling_pays(int rows, int N, int bn, int nseg = 1) {
    const long tx = (N + bn - 1) / bn, ty = (rows + kBigBM - 1) / kBigBM, tiles = tx * ty * nseg;
    const double fill = (double)rows * N / ((double)tx * ty * kBigBM * bn);
    const double eff = (double)tiles / (double)(((tiles + 255) / 256) * 256);
    return tiles >= 64 && fill * eff >= 0.66;
}

// second pass of a split-K GEMM: sum the slices in index order (deterministic), then the ordinary epilogue
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs a) {
    const int rows = a.M - a.m_lo;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)rows * a.N) return;
    const int mm = (int)(e / a.N), n = (int)(e % a.N), m = a.m_lo + mm;
    float sum = 0.f;
    for (int z = 0; z < a.ksplit; ++z) sum += a.ws[(size_t)z * rows * a.N + e];
    const int ch = n % a.bias_mod;
    float v = DT<T>::rnd(sum + (a.bias ? DT<T>::ld(reinterpret_cast<const T*>(a.bias) + ch) : 0.f));
    if (a.act == 1) v = DT<T>::rnd(gelu_exact(v));
    if (a.act == 3) v = DT<T>::rnd(v / (1.0f + expf(-v)));
    if (a.scale) v = DT<T>::rnd(DT<T>::ld(reinterpret_cast<const T*>(a.scale) + n) * v);
    if (a.res) v = v + DT<T>::ld(reinterpret_cast<const T*>(a.res) + (size_t)m * a.ldr + n);
    v = DT<T>::rnd(v);
    if (a.Y) DT<T>::st(reinterpret_cast<T*>(a.Y) + (size_t)m * a.ldy + n, v);
    if (a.Y2) DT<T>::st(reinterpret_cast<T*>(a.Y2) + (size_t)m * a.ldy + n,
                        snake_apply<T>(v, DT<T>::ld(reinterpret_cast<const T*>(a.sn_a) + ch), DT<T>::ld(reinterpret_cast<const T*>(a.sn_ib) + ch)));
}

// ---- host-side launch: tile shape per layer ---------------------------------------------------------------------
template <typename T, int BM, int BN, typename TE = T>
inline void gemm_go(const GemmArgs& a, hipStream_t s) {
    // register-prefetch depth: the small tiles serve skinny layers (few workgroups, long K) and need 8 tiles in flight; the
    // 128x64 tile runs with >= 2 workgroups per CU, where 2 is best (530 / 498 / 441 TFLOP/s at PF = 2 / 4 / 8, 4096^3)
    constexpr int PF = (BM + BN) > 160 ? 2 : ((BM + BN) > 128 ? 4 : 8);
    dim3 grid((a.N + BN - 1) / BN, (a.M - a.m_lo + BM - 1) / BM, a.n_seg > 1 ? a.n_seg : 1);
    hipLaunchKernelGGL((conv_gemm_kernel<T, BM, BN, PF, TE>), grid, dim3(256), 0, s, a);
}
inline bool skinny_ok(const GemmArgs& a) {
    return a.ws && !a.no_skinny && a.n_taps == 1 && a.tap_off[0] == 0 && a.m_lo == 0 && a.M <= kSkinnyMaxRows && a.a_rows >= a.M && a.act == 0 &&
           !a.bias && !a.scale && !a.Y2 && a.Y && skinny_k_ok(a.Cin) && a.N % 32 == 0 && a.lda % 8 == 0 && a.ldy % 4 == 0 && (!a.res || a.ldr % 4 == 0);
}
// T = operand type, TE = storage type of the outputs (see conv_gemm_kernel).  nseg independent problems share a launch
// (batched decode): what counts for the tile choice is the number of workgroups the whole launch has.
template <typename T, typename TE>
inline void gemm_launch_te(const GemmArgs& a, hipStream_t s) {
    const int rows = a.M - a.m_lo, nseg = a.n_seg > 1 ? a.n_seg : 1;
    if (rows <= 0) return;
    if constexpr (sizeof(T) == 4) { gemm_go<T, 64, 64>(a, s); return; }      // fp32 = parity mode, one shape
    else {
        constexpr bool kPlainBf16 = std::is_same<TE, bf16_t>::value;
        auto wgs = [&](int bm, int bn) { return (long)((rows + bm - 1) / bm) * ((a.N + bn - 1) / bn) * nseg; };
        // few rows against a whole weight matrix (the short-prompt prefill): weight-stationary kernel (skinny_gemm.cuh).  Its
        // accumulation order is its own, so -- like split-K below -- only for callers that lend a workspace.
        if constexpr (kPlainBf16) {
        if (nseg == 1 && skinny_ok(a)) {
            SkinnyArgs k{};
            k.X = reinterpret_cast<const bf16_t*>(a.A); k.ldx = a.lda; k.M = a.M; k.W = reinterpret_cast<const bf16_t*>(a.W); k.N = a.N;
            k.res = reinterpret_cast<const bf16_t*>(a.res); k.ldr = a.ldr; k.Y = reinterpret_cast<bf16_t*>(a.Y); k.ldy = a.ldy;
            k.Wp = reinterpret_cast<const bf16_t*>(a.Wp);
            if (a.res) skinny_launch<SK_RESIDUAL>(k, a.Cin, s); else skinny_launch<SK_STORE>(k, a.Cin, s);
            return;
        }
        // few rows, one tap, long K, narrow N (the 200-token prefill's o_proj / down): split K over workgroups, two passes.
        // Only where the caller lends a workspace: the codec does not (a tail decode must stay bit-identical to a full one).
        if (nseg == 1 && a.ws && a.n_taps == 1 && a.act != 2 && wgs(64, 32) < 384 && a.Cin >= 1536) {
            int S = 0;
            for (int cand : {8, 6, 4, 3, 2})
                if (a.Cin % (cand * 32) == 0 && a.Cin / (cand * 32) >= 12 && wgs(64, 32) * cand <= 1024 &&
                    (long)cand * rows * a.N <= a.ws_floats) { S = cand; break; }
            if (S > 1) {
                GemmArgs b = a;
                b.ksplit = S;
                dim3 grid((a.N + 31) / 32, (rows + 63) / 64, S);
                hipLaunchKernelGGL((conv_gemm_kernel<T, 64, 32, 8>), grid, dim3(256), 0, s, b);
                hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3((unsigned)(((size_t)rows * a.N + 255) / 256)), dim3(256), 0, s, b);
                return;
            }
        }
        }
        // Measured on MI355X (tools/microbench/gemm_bench.hip): the 128x64 tile reaches 390 TFLOP/s where 128x128 stays at
        // 80-110 (130 VGPRs / 40 KB LDS leave 3 workgroups per CU against 5: the one-barrier K step is latency-bound, so
        // residency beats arithmetic intensity here) -- the 128x128 shape is therefore not used.
        const bool n64 = a.N % 64 == 0 || a.N >= 512;
        // 256 x 256 ring-of-four tile (932 TFLOP/s at 4096^3 on random operands against 622 for the 128 x 64 glds tile): one
        // workgroup per CU, so it pays only when the tiles are well filled AND the last wave of workgroups is mostly full
        if (a.act != 2 && a.Cin % 32 == 0) {
            if (big_tiling_pays(rows, a.N, kBigBN, nseg)) { big_go<TE>(a, s); return; }
            // 256 x 128 tiles where the 256-wide ones would leave half the CUs idle (N = 2048 at M = 4096: o_proj / down at 1.7B)
            if (rows >= 1024 && a.N % 128 == 0 && big_tiling_pays(rows, a.N, 128, nseg)) {
                const bool plain = a.act == 0 && !a.scale && !a.res && !a.Y2;
                if constexpr (kPlainBf16) { if (plain) { big_go_t<true, 128>(a, s); return; } }
                big_go_t<false, 128, TE>(a, s);      // full epilogue: LDS-parked, whole rows per store
                return;
            }
        }
        // LDS-DMA 128 x 64 tile.  Up to ~two rounds of tiles: by EIGHT waves (a wave takes delivery of ~3.6 B/clk, so a CU with one 4-wave
        // tile starves: profiles/r06_dma_rate.txt) -- also for the few-row GEMMs of a streaming chunk (M = 52 .. 2080 against K = 5376 ..
        // 14336: -25 .. -34 % against the register-prefetch tiles, profiles/r06_glds_8waves.txt, r06_glds_few_rows.txt).  Bit-identical.
        const long t64 = wgs(128, 64);
        const bool glds_shape = a.N % 64 == 0 && a.Cin % 64 == 0;
        const bool parks = a.act != 2 && a.ksplit <= 1 && !a.epi_legacy && a.glds_waves >= 0;
        // a handful of output tiles against a long K (a streaming chunk's dec.0, the frame-level transformer's GEMMs): the chain kernel --
        // NW waves deliver the operands of ONE 32 x 32 tile, the accumulators travel (-20 .. -40 % against the eight-wave tile up to ~200
        // tiles, slower beyond: profiles/r06_chain_gemm.txt, r06_chain_small.txt, r06_chain_taps.txt).  Bit-identical.
        const long t32 = wgs(32, 32);
        // (K <= 4096: behind that the 64 x 64 eight-wave tile is ahead again -- dec.0 at 52 rows, K = 14336: 84 against 94 us,
        // profiles/r06_chain_vs_64rows.txt)
        if (a.chain >= 0 && chain_ok(a) && t32 <= 224 && (long)a.n_taps * a.Cin >= 512 && (long)a.n_taps * a.Cin <= 4096) { chain_go<8, 8, TE>(a, s); return; }
        // (bf16 x 2 outputs: the four-wave tile's epilogue needs 224 registers -- two workgroups = eight waves per CU whatever the grid -- so the
        // eight-wave tiles (74 .. 82 registers, two or three workgroups per CU) serve the large grids too: a group of 32 first chunks 0.327 ->
        // 0.275 ms each, 370 frames 12.4 -> 12.0 ms, profiles/r06_codec_time_bf16x2_cap8.txt)
        const long cap8 = std::is_same<TE, bfs_t>::value ? (a.glds_cap8 > 0 ? a.glds_cap8 : (1L << 30)) : 512;
        if (glds_shape && parks && t64 <= cap8 && (long)a.n_taps * a.Cin >= 512) {
            // which eight-wave tile: 64-row tiles put twice the workgroups on the chip (profiles/r06_glds_64rows.txt): 64 x 128 where that
            // fills 140 .. 512 CUs' worth, 64 x 64 for the smaller grids, 128 x 64 where 64-row tiles would run to several rounds
            const long ta = a.N % 128 == 0 ? wgs(64, 128) : 0, tb = wgs(64, 64);
            if (ta >= 140 && ta <= cap8) glds_go<128, 3, TE, 512, 64>(a, s);
            else if (tb <= 256) glds_go<64, 3, TE, 512, 64>(a, s);
            else if (t64 <= 256) glds_go<64, 3, TE, 512>(a, s);
            else glds_go<64, 2, TE, 512>(a, s);
        } else if (glds_shape && t64 >= (a.glds_min_wgs > 0 ? a.glds_min_wgs : kGldsMinWgs)) {
            glds_go<64, 2, TE>(a, s);
        } else if (n64 && wgs(128, 64) >= 512) gemm_go<T, 128, 64, TE>(a, s);
        else if (a.act != 2 && !n64 && a.N % 96 == 0 && wgs(128, 96) >= 256) gemm_go<T, 128, 96, TE>(a, s);      // N = 96 / 288 (codec block 4): 3x fewer reads of the A tile than 32-wide tiles
        else if (a.act != 2 && !n64 && wgs(128, 32) >= 256) gemm_go<T, 128, 32, TE>(a, s);
        else if (n64 && (wgs(64, 64) >= 256 || a.act == 2)) gemm_go<T, 64, 64, TE>(a, s);
        else if (a.act == 2) gemm_go<T, 64, 64, TE>(a, s);
        else gemm_go<T, 64, 32, TE>(a, s);
    }
}
// Storage type T in {bf16, fp32, bfs_t}.  bfs_t (the codec's bf16 x 2 mode): the A operand is the [rows][2 lda] bf16 image of the
// bfs_t tensor -- hi / lo interleaved along K -- against a bf16 weight whose K columns are duplicated at pack time
// ([N][taps][2 Cin]); everything else (bias, residual, outputs) is bfs_t.
template <typename T>
inline void gemm_launch(const GemmArgs& a, hipStream_t s) {
    if constexpr (std::is_same<T, bfs_t>::value) {
        GemmArgs b = a;
        b.lda = 2 * a.lda; b.Cin = 2 * a.Cin; b.a_seg = 2 * a.a_seg;
        b.ws = nullptr;
        gemm_launch_te<bf16_t, bfs_t>(b, s);
    } else gemm_launch_te<T, T>(a, s);
}

// forward declaration target of gemm_swiglu_halves
template <typename T> __global__ void silu_mul_kernel(const T* gu, T* y, int rows, int I);

// y[M][I] = silu(x W_gate^T) * (x W_up^T) for a weight of two halves [gate | up] (N = 2 I): one weight-stationary launch where
// skinny_gemm.cuh serves the shape (the [M][2I] image is never written), else the GEMM into `gu` and the elementwise pass.
template <typename T>
inline void gemm_swiglu_halves(const GemmArgs& a, void* y, hipStream_t s) {
    const int I = a.N / 2;
    if constexpr (sizeof(T) == 2) {
        if (skinny_ok(a) && !a.res && I % 8 == 0) {
            SkinnyArgs k{};
            k.X = reinterpret_cast<const bf16_t*>(a.A); k.ldx = a.lda; k.M = a.M; k.W = reinterpret_cast<const bf16_t*>(a.W); k.N = a.N;
            k.Y = reinterpret_cast<bf16_t*>(y); k.ldy = I;
            k.Wp = reinterpret_cast<const bf16_t*>(a.Wp);
            skinny_launch<SK_SWIGLU>(k, a.Cin, s);
            return;
        }
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
        // many rows (packed prefills, long prompts): the 256-wide ring tile over the 16-row-interleaved copy, SwiGLU in its epilogue -- the
        // [M][2I] image is never written and the elementwise launch (13 us per layer of a 10 x 200 pack) is gone.  Bit-identical to the pair.
        const int rows = a.M - a.m_lo;
        if (a.Wi && !a.res && !a.bias && !a.scale && a.n_taps == 1 && a.N % 512 == 0 && a.Cin % 32 == 0 && a.n_seg <= 1 && I % 4 == 0 &&
            big_tiling_pays(rows, a.N, kBigBN, 1)) {
            GemmArgs b = a;
            b.W = a.Wi; b.act = 2; b.Y = y; b.ldy = I;
            big_go_t<true, kBigBN>(b, s);
            return;
        }
    }
    gemm_launch<T>(a, s);
    hipLaunchKernelGGL((silu_mul_kernel<T>), dim3((unsigned)(((size_t)a.M * I + 255) / 256)), dim3(256), 0, s, (const T*)a.Y, (T*)y, a.M, I);
}

// ---- RVQ: sequential (rounded) sum of codebook rows; one block per frame ------------------------------
struct RvqArgs { const void* books[32]; int nq; int n_first; int dim; };
template <typename T>
__global__ void rvq_gather_kernel(RvqArgs a, const int64_t* codes, T* first, T* rest, int Tn, int row_lo) {
    const int t = row_lo + blockIdx.x;               // rows [row_lo, Tn): a decode behind a cached reference prefix starts at its first new frame
    {   // batched decode: utterance blockIdx.y (tensors compact per utterance: Tn rows each)
        const size_t g = blockIdx.y;
        codes += g * Tn * a.nq; first += g * Tn * a.dim; rest += g * Tn * a.dim;
    }
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

// ---- cached reference-prefix rows -> an utterance's workspace (fq3_codec.hip: fq3_codec_prefix) -----------------
// utterance u = blockIdx.y: n_rows rows of n_cols elements from src[u] (+ src_off, leading dimension src_ld) to rows dst_row0 .. of the
// utterance's [rows_per_utt][dst_ld] tensor, columns dst_col0 ..; 16-byte pieces (n_cols, the offsets and leading dimensions are multiples of 16 bytes)
struct PrefixSrc { const void* p[128]; };
template <typename T>
__global__ __launch_bounds__(256) void prefix_rows_kernel(PrefixSrc src, size_t src_off, int src_ld, T* dst, size_t dst_seg, int dst_ld,
                                                          int dst_row0, int dst_col0, int n_rows, int n_cols) {
    constexpr int EPC = 16 / sizeof(T);
    const int cpr = n_cols / EPC;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n_rows * cpr) return;
    const int r = e / cpr, c = (e - r * cpr) * EPC;
    const T* sp = reinterpret_cast<const T*>(src.p[blockIdx.y]) + src_off + (size_t)r * src_ld + c;
    T* dp = dst + (size_t)blockIdx.y * dst_seg + (size_t)(dst_row0 + r) * dst_ld + dst_col0 + c;
    *reinterpret_cast<u32x4*>(dp) = *reinterpret_cast<const u32x4*>(sp);
}

// ---- row norms: one wave per row, rows [row_lo, rows) ---------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const T* x, const T* w, T* y, int row_lo, int rows, int C, float eps) {
    const int row = row_lo + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    x += (size_t)blockIdx.y * rows * C; y += (size_t)blockIdx.y * rows * C;       // batched decode: utterance blockIdx.y
    const T* xr = x + (size_t)row * C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) { const float v = DT<T>::ld(xr + c); ss = fmaf(v, v, ss); }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)C + eps);
    for (int c = lane; c < C; c += 64)
        DT<T>::st(y + (size_t)row * C + c, DT<T>::ld(w + c) * DT<T>::rnd(DT<T>::ld(xr + c) * rs));
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(const T* x, const T* w, const T* b, T* y, int row_lo, int rows, int C, float eps) {
    const int row = row_lo + blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    x += (size_t)blockIdx.y * rows * C; y += (size_t)blockIdx.y * rows * C;       // batched decode: utterance blockIdx.y
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

// causal depthwise conv k=7 over time, channels-last, rows [row_lo, rows)
template <typename T>
__global__ void dwconv7_kernel(const T* x, const T* w /*[C][7]*/, const T* b, T* y, int row_lo, int rows, int C) {
    const size_t i = (size_t)row_lo * C + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * C) return;
    x += (size_t)blockIdx.y * rows * C; y += (size_t)blockIdx.y * rows * C;       // batched decode: utterance blockIdx.y
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
__global__ void silu_mul_kernel(const T* gu, T* y, int rows, int I) {     // gu [rows][2I] = gate | up (not interleaved)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)rows * I) return;
    const int r = (int)(i / I), c = (int)(i % I);
    const float g = DT<T>::ld(gu + (size_t)r * 2 * I + c), u = DT<T>::ld(gu + (size_t)r * 2 * I + I + c);
    DT<T>::st(y + i, DT<T>::rnd(g / (1.0f + expf(-g))) * u);
}

// RoPE on the q and k thirds of qkv [rows][3*QD], head_dim HD (rotate_half convention), position = row
template <typename T>
__global__ void rope_rows_kernel(T* qkv, const float* cos_tab, const float* sin_tab, int rows, int QD, int HD, int row_lo) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int half = HD / 2, per_row = 2 * (QD / HD) * half;
    if (i >= (size_t)(rows - row_lo) * per_row) return;
    qkv += (size_t)blockIdx.y * rows * 3 * QD;                                     // batched decode: utterance blockIdx.y (positions restart at 0)
    const int t = row_lo + (int)(i / per_row), r = (int)(i % per_row);            // rows [row_lo, rows); position = absolute row
    const int head = r / half, j = r % half;              // heads 0..QD/HD-1 = q, then k
    T* p = qkv + (size_t)t * 3 * QD + (size_t)head * HD;
    const float cs = cos_tab[(size_t)t * half + j], sn = sin_tab[(size_t)t * half + j];
    const float x0 = DT<T>::ld(p + j), x1 = DT<T>::ld(p + j + half);
    DT<T>::st(p + j, DT<T>::rnd(x0 * cs) + DT<T>::rnd(-x1 * sn));
    DT<T>::st(p + j + half, DT<T>::rnd(x1 * cs) + DT<T>::rnd(x0 * sn));
}

// Causal sliding-window attention, one wave per (query, head), head_dim in {32, 64, 128}, window <= 128; fp32 math, one
// rounding.  Scores: one KEY per lane (the lane reads its key row with 16-byte loads, q is broadcast from LDS);
// softmax across lanes; P.V: one (or two) head dims per lane, looping over the window with the probability broadcast by
// v_readlane.  ~8x fewer instructions than a wave-wide reduction per key.
template <typename T, int HD>
__global__ __launch_bounds__(256) void swa_attn_kernel(const T* qkv, T* out, int Tn, int NH, int window, float scale, int row_lo) {
    constexpr int NCH = HD / 8;
    __shared__ float qs[4][HD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = row_lo + blockIdx.x * 4 + wave, h = blockIdx.y;      // queries [row_lo, Tn) (their keys reach back to row q - window + 1)
    const int QD = NH * HD;
    qkv += (size_t)blockIdx.z * Tn * 3 * QD; out += (size_t)blockIdx.z * Tn * QD;  // batched decode: utterance blockIdx.z
    const bool live = q < Tn;
    const int qq = live ? q : Tn - 1;
    const T* qp = qkv + (size_t)qq * 3 * QD + (size_t)h * HD;
    for (int d = lane; d < HD; d += 64) qs[wave][d] = DT<T>::ld(qp + d);
    __syncthreads();
    if (!live) return;
    const int k_lo = max(0, q - window + 1), nk = q - k_lo + 1;       // nk in [1, window]
    float sc[2], p[2];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
        const int j = ps * 64 + lane;
        const int kk = k_lo + (j < nk ? j : 0);
        const T* kp = qkv + (size_t)kk * 3 * QD + QD + (size_t)h * HD;
        Raw8<T> kr[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) ldraw<false>(kr[c], kp + c * 8);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float kf[8];
            unpack(kr[c], kf);
#pragma unroll
            for (int i = 0; i < 8; ++i) s = fmaf(qs[wave][c * 8 + i], kf[i], s);
        }
        sc[ps] = j < nk ? s * scale : -INFINITY;
    }
    const float mx = wave_max(fmaxf(sc[0], sc[1]));
    p[0] = __expf(sc[0] - mx); p[1] = __expf(sc[1] - mx);
    const float l = wave_sum(p[0] + p[1]);
    float o0 = 0.f, o1 = 0.f;
    const T* vbase = qkv + (size_t)k_lo * 3 * QD + 2 * QD + (size_t)h * HD;
    for (int j = 0; j < nk; ++j) {
        const float pj = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(j < 64 ? p[0] : p[1]), j & 63));
        const T* vp = vbase + (size_t)j * 3 * QD;
        if (HD >= 64 || lane < HD) o0 = fmaf(pj, DT<T>::ld(vp + (lane < HD ? lane : 0)), o0);
        if (HD > 64) o1 = fmaf(pj, DT<T>::ld(vp + lane + 64), o1);
    }
    T* op = out + (size_t)q * QD + (size_t)h * HD;
    if (lane < HD) DT<T>::st(op + lane, o0 / l);
    if (HD > 64) DT<T>::st(op + lane + 64, o1 / l);
}

// Output conv k=7, C -> 1, + clamp to [-1, 1]; fp32 PCM out for samples [t_lo, rows), written to pcm[t - t_lo].
// One workgroup per `spw` consecutive samples (a multiple of 64, <= 256; one thread per sample): the (spw + 6) x C input window is
// copied to LDS as it is (16-byte chunks, no conversion; rows padded by 16 bytes so that consecutive samples' 16-byte reads fall
// into distinct bank quads), the 7 x C weights as fp32.  A thread walks its 7 rows with 16-byte LDS reads (the weights are
// broadcast reads: every lane the same address) and one fp32 fma chain in (tap, channel) order.  The input is read from HBM once:
// 110 MB for 300 decoded frames (the round-2 kernel staged fp32 element by element and took 331 us for them).
constexpr int kFinalSpwMax = 256;
template <typename T>
__global__ __launch_bounds__(256) void final_conv_kernel(const T* x, const T* w /*[7][C]*/, const T* b, float* pcm, int t_lo, int rows, int C, int spw) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm_raw[];
    constexpr int EPC = 16 / sizeof(T);                       // elements per 16-byte chunk
    const int pitch = C + EPC;                                // elements per staged row
    T* xs = reinterpret_cast<T*>(fsm_raw);                    // [spw + 6][pitch]
    float* ws = reinterpret_cast<float*>(fsm_raw + (((size_t)(spw + 6) * pitch * sizeof(T) + 15) & ~(size_t)15));       // [7][C]
    const int t0 = t_lo + blockIdx.x * spw;
    const int cpr = C / EPC;                                  // chunks per row
    x += (size_t)blockIdx.y * rows * C; pcm += (size_t)blockIdx.y * (rows - t_lo);     // batched decode: utterance blockIdx.y
    for (int e = threadIdx.x; e < (spw + 6) * cpr; e += 256) {
        const int r = e / cpr, c = (e - r * cpr) * EPC, tt = t0 - 6 + r;
        u32x4 v = u32x4{0u, 0u, 0u, 0u};
        if (tt >= 0 && tt < rows) v = *reinterpret_cast<const u32x4*>(x + (size_t)tt * C + c);
        *reinterpret_cast<u32x4*>(xs + (size_t)r * pitch + c) = v;
    }
    for (int e = threadIdx.x; e < 7 * C; e += 256) ws[e] = DT<T>::ld(w + e);
    __syncthreads();
    const int s = threadIdx.x, t = t0 + s;
    if (s >= spw || t >= rows) return;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const T* xr = xs + (size_t)(s + k) * pitch;
        const float* wr = ws + k * C;
        for (int c = 0; c < C; c += 8) {
            float xf[8];
            if constexpr (sizeof(T) == 2) {
                Raw8<T> q;
                q.v = *reinterpret_cast<const u32x4*>(xr + c);
                unpack(q, xf);
            } else if constexpr (std::is_same<T, bfs_t>::value) {
                Raw8<T> q;
                q.a = *reinterpret_cast<const u32x4*>(xr + c); q.b = *reinterpret_cast<const u32x4*>(xr + c + 4);
                unpack(q, xf);
            } else {
                const f32x4 lo = *reinterpret_cast<const f32x4*>(xr + c), hi = *reinterpret_cast<const f32x4*>(xr + c + 4);
                xf[0] = lo.x; xf[1] = lo.y; xf[2] = lo.z; xf[3] = lo.w; xf[4] = hi.x; xf[5] = hi.y; xf[6] = hi.z; xf[7] = hi.w;
            }
            const f32x4 w0 = *reinterpret_cast<const f32x4*>(wr + c), w1 = *reinterpret_cast<const f32x4*>(wr + c + 4);
            acc = fmaf(w0.x, xf[0], acc); acc = fmaf(w0.y, xf[1], acc); acc = fmaf(w0.z, xf[2], acc); acc = fmaf(w0.w, xf[3], acc);
            acc = fmaf(w1.x, xf[4], acc); acc = fmaf(w1.y, xf[5], acc); acc = fmaf(w1.z, xf[6], acc); acc = fmaf(w1.w, xf[7], acc);
        }
    }
    const float v = DT<T>::rnd(acc + DT<T>::ld(b));
    pcm[t - t_lo] = fminf(1.f, fmaxf(-1.f, v));
}
// samples per workgroup and dynamic LDS bytes of final_conv_kernel for C channels of element size esz (0: C does not fit)
inline int final_conv_spw(int C, int esz, size_t* shm) {
    const int pitch = C + 16 / esz;
    for (int spw = kFinalSpwMax; spw >= 64; spw -= 64) {
        const size_t bytes = (((size_t)(spw + 6) * pitch * esz + 15) & ~(size_t)15) + (size_t)7 * C * sizeof(float);
        if (bytes <= 64 * 1024 || (spw == 64 && bytes <= 150 * 1024)) { *shm = bytes; return spw; }
    }
    return 0;
}

}  // namespace fq3
