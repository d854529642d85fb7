// Decode-step kernels for the Qwen3-TTS talker / code predictor (gfx950, wave64).
//
// Design: batch-1 decode is a chain of M=1 GEMVs; every kernel boundary costs
// ~1.2-1.5 us on MI355X and a grid barrier costs more, so a layer is FIVE weight-streaming launches
// with everything else fused into their prologues/epilogues:
//   1. RMSNorm  -> [q|k|v] GEMV                                  (gemv<PRO_NORM, EPI_STORE>)
//   2. q/k head-RMSNorm + RoPE + KV append + split-KV attention   (attn_decode)
//   3. split combine -> o_proj GEMV -> + residual                 (gemv<PRO_COMBINE, EPI_RESIDUAL>)
//   4. RMSNorm -> [gate|up] GEMV -> SiLU(gate)*up                 (gemv<PRO_NORM, EPI_SWIGLU>)
//   5. down GEMV -> + residual                                    (gemv<PRO_PLAIN, EPI_RESIDUAL>)
// Weights go HBM -> VGPR with 16-byte loads, all of a wave's rows in flight before the first use;
// x is staged once per block in LDS (fp32) and then held in registers; reductions are wave
// butterflies.  Rounding points follow the module-by-module Torch execution the reference replays
// (bf16 after every Linear / norm / RoPE / residual add; fp32 inside dot products and softmax).
#pragma once
#include "fq3_common.cuh"

namespace fq3 {

enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_COMBINE = 2, PRO_ATTN = 3 };
enum { EPI_STORE = 0, EPI_RESIDUAL = 1, EPI_SWIGLU = 2 };

constexpr int kHeadDim = 128;        // Qwen3-TTS talker / predictor head_dim
constexpr int kKeysPerTile = 64;     // keys one attention workgroup handles per loop trip
constexpr int kMaxWorkers = 8;       // attention workgroups (partial slots) per kv head
constexpr int kPartStride = kHeadDim + 4;     // 128 values + {m, l} + pad: keeps every slot 16-byte aligned

// Latency notes (measured on MI355X, profiles/r01_decode_trace_v0.txt): a dependent global round
// trip between two kernels costs ~0.5-1 us, so every kernel below issues ALL of its global loads
// (weights, input vector, norm gains, residual, partial slots, position) before the first wait, and
// needs at most one workgroup barrier.

struct GemvArgs {
    const void* W; int N; int K; int R;           // N logical rows (pairs for SWIGLU); R rows per wave
    const void* x;                                 // input vector T[K] (PRO_PLAIN / PRO_NORM)
    const void* norm_w; float eps;
    const void* bias;
    void* y;                                       // T[N]
    const void* res;                               // residual input T[N] (EPI_RESIDUAL), may alias y
    void* xn_out;                                  // optional: block 0 stores the prologue result as T[K]
    int up_off;                                    // SWIGLU: first "up" row
    const float* part; int n_part; int rep;        // PRO_COMBINE: attention partial slots
    // second token of an M = 2 launch
    const void* x2; void* y2; const void* res2; size_t part_stride2;
    // PRO_ATTN (short-context attention computed redundantly in every workgroup: code predictor, <= 17 keys)
    const void* qkv; const void* q_norm_w; const void* k_norm_w;
    const float* cos_row; const float* sin_row;
    void* kcache; void* vcache; int max_seq; int pos; int n_kv; float scale;
};

// Attention over a SHORT context for all heads inside one workgroup, result (T-rounded, fp32 image) into
// xs[q_dim].  Wave w owns kv heads w, w+4, ...: no cross-wave merge.  Used as the o_proj prologue of the
// code predictor (context <= 2 + 15 tokens, predictor_graph.py:46), which removes one launch per layer.
template <typename T, int REP>
__device__ __forceinline__ void attn_small_prologue(const GemvArgs& a, float* xs, float* scratch) {
    constexpr int HD = kHeadDim;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane >> 4, c = lane & 15;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const int q_dim = a.n_kv * REP * HD, kv_dim = a.n_kv * HD;
    float* qs = scratch + wave * (REP + 2) * HD;          // [REP][HD] q, then k, then v (wave-private)
    float* knew = qs + REP * HD;
    float* vnew = knew + HD;
    const float cs = a.cos_row[lane], sn = a.sin_row[lane];
    const T* qw = reinterpret_cast<const T*>(a.q_norm_w);
    const T* kw = reinterpret_cast<const T*>(a.k_norm_w);
    const float qw0 = DT<T>::ld(qw + lane), qw1 = DT<T>::ld(qw + lane + 64);
    const float kw0 = DT<T>::ld(kw + lane), kw1 = DT<T>::ld(kw + lane + 64);
    const int pos = a.pos;
    const int iters = (a.n_kv + 3) / 4;
    for (int it = 0; it < iters; ++it) {
        const int g = it * 4 + wave;
        const bool active = g < a.n_kv;
        const int gg = active ? g : 0;
        T* kc = reinterpret_cast<T*>(a.kcache) + (size_t)gg * a.max_seq * HD;
        T* vc = reinterpret_cast<T*>(a.vcache) + (size_t)gg * a.max_seq * HD;
        Raw8<T> kr[4], vr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = i * 4 + sub;
            key = key < a.max_seq ? key : a.max_seq - 1;
            ldraw<false>(kr[i], kc + (size_t)key * HD + c * 8);
            ldraw<false>(vr[i], vc + (size_t)key * HD + c * 8);
        }
        // q heads, new k (norm + RoPE), new v
#pragma unroll
        for (int vec = 0; vec < REP + 2; ++vec) {
            const T* src = vec < REP ? qkv + (size_t)(gg * REP + vec) * HD
                         : (vec == REP ? qkv + q_dim + (size_t)gg * HD : qkv + q_dim + kv_dim + (size_t)gg * HD);
            float x0 = DT<T>::ld(src + lane), x1 = DT<T>::ld(src + lane + 64);
            if (vec <= REP) {
                const float w0 = vec < REP ? qw0 : kw0, w1 = vec < REP ? qw1 : kw1;
                const float ss = wave_sum(fmaf(x0, x0, x1 * x1));
                const float rs = 1.0f / sqrtf(ss / (float)HD + a.eps);
                const float n0 = DT<T>::rnd(w0 * DT<T>::rnd(x0 * rs));
                const float n1 = DT<T>::rnd(w1 * DT<T>::rnd(x1 * rs));
                x0 = DT<T>::rnd(DT<T>::rnd(n0 * cs) + DT<T>::rnd(-n1 * sn));
                x1 = DT<T>::rnd(DT<T>::rnd(n1 * cs) + DT<T>::rnd(n0 * sn));
            }
            float* dst = vec < REP ? qs + vec * HD : (vec == REP ? knew : vnew);
            dst[lane] = x0; dst[lane + 64] = x1;
            if (vec >= REP && active && blockIdx.x == 0) {        // one workgroup appends K/V to the cache
                T* cp = (vec == REP ? kc : vc) + (size_t)pos * HD;
                DT<T>::st(cp + lane, x0); DT<T>::st(cp + lane + 64, x1);
            }
        }
        __syncthreads();
        float qr[REP][8], m[REP], l[REP], o[REP][8];
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            m[h] = -1e30f; l[h] = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) { qr[h][i] = qs[h * HD + c * 8 + i]; o[h][i] = 0.f; }
        }
        auto step = [&](const float (&kf)[8], const float (&vf)[8], bool valid) {
#pragma unroll
            for (int h = 0; h < REP; ++h) {
                float sc = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) sc = fmaf(qr[h][i], kf[i], sc);
                sc += __shfl_xor(sc, 1, 64); sc += __shfl_xor(sc, 2, 64);
                sc += __shfl_xor(sc, 4, 64); sc += __shfl_xor(sc, 8, 64);
                sc = valid ? sc * a.scale : -INFINITY;
                const float mn = fmaxf(m[h], sc);
                const float al = __expf(m[h] - mn), p = __expf(sc - mn);
                l[h] = fmaf(l[h], al, p);
#pragma unroll
                for (int i = 0; i < 8; ++i) o[h][i] = fmaf(o[h][i], al, valid ? p * vf[i] : 0.f);
                m[h] = mn;
            }
        };
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float kf[8], vf[8];
            unpack(kr[i], kf); unpack(vr[i], vf);
            step(kf, vf, i * 4 + sub < pos);
        }
        {
            float kf[8], vf[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { kf[i] = knew[c * 8 + i]; vf[i] = vnew[c * 8 + i]; }
            step(kf, vf, sub == 0);
        }
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
            for (int h = 0; h < REP; ++h) {
                const float mo = __shfl_xor(m[h], off, 64), lo = __shfl_xor(l[h], off, 64);
                const float M = fmaxf(m[h], mo);
                const float wa = __expf(m[h] - M), wb = __expf(mo - M);
                l[h] = l[h] * wa + lo * wb;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float oo = __shfl_xor(o[h][i], off, 64);
                    o[h][i] = o[h][i] * wa + oo * wb;
                }
                m[h] = M;
            }
        }
        if (active && sub == 0) {
#pragma unroll
            for (int h = 0; h < REP; ++h) {
                const float inv = 1.0f / l[h];
#pragma unroll
                for (int i = 0; i < 8; ++i) xs[(size_t)(g * REP + h) * HD + c * 8 + i] = DT<T>::rnd(o[h][i] * inv);
            }
        }
        __syncthreads();
    }
}

template <int NCH> struct RowsInFlight { static constexpr int v = NCH >= 12 ? 1 : (NCH >= 6 ? 2 : 4); };

// M = number of tokens that share one pass over the weights (2 only for the code predictor's two-token
// prefill, predictor_graph.py:121-128: weights are read once, both tokens' dot products are formed).
template <typename T, int NCH, int PRO, int EPI, bool NT, int M = 1>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) {
    static_assert(M == 1 || PRO != PRO_ATTN, "the fused short-context attention prologue is single-token");
    constexpr int NR = (EPI == EPI_SWIGLU) ? 2 : 1;     // physical rows per logical row
    constexpr int RB = (RowsInFlight<NCH>::v / NR) > 0 ? (RowsInFlight<NCH>::v / NR) : 1;
    extern __shared__ __attribute__((aligned(16))) float xs[];      // PRO_COMBINE / PRO_ATTN only: M*K floats (+ scratch)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = a.K;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int row0 = (blockIdx.x * 4 + wave) * a.R;

    // ---- 1. issue every global load this thread will ever need ------------------------------------
    Raw8<T> raw[NR][RB][NCH];
#pragma unroll
    for (int h = 0; h < NR; ++h)
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = row0 + r;
            const bool rv = (r < a.R) && (row < a.N);
            const T* wr = W + (size_t)(row + h * a.up_off) * K;
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int off = j * 512 + lane * 8;
                if (rv && off < K) ldraw<NT>(raw[h][r][j], wr + off);
                else zero(raw[h][r][j]);
            }
        }
    float resv[M][RB], biasv[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = row0 + r;
        const bool rv = (r < a.R) && (row < a.N);
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const T* rp = reinterpret_cast<const T*>(m == 0 ? a.res : a.res2);
            resv[m][r] = (EPI == EPI_RESIDUAL && rv) ? DT<T>::ld(rp + row) : 0.f;
        }
        biasv[r] = (a.bias && rv) ? DT<T>::ld(reinterpret_cast<const T*>(a.bias) + row) : 0.f;
    }
    float xr[M][NCH][8];
    if (PRO == PRO_ATTN) {
        float* scratch = xs + K;
        if (a.rep == 1) attn_small_prologue<T, 1>(a, xs, scratch);
        else if (a.rep == 2) attn_small_prologue<T, 2>(a, xs, scratch);
        else attn_small_prologue<T, 4>(a, xs, scratch);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) xr[0][j][i] = off < K ? xs[off + i] : 0.f;
        }
    } else if (PRO == PRO_COMBINE) {
        // thread t merges the attention partials of elements [8t, 8t+8) (one head); result via LDS
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const float* part = a.part + (size_t)m * a.part_stride2;
            for (int c0 = tid; c0 * 8 < K; c0 += 256) {
                const int e0 = c0 * 8;
                const int head = e0 / kHeadDim, d0 = e0 - head * kHeadDim;
                const int g = head / a.rep, hh = head - g * a.rep;
                const float* p0 = part + ((size_t)(g * kMaxWorkers) * a.rep + hh) * kPartStride;
                const size_t sstride = (size_t)a.rep * kPartStride;
                f32x4 oa[kMaxWorkers], ob[kMaxWorkers];
                float pm[kMaxWorkers], pl[kMaxWorkers];
#pragma unroll
                for (int s = 0; s < kMaxWorkers; ++s) {
                    if (s < a.n_part) {
                        const float* ps = p0 + s * sstride;
                        oa[s] = *reinterpret_cast<const f32x4*>(ps + d0);
                        ob[s] = *reinterpret_cast<const f32x4*>(ps + d0 + 4);
                        pm[s] = ps[kHeadDim]; pl[s] = ps[kHeadDim + 1];
                    } else { oa[s] = f32x4{0.f, 0.f, 0.f, 0.f}; ob[s] = oa[s]; pm[s] = -1e30f; pl[s] = 0.f; }
                }
                float Mx = pm[0];
#pragma unroll
                for (int s = 1; s < kMaxWorkers; ++s) Mx = fmaxf(Mx, pm[s]);
                f32x4 na = f32x4{0.f, 0.f, 0.f, 0.f}, nb = na;
                float den = 0.f;
#pragma unroll
                for (int s = 0; s < kMaxWorkers; ++s) {
                    const float w = __expf(pm[s] - Mx);
                    na += oa[s] * w; nb += ob[s] * w; den = fmaf(w, pl[s], den);
                }
                const float inv = 1.0f / den;
                float* d = xs + (size_t)m * K + e0;
                d[0] = DT<T>::rnd(na.x * inv); d[1] = DT<T>::rnd(na.y * inv); d[2] = DT<T>::rnd(na.z * inv); d[3] = DT<T>::rnd(na.w * inv);
                d[4] = DT<T>::rnd(nb.x * inv); d[5] = DT<T>::rnd(nb.y * inv); d[6] = DT<T>::rnd(nb.z * inv); d[7] = DT<T>::rnd(nb.w * inv);
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int off = j * 512 + lane * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) xr[m][j][i] = off < K ? xs[(size_t)m * K + off + i] : 0.f;
            }
    } else {
        // every wave reads the whole input vector itself (L2-resident, 2-12 KB): no LDS, no barrier
        Raw8<T> xraw[M][NCH], nraw[NCH];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const T* x = reinterpret_cast<const T*>(m == 0 ? a.x : a.x2);
                if (off < K) ldraw<false>(xraw[m][j], x + off); else zero(xraw[m][j]);
            }
            if (PRO == PRO_NORM) {
                if (off < K) ldraw<false>(nraw[j], reinterpret_cast<const T*>(a.norm_w) + off); else zero(nraw[j]);
            }
        }
#pragma unroll
        for (int m = 0; m < M; ++m) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) unpack(xraw[m][j], xr[m][j]);
            if (PRO == PRO_NORM) {
                float ss = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; ++j)
#pragma unroll
                    for (int i = 0; i < 8; ++i) ss = fmaf(xr[m][j][i], xr[m][j][i], ss);
                ss = wave_sum(ss);
                const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    float nw[8];
                    unpack(nraw[j], nw);
#pragma unroll
                    for (int i = 0; i < 8; ++i) xr[m][j][i] = DT<T>::rnd(nw[i] * DT<T>::rnd(xr[m][j][i] * rs));
                }
            }
        }
    }
    if (a.xn_out && blockIdx.x == 0 && wave == 0) {
        T* o = reinterpret_cast<T*>(a.xn_out);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8;
            if (off < K)
#pragma unroll
                for (int i = 0; i < 8; ++i) DT<T>::st(o + off + i, xr[0][j][i]);
        }
    }

    // ---- 2. dot products + epilogue ------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = row0 + r;
        if (r >= a.R || row >= a.N) break;            // wave-uniform
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float acc[NR];
#pragma unroll
            for (int h = 0; h < NR; ++h) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; ++j) s = dot8<T>(raw[h][r][j], xr[m][j], s);
                acc[h] = wave_sum(s);
            }
            if (lane == 0) {
                T* y = reinterpret_cast<T*>(m == 0 ? a.y : a.y2);
                if (EPI == EPI_SWIGLU) {
                    const float g = DT<T>::rnd(acc[0]);
                    const float u = DT<T>::rnd(acc[NR - 1]);
                    const float sg = DT<T>::rnd(g / (1.0f + expf(-g)));
                    DT<T>::st(y + row, sg * u);
                } else {
                    float v = DT<T>::rnd(acc[0] + biasv[r]);
                    if (EPI == EPI_RESIDUAL) v = v + resv[m][r];
                    DT<T>::st(y + row, v);
                }
            }
        }
    }
}

// ================================================================================================
// Attention for one new token over a contiguous static KV cache [n_kv][max_seq][128].
// grid = (n_kv_heads, n_workers); worker s walks key tiles s, s+S, ... of 64 keys and ALWAYS writes
// its partial slot (empty = {m=-1e30, l=0}), so the consumer (o_proj prologue) needs no position.
// Prologue (every block, redundantly): per-head RMSNorm + RoPE of this group's q heads and of the
// new k; the worker that owns `pos` appends K/V to the cache.  Only live keys are read (the
// reference reads all max_seq slots under an additive mask, talker_graph.py:71-92).
// ================================================================================================
struct AttnArgs {
    const void* qkv;
    const void* q_norm_w; const void* k_norm_w; float eps;
    const float* cos_row; const float* sin_row;       // [64] each: RoPE row of this token's position
    void* kcache; void* vcache; int max_seq;
    const int* pos_ptr; int pos_imm; int n_pad;
    int n_kv; float* part; float scale;
};

template <typename T, int REP>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) {
    constexpr int HD = kHeadDim, KS = kKeysPerTile, NG = KS / 16;
    const int g = blockIdx.x, s = blockIdx.y, S = gridDim.y;
    __shared__ float qs[REP][HD];
    __shared__ float knew[HD], vnew[HD];
    __shared__ float wo[4][REP][HD];
    __shared__ float wm[4][REP], wl[4][REP];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int sub = lane >> 4, c = lane & 15;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const int q_dim = a.n_kv * REP * HD, kv_dim = a.n_kv * HD;
    T* kc = reinterpret_cast<T*>(a.kcache) + (size_t)g * a.max_seq * HD;
    T* vc = reinterpret_cast<T*>(a.vcache) + (size_t)g * a.max_seq * HD;

    // ---- issue: first key tile (unconditionally, clamped), q/k/v rows, gains, rope row, position ------
    Raw8<T> kr[NG], vr[NG];
    auto issue_tile = [&](int tile) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            int key = tile * KS + (i * 4 + wave) * 4 + sub;
            key = key < a.max_seq ? key : a.max_seq - 1;
            ldraw<false>(kr[i], kc + (size_t)key * HD + c * 8);
            ldraw<false>(vr[i], vc + (size_t)key * HD + c * 8);
        }
    };
    issue_tile(s);
    // prologue vectors: wave w handles vec w (and w+4 when REP == 4): q heads, then k, then v
    constexpr int NV = (REP + 2 + 3) / 4;
    float px0[NV], px1[NV], pw0[NV], pw1[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = wave + 4 * i;
        px0[i] = px1[i] = pw0[i] = pw1[i] = 0.f;
        if (vec < REP + 2) {
            const T* src = vec < REP ? qkv + (size_t)(g * REP + vec) * HD
                         : (vec == REP ? qkv + q_dim + (size_t)g * HD : qkv + q_dim + kv_dim + (size_t)g * HD);
            px0[i] = DT<T>::ld(src + lane); px1[i] = DT<T>::ld(src + lane + 64);
            if (vec <= REP) {
                const T* w = reinterpret_cast<const T*>(vec < REP ? a.q_norm_w : a.k_norm_w);
                pw0[i] = DT<T>::ld(w + lane); pw1[i] = DT<T>::ld(w + lane + 64);
            }
        }
    }
    const float cs = a.cos_row[lane], sn = a.sin_row[lane];
    const int pos = a.pos_ptr ? *a.pos_ptr : a.pos_imm;

    const int t_pos = pos / KS;
    const bool owner = (t_pos % S) == s;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = wave + 4 * i;
        if (vec >= REP + 2) continue;
        float x0 = px0[i], x1 = px1[i];
        if (vec <= REP) {
            const float ss = wave_sum(fmaf(x0, x0, x1 * x1));
            const float rs = 1.0f / sqrtf(ss / (float)HD + a.eps);
            const float n0 = DT<T>::rnd(pw0[i] * DT<T>::rnd(x0 * rs));
            const float n1 = DT<T>::rnd(pw1[i] * DT<T>::rnd(x1 * rs));
            x0 = DT<T>::rnd(DT<T>::rnd(n0 * cs) + DT<T>::rnd(-n1 * sn));     // rotate_half: (-x2, x1)
            x1 = DT<T>::rnd(DT<T>::rnd(n1 * cs) + DT<T>::rnd(n0 * sn));
        }
        if (vec < REP) { qs[vec][lane] = x0; qs[vec][lane + 64] = x1; }
        else {
            float* dst = vec == REP ? knew : vnew;
            dst[lane] = x0; dst[lane + 64] = x1;
            if (owner) {
                T* cp = (vec == REP ? kc : vc) + (size_t)pos * HD;
                DT<T>::st(cp + lane, x0); DT<T>::st(cp + lane + 64, x1);
            }
        }
    }
    __syncthreads();

    float qr[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) qr[h][i] = qs[h][c * 8 + i];

    float m[REP], l[REP], o[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        m[h] = -1e30f; l[h] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[h][i] = 0.f;
    }
    auto step = [&](const float (&kf)[8], const float (&vf)[8], bool valid) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float sc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sc = fmaf(qr[h][i], kf[i], sc);
            sc += __shfl_xor(sc, 1, 64); sc += __shfl_xor(sc, 2, 64);
            sc += __shfl_xor(sc, 4, 64); sc += __shfl_xor(sc, 8, 64);
            sc = valid ? sc * a.scale : -INFINITY;
            const float mn = fmaxf(m[h], sc);
            const float al = __expf(m[h] - mn);
            const float p = __expf(sc - mn);
            l[h] = fmaf(l[h], al, p);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[h][i] = fmaf(o[h][i], al, valid ? p * vf[i] : 0.f);
            m[h] = mn;
        }
    };
    for (int tile = s; tile * KS < pos || tile == s; tile += S) {
        if (tile != s) issue_tile(tile);
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int key = tile * KS + (i * 4 + wave) * 4 + sub;
            float kf[8], vf[8];
            unpack(kr[i], kf); unpack(vr[i], vf);
            step(kf, vf, key >= a.n_pad && key < pos);       // slot `pos` itself comes from LDS below
        }
    }
    {   // the new token's own key/value, by lane group 0 of wave 0 of the owning worker
        float kf[8], vf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { kf[i] = knew[c * 8 + i]; vf[i] = vnew[c * 8 + i]; }
        step(kf, vf, owner && wave == 0 && sub == 0 && pos >= a.n_pad);
    }
    // merge the four 16-lane key groups of the wave
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            const float mo = __shfl_xor(m[h], off, 64), lo = __shfl_xor(l[h], off, 64);
            const float M = fmaxf(m[h], mo);
            const float wa = __expf(m[h] - M), wb = __expf(mo - M);
            l[h] = l[h] * wa + lo * wb;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float oo = __shfl_xor(o[h][i], off, 64);
                o[h][i] = o[h][i] * wa + oo * wb;
            }
            m[h] = M;
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wo[wave][h][c * 8 + i] = o[h][i];
            if (c == 0) { wm[wave][h] = m[h]; wl[wave][h] = l[h]; }
        }
    }
    __syncthreads();
    for (int e = tid; e < REP * HD; e += 256) {
        const int h = e / HD, d = e - h * HD;
        float M = fmaxf(fmaxf(wm[0][h], wm[1][h]), fmaxf(wm[2][h], wm[3][h]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float ww = __expf(wm[w][h] - M);
            num = fmaf(ww, wo[w][h][d], num);
            den = fmaf(ww, wl[w][h], den);
        }
        float* p = a.part + (((size_t)g * kMaxWorkers + s) * REP + h) * kPartStride;
        p[d] = num;
        if (d == 0) { p[HD] = M; p[HD + 1] = den; }
    }
}

// ================================================================================================
// Code-predictor attention (context <= 17 keys, GQA ratio 2): ONE wave per kv group, everything in
// registers -- vectors are loaded in the 16-lanes-per-vector layout, head RMSNorm / RoPE / scores use
// 16-lane shuffles, the 16 cached keys are covered by 4 key groups x 4 sub-groups, no LDS, no barrier,
// 8 workgroups in total.  Writes partial slot 0 in the format the o_proj combine prologue reads.
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(64) void attn_pred_kernel(AttnArgs a) {
    constexpr int HD = kHeadDim, REP = 2;
    const int g = blockIdx.x, lane = threadIdx.x & 63;
    const int sub = lane >> 4, c = lane & 15;
    const int pos = a.pos_imm;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const int q_dim = a.n_kv * REP * HD, kv_dim = a.n_kv * HD;
    T* kc = reinterpret_cast<T*>(a.kcache) + (size_t)g * a.max_seq * HD;
    T* vc = reinterpret_cast<T*>(a.vcache) + (size_t)g * a.max_seq * HD;
    Raw8<T> kr[4], vr[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int key = i * 4 + sub;
        key = key < a.max_seq ? key : a.max_seq - 1;
        ldraw<false>(kr[i], kc + (size_t)key * HD + c * 8);
        ldraw<false>(vr[i], vc + (size_t)key * HD + c * 8);
    }
    Raw8<T> qraw[REP], knraw, vnraw, qwraw, kwraw;
#pragma unroll
    for (int h = 0; h < REP; ++h) ldraw<false>(qraw[h], qkv + (size_t)(g * REP + h) * HD + c * 8);
    ldraw<false>(knraw, qkv + q_dim + (size_t)g * HD + c * 8);
    ldraw<false>(vnraw, qkv + q_dim + kv_dim + (size_t)g * HD + c * 8);
    ldraw<false>(qwraw, reinterpret_cast<const T*>(a.q_norm_w) + c * 8);
    ldraw<false>(kwraw, reinterpret_cast<const T*>(a.k_norm_w) + c * 8);
    float cs[8], sn[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { cs[i] = a.cos_row[(c * 8 + i) & 63]; sn[i] = a.sin_row[(c * 8 + i) & 63]; }
    // head RMSNorm + RoPE; the rotate_half partner of dim d (d +- 64) lives in lane c ^ 8 of the same 16-lane group
    auto norm_rope = [&](const Raw8<T>& raw, const Raw8<T>& wraw, float (&out)[8]) {
        float x[8], w[8];
        unpack(raw, x); unpack(wraw, w);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss = fmaf(x[i], x[i], ss);
        ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64); ss += __shfl_xor(ss, 8, 64);
        const float rs = 1.0f / sqrtf(ss / (float)HD + a.eps);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = DT<T>::rnd(w[i] * DT<T>::rnd(x[i] * rs));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float partner = __shfl_xor(x[i], 8, 64);
            const float rot = c < 8 ? -partner : partner;
            out[i] = DT<T>::rnd(DT<T>::rnd(x[i] * cs[i]) + DT<T>::rnd(rot * sn[i]));
        }
    };
    float qr[REP][8], knew[8], vnew[8];
#pragma unroll
    for (int h = 0; h < REP; ++h) norm_rope(qraw[h], qwraw, qr[h]);
    norm_rope(knraw, kwraw, knew);
    unpack(vnraw, vnew);
    if (sub == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { DT<T>::st(kc + (size_t)pos * HD + c * 8 + i, knew[i]); DT<T>::st(vc + (size_t)pos * HD + c * 8 + i, vnew[i]); }
    }
    float m[REP], l[REP], o[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        m[h] = -1e30f; l[h] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[h][i] = 0.f;
    }
    auto step = [&](const float (&kf)[8], const float (&vf)[8], bool valid) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float sc = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) sc = fmaf(qr[h][i], kf[i], sc);
            sc += __shfl_xor(sc, 1, 64); sc += __shfl_xor(sc, 2, 64); sc += __shfl_xor(sc, 4, 64); sc += __shfl_xor(sc, 8, 64);
            sc = valid ? sc * a.scale : -INFINITY;
            const float mn = fmaxf(m[h], sc), al = __expf(m[h] - mn), p = __expf(sc - mn);
            l[h] = fmaf(l[h], al, p);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[h][i] = fmaf(o[h][i], al, valid ? p * vf[i] : 0.f);
            m[h] = mn;
        }
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float kf[8], vf[8];
        unpack(kr[i], kf); unpack(vr[i], vf);
        step(kf, vf, i * 4 + sub < pos);
    }
    step(knew, vnew, sub == 0);
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            const float mo = __shfl_xor(m[h], off, 64), lo = __shfl_xor(l[h], off, 64);
            const float M = fmaxf(m[h], mo), wa = __expf(m[h] - M), wb = __expf(mo - M);
            l[h] = l[h] * wa + lo * wb;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float oo = __shfl_xor(o[h][i], off, 64); o[h][i] = o[h][i] * wa + oo * wb; }
            m[h] = M;
        }
    }
    if (sub == 0) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float* p = a.part + (((size_t)g * kMaxWorkers + 0) * REP + h) * kPartStride;
            *reinterpret_cast<f32x4*>(p + c * 8) = f32x4{o[h][0], o[h][1], o[h][2], o[h][3]};
            *reinterpret_cast<f32x4*>(p + c * 8 + 4) = f32x4{o[h][4], o[h][5], o[h][6], o[h][7]};
            if (c == 0) { p[HD] = m[h]; p[HD + 1] = l[h]; }
        }
    }
}

// ================================================================================================
// Small glue kernels
// ================================================================================================
template <typename T>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const T* x, const T* w, T* y, int K, float eps) {
    __shared__ float red[8];
    float ss = 0.f;
    for (int e = threadIdx.x; e < K; e += 256) { const float v = DT<T>::ld(x + e); ss = fmaf(v, v, ss); }
    ss = block_sum<4>(ss, red);
    const float rs = 1.0f / sqrtf(ss / (float)K + eps);
    for (int e = threadIdx.x; e < K; e += 256)
        DT<T>::st(y + e, DT<T>::ld(w + e) * DT<T>::rnd(DT<T>::ld(x + e) * rs));
}

// [n_kv][L][128] <-> cache [n_kv][max_seq][128]
template <typename T>
__global__ void kv_copy_kernel(T* dst, const T* src, int L, int dst_stride, int src_stride, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int per = L * kHeadDim;
    const int h = i / per, r = i - h * per;
    dst[(size_t)h * dst_stride + r] = src[(size_t)h * src_stride + r];
}

template <typename T>
__global__ void copy_kernel(T* dst, const T* src, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// ================================================================================================
// On-device decode-loop state (generate.py:149-199 hoisted onto the GPU so the host never syncs
// per frame; the reference does token.item() every frame, generate.py:150).
// ================================================================================================
struct DecodeState {
    int token, frame, pos, gen_step, done;
    int min_new, max_new, trailing_len, noise_frames, eos_id, max_seq, sup_lo, sup_hi;
    float t_temperature; int t_top_k; float t_top_p; int t_do_sample; float t_rep_penalty;
    float p_temperature; int p_top_k; float p_top_p; int p_do_sample;
    const void* trailing_text; const void* tts_pad; const void* talker_noise; const void* pred_noise;
    const void* past_hidden_init;
};

// frame prologue: EOS / limit test, record first-codebook id, history bitmap, predictor input
// [past_hidden ; embed(token)]  (generate.py:150-159)
template <typename T>
__global__ __launch_bounds__(256) void frame_begin_kernel(DecodeState* st, const T* codec_emb, const T* past_hidden,
                                                          T* pred_in, int* codes, unsigned char* seen, int H, int G) {
    if (st->done) return;
    const int tok = st->token;
    if (tok == st->eos_id || st->frame >= st->max_new) {
        __syncthreads();
        if (threadIdx.x == 0) st->done = 1;
        return;
    }
    if (threadIdx.x == 0) { codes[(size_t)st->frame * G] = tok; seen[tok] = 1; }
    for (int e = threadIdx.x; e < H; e += 256) {
        pred_in[e] = past_hidden[e];
        pred_in[H + e] = codec_emb[(size_t)tok * H + e];
    }
}

// 16-way embedding sum + text/pad embed -> talker input (generate.py:162-171); position limit test
// (generate.py:174-177) happens here, after the frame's codes have been recorded.
struct EmbTables { const void* t[32]; };      // [0] = talker codec embedding, [1..G-1] = predictor tables

template <typename T, int G>
__global__ __launch_bounds__(256) void embed_sum_kernel(DecodeState* st, EmbTables tabs, const int* codes, T* x, int H,
                                                        const float* cos_tab, const float* sin_tab, int rope_len,
                                                        int rope_delta, float* rope_now) {
    // one round trip for the state + this frame's 16 ids, one for the 16 embedding rows
    const int done = st->done, frame = st->frame, pos = st->pos, gen_step = st->gen_step;
    const int trailing_len = st->trailing_len, max_seq = st->max_seq;
    const T* trailing = reinterpret_cast<const T*>(st->trailing_text);
    const T* pad = reinterpret_cast<const T*>(st->tts_pad);
    if (done) return;
    if (pos >= max_seq - 1) {
        __syncthreads();
        if (threadIdx.x == 0) { st->frame = frame + 1; st->done = 1; }
        return;
    }
    int ids[G];
#pragma unroll
    for (int gI = 0; gI < G; ++gI) ids[gI] = codes[(size_t)frame * G + gI];
    // RoPE row of this frame's position, staged at a fixed address so the 28 attention launches of
    // the talker step do not have to chase `pos` before they can load it
    if (threadIdx.x < kHeadDim) {
        int rp = pos + rope_delta;
        rp = rp < 0 ? 0 : (rp >= rope_len ? rope_len - 1 : rp);
        const int i = threadIdx.x & 63;
        rope_now[threadIdx.x] = threadIdx.x < 64 ? cos_tab[(size_t)rp * 64 + i] : sin_tab[(size_t)rp * 64 + i];
    }
    const T* text = gen_step < trailing_len ? trailing + (size_t)gen_step * H : pad;
    for (int e0 = threadIdx.x * 8; e0 < H; e0 += 256 * 8) {
        Raw8<T> r[G], tr;
#pragma unroll
        for (int gI = 0; gI < G; ++gI) ldraw<false>(r[gI], reinterpret_cast<const T*>(tabs.t[gI]) + (size_t)ids[gI] * H + e0);
        ldraw<false>(tr, text + e0);
        float sum[8], f[8];
        unpack(r[0], sum);
#pragma unroll
        for (int gI = 1; gI < G; ++gI) {
            unpack(r[gI], f);
#pragma unroll
            for (int i = 0; i < 8; ++i) sum[i] += f[i];
        }
        unpack(tr, f);
#pragma unroll
        for (int i = 0; i < 8; ++i) DT<T>::st(x + e0 + i, DT<T>::rnd(sum[i]) + f[i]);
    }
}

}  // namespace fq3
