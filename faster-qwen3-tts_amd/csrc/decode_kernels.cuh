// Decode-step kernels for the Qwen3-TTS talker / code predictor (gfx950, wave64).
//
// Design: batch-1 decode is a chain of M=1 GEMVs; an empty hipGraph node costs 1.56 us on
// MI355X, a lean GEMV ~2.2-3.3 us, a device-wide barrier inside a persistent kernel 6.6-16 us, so a layer is FIVE
// weight-streaming launches with everything else fused into their prologues/epilogues:
//   1. RMSNorm  -> [q|k|v] GEMV                                   (gemv<PRO_NORM, EPI_STORE>)
//   2. q/k head-RMSNorm + RoPE + KV append + attention             (attn_pred: final output | attn_decode: split-KV)
//   3. [split merge ->] o_proj GEMV -> + residual                  (gemv<PRO_PLAIN | PRO_COMBINE, EPI_RESIDUAL>)
//   4. RMSNorm -> [gate|up] GEMV -> SiLU(gate)*up                  (gemv<PRO_NORM, EPI_SWIGLU>)
//   5. down GEMV -> + residual                                     (gemv<PRO_PLAIN, EPI_RESIDUAL>)
// Per-kernel rules (each one measured with tools/microbench/kernel_chain.hip): input-side loads are issued before
// the weight rows so the prologue arithmetic runs while the weights are in flight; every load is unconditional
// (clamped address) so that s_waitcnt stays exact; lane reductions are DPP / permlane swaps, never ds_bpermute;
// bf16 rounding is v_cvt_pk_bf16_f32.  Rounding points follow the module-by-module Torch execution the reference
// replays (bf16 after every Linear / norm / RoPE / residual add; fp32 inside dot products and softmax).
#pragma once
#include "fq3_common.cuh"

namespace fq3 {

enum { PRO_PLAIN = 0, PRO_NORM = 1, PRO_COMBINE = 2 };
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
    const void* W; int N; int K;                  // N logical rows (pairs for SWIGLU)
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
};

// one thread's share of the split-KV merge: the partial slots of 8 consecutive head dims
struct CombineRegs {
    f32x4 oa[kMaxWorkers], ob[kMaxWorkers];
    float pm[kMaxWorkers], pl[kMaxWorkers];
};
// (all loads unconditional -- slots >= n_part re-read slot 0 and are neutralised afterwards -- because a load inside
// a branch makes the compiler's s_waitcnt bookkeeping fall back to vmcnt(0), which would also wait for the weights)
__device__ __forceinline__ void combine_load(CombineRegs& c, const float* part, int e0, int rep, int n_part) {
    const int head = e0 / kHeadDim, d0 = e0 - head * kHeadDim;
    const int g = head / rep, hh = head - g * rep;
    const float* p0 = part + ((size_t)(g * kMaxWorkers) * rep + hh) * kPartStride;
    const size_t sstride = (size_t)rep * kPartStride;
#pragma unroll
    for (int s = 0; s < kMaxWorkers; ++s) {
        const float* ps = p0 + (s < n_part ? s : 0) * sstride;
        c.oa[s] = *reinterpret_cast<const f32x4*>(ps + d0);
        c.ob[s] = *reinterpret_cast<const f32x4*>(ps + d0 + 4);
        c.pm[s] = ps[kHeadDim]; c.pl[s] = ps[kHeadDim + 1];
    }
}
template <typename T>
__device__ __forceinline__ void combine_finish(CombineRegs& c, int n_part, float (&f)[8]) {
#pragma unroll
    for (int s = 1; s < kMaxWorkers; ++s)
        if (s >= n_part) { c.pm[s] = -1e30f; c.pl[s] = 0.f; }       // weight exp(-1e30 - Mx) = 0 on finite slot-0 data
    float Mx = c.pm[0];
#pragma unroll
    for (int s = 1; s < kMaxWorkers; ++s) Mx = fmaxf(Mx, c.pm[s]);
    f32x4 na = f32x4{0.f, 0.f, 0.f, 0.f}, nb = na;
    float den = 0.f;
#pragma unroll
    for (int s = 0; s < kMaxWorkers; ++s) {
        const float w = __expf(c.pm[s] - Mx);
        na += c.oa[s] * w; nb += c.ob[s] * w; den = fmaf(w, c.pl[s], den);
    }
    const float inv = 1.0f / den;
    f[0] = na.x * inv; f[1] = na.y * inv; f[2] = na.z * inv; f[3] = na.w * inv;
    f[4] = nb.x * inv; f[5] = nb.y * inv; f[6] = nb.z * inv; f[7] = nb.w * inv;
#pragma unroll
    for (int i = 0; i < 8; i += 2) DT<T>::rnd2(f[i], f[i + 1]);
}

// Rows per wave (R, compile time): 2 for the big matrices so that >= 512 workgroups are in flight per launch
// (measured 2.97 vs 3.07 ms/frame against 4 rows per wave), 1 when N <= 1024 or a row is long.
template <int NCH, int EPI> struct MaxRows {
    static constexpr int NR = EPI == EPI_SWIGLU ? 2 : 1;
    static constexpr int v = NCH >= 12 ? 1 : (NCH >= 6 ? (2 / NR) : 2);
};

// M = number of tokens that share one pass over the weights (2 only for the code predictor's two-token
// prefill, predictor_graph.py:121-128: weights are read once, both tokens' dot products are formed).
// Order of work (profiles/r01_kernel_chain.txt): the small input-side loads are issued FIRST, the weight rows
// second, so that the prologue arithmetic (RMSNorm / split-KV merge, ~0.5-1 us of VALU time) runs while the
// weights are still in flight; vmcnt retires in order, so the opposite order exposes it.
template <typename T, int NCH, int PRO, int EPI, bool NT, int M = 1, int R = 1>
__device__ __forceinline__ void gemv_body(const GemvArgs& a) {
    constexpr int NR = (EPI == EPI_SWIGLU) ? 2 : 1;     // physical rows per logical row
    extern __shared__ __attribute__((aligned(16))) float xs[];      // PRO_COMBINE only: M*K floats
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = a.K;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int row0 = (blockIdx.x * 4 + wave) * R;

    // ---- 1. input-side loads (unconditional: out-of-range chunks re-read chunk 0 and are zeroed later) ------
    Raw8<T> xraw[M][NCH], nraw[NCH];
    CombineRegs cr[M];
    if constexpr (PRO == PRO_COMBINE) {
        const int e0 = tid * 8 < K ? tid * 8 : 0;
#pragma unroll
        for (int m = 0; m < M; ++m) combine_load(cr[m], a.part + (size_t)m * a.part_stride2, e0, a.rep, a.n_part);
    } else {
        // every wave reads the whole input vector itself (L2-resident, 2-12 KB): no LDS, no barrier
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
#pragma unroll
            for (int m = 0; m < M; ++m) ldraw<false>(xraw[m][j], reinterpret_cast<const T*>(m == 0 ? a.x : a.x2) + offc);
            if constexpr (PRO == PRO_NORM) ldraw<false>(nraw[j], reinterpret_cast<const T*>(a.norm_w) + offc);
        }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 2. weight rows (tail wave: clamped row, result discarded; short K: clamped chunk, x is zero there) ----
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
    float resv[M][R], biasv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r < a.N ? row0 + r : a.N - 1;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            const T* rp = reinterpret_cast<const T*>(m == 0 ? a.res : a.res2);
            resv[m][r] = 0.f;
            if constexpr (EPI == EPI_RESIDUAL) resv[m][r] = DT<T>::ld(rp + row);
        }
        const T* bp = a.bias ? reinterpret_cast<const T*>(a.bias) + row : W;       // branch-free: always a valid address
        const float bv = DT<T>::ld(bp);
        biasv[r] = a.bias ? bv : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- 3. prologue arithmetic (weights still in flight) --------------------------------------------------
    float xr[M][NCH][8];
    if constexpr (PRO == PRO_COMBINE) {
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float f[8];
            combine_finish<T>(cr[m], a.n_part, f);
            if (tid * 8 < K) {
                float* d = xs + (size_t)m * K + tid * 8;
                reinterpret_cast<f32x4*>(d)[0] = f32x4{f[0], f[1], f[2], f[3]};
                reinterpret_cast<f32x4*>(d)[1] = f32x4{f[4], f[5], f[6], f[7]};
            }
        }
        __syncthreads();
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
                const float z = off < K ? 1.f : 0.f;
                const f32x4 lo = *reinterpret_cast<const f32x4*>(xs + (size_t)m * K + offc) * z;
                const f32x4 hi = *reinterpret_cast<const f32x4*>(xs + (size_t)m * K + offc + 4) * z;
                xr[m][j][0] = lo.x; xr[m][j][1] = lo.y; xr[m][j][2] = lo.z; xr[m][j][3] = lo.w;
                xr[m][j][4] = hi.x; xr[m][j][5] = hi.y; xr[m][j][6] = hi.z; xr[m][j][7] = hi.w;
            }
    } else {
#pragma unroll
        for (int m = 0; m < M; ++m) {
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                if (j * 512 + lane * 8 >= K) zero(xraw[m][j]);
                unpack(xraw[m][j], xr[m][j]);
            }
            if constexpr (PRO == PRO_NORM) {
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
                    for (int i = 0; i < 8; i += 2) {
                        float u = xr[m][j][i] * rs, v = xr[m][j][i + 1] * rs;
                        DT<T>::rnd2(u, v);
                        u *= nw[i]; v *= nw[i + 1];
                        DT<T>::rnd2(u, v);
                        xr[m][j][i] = u; xr[m][j][i + 1] = v;
                    }
                }
            }
        }
    }

    // ---- 4. dot products, all reductions together, epilogue ----------------------------------------------
    float acc[M][R][NR];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int h = 0; h < NR; ++h) {
                float s = 0.f;
#pragma unroll
                for (int j = 0; j < NCH; ++j) s = dot8<T>(raw[h][r][j], xr[m][j], s);
                acc[m][r][h] = s;
            }
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int h = 0; h < NR; ++h) acc[m][r][h] = wave_sum(acc[m][r][h]);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = row0 + r;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float v;
            if constexpr (EPI == EPI_SWIGLU) {
                const float g = DT<T>::rnd(acc[m][r][0]);
                const float u = DT<T>::rnd(acc[m][r][NR - 1]);
                const float sg = DT<T>::rnd(g / (1.0f + expf(-g)));
                v = sg * u;
            } else {
                v = DT<T>::rnd(acc[m][r][0] + biasv[r]);
                if constexpr (EPI == EPI_RESIDUAL) v = v + resv[m][r];
            }
            if (lane == 0 && row < a.N) DT<T>::st(reinterpret_cast<T*>(m == 0 ? a.y : a.y2) + row, v);
        }
    }
    // optional copy of the prologue result (codec_head: the normed hidden is the next frame's past_hidden); last,
    // so that the conditional store does not disturb the waitcnt bookkeeping of the loads above
    if (a.xn_out && blockIdx.x == 0 && wave == 0) {
        T* o = reinterpret_cast<T*>(a.xn_out);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8;
            if (off < K) DT<T>::st8(o + off, xr[0][j]);
        }
    }
}

template <typename T, int NCH, int PRO, int EPI, bool NT, int M = 1, int R = 1>
__global__ __launch_bounds__(256) void gemv_kernel(GemvArgs a) { gemv_body<T, NCH, PRO, EPI, NT, M, R>(a); }

// ================================================================================================
// Attention for one new token over the talker's PAGED KV cache: fixed-size blocks of 64 keys (= one
// attention tile) [n_kv][64][128] drawn from a pool shared by every context of a scheduler, addressed
// through the context's block table (table[key / 64] = block id; fq3_ctx.h).  The code predictor's
// 17-slot cache and the kernel-chain harness keep the contiguous layout [n_kv][max_seq][128]
// (PAGED = false).
// grid = (n_kv_heads, n_workers); worker s walks key tiles s, s+S, ... of 64 keys and ALWAYS writes
// its partial slot (empty = {m=-1e30, l=0}), so the consumer (o_proj prologue) needs no position.
// Prologue (every block, redundantly): per-head RMSNorm + RoPE of this group's q heads and of the
// new k; the worker that owns `pos` appends K/V to the cache.  Only live keys are read (the
// reference reads all max_seq slots under an additive mask, talker_graph.py:71-92).
// A block-table entry is a wave-uniform scalar load; the first tile's entry is fetched first and the
// token-side loads go out while it is in flight, a later tile's entry one tile ahead.
// ================================================================================================
struct AttnArgs {
    const void* qkv;
    const void* q_norm_w; const void* k_norm_w; float eps;
    const float* cos_row; const float* sin_row;       // [64] each: RoPE row of this token's position
    void* kcache; void* vcache; int max_seq;          // contiguous: [n_kv][max_seq][128]; paged: the layer's block pool [n_blocks][n_kv][64][128]
    const int* table; int blk_stride;                 // paged: block table of the context, elements per block (n_kv * 64 * 128)
    const int* pos_ptr; int pos_imm; int n_pad;
    const int* done_ptr;                              // fused loop: the context's DecodeState::done -- a loop that is done never appends (null: API calls)
    int n_kv; float* part; float scale;
    int rep; void* out;                               // attn_pred_kernel: q heads per kv head, final output T[q_dim]
};

template <typename T, int REP, bool PAGED>
__device__ __forceinline__ void attn_decode_body(const AttnArgs& a) {
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
    // contiguous: this kv head's rows; paged: this kv head's 64 rows inside block 0 (a block id adds blk_stride elements)
    T* kc = reinterpret_cast<T*>(a.kcache) + (PAGED ? (size_t)g * KS * HD : (size_t)g * a.max_seq * HD);
    T* vc = reinterpret_cast<T*>(a.vcache) + (PAGED ? (size_t)g * KS * HD : (size_t)g * a.max_seq * HD);
    const int n_tiles = (a.max_seq + KS - 1) / KS;
    auto block_of = [&](int tile) -> int {            // wave-uniform: a scalar load
        if constexpr (PAGED) return a.table[tile < n_tiles ? tile : n_tiles - 1];
        else return 0;
    };
    int blk = block_of(s), blk_next = block_of(s + S);

    // ---- issue: first key tile (unconditionally, clamped), q/k/v rows, gains, rope row, position ------
    Raw8<T> kr[NG], vr[NG];
    auto issue_tile = [&](int tile, int b) {
#pragma unroll
        for (int i = 0; i < NG; ++i) {
            const int kt = (i * 4 + wave) * 4 + sub;              // key inside the tile
            size_t off;
            if constexpr (PAGED) off = (size_t)b * a.blk_stride + (size_t)kt * HD;      // (a block is 64 full rows: no clamp)
            else { int key = tile * KS + kt; key = key < a.max_seq ? key : a.max_seq - 1; off = (size_t)key * HD; }
            ldraw<false>(kr[i], kc + off + c * 8);
            ldraw<false>(vr[i], vc + off + c * 8);
        }
    };
    if constexpr (!PAGED) issue_tile(s, 0);
    // prologue vectors: wave w handles vec w (and w+4 when REP == 4): q heads, then k, then v
    constexpr int NV = (REP + 2 + 3) / 4;
    float px0[NV], px1[NV], pw0[NV], pw1[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        // unconditional loads (a surplus wave re-reads the v row and drops it): a load inside a branch makes the
        // compiler fall back to s_waitcnt vmcnt(0) right there, which also waits for the key tile above
        const int vec = wave + 4 * i < REP + 2 ? wave + 4 * i : REP + 1;
        const T* src = vec < REP ? qkv + (size_t)(g * REP + vec) * HD
                     : (vec == REP ? qkv + q_dim + (size_t)g * HD : qkv + q_dim + kv_dim + (size_t)g * HD);
        px0[i] = DT<T>::ld(src + lane); px1[i] = DT<T>::ld(src + lane + 64);
        const T* w = reinterpret_cast<const T*>(vec < REP ? a.q_norm_w : a.k_norm_w);
        pw0[i] = DT<T>::ld(w + lane); pw1[i] = DT<T>::ld(w + lane + 64);
    }
    const float cs = a.cos_row[lane], sn = a.sin_row[lane];
    const int pos = a.pos_ptr ? *a.pos_ptr : a.pos_imm;
    // A context whose loop is DONE (EOS / limits / cancelled) keeps being launched by frames that were queued ahead; its blocks may have
    // gone back to the pool (fq3_kv_release, fq3_kv_adopt) and belong to somebody else by now, so it must not append (round-4 advisor;
    // the lock-step batch has the same guard in attn_decode_batch_kernel).  Branch-free: an unconditional load from a valid address.
    const int done_word = *(a.done_ptr ? a.done_ptr : reinterpret_cast<const int*>(a.cos_row));
    const bool loop_done = a.done_ptr != nullptr && done_word != 0;
    if constexpr (PAGED) issue_tile(s, blk);     // after the token-side loads: they flew while the table entry arrived
    __builtin_amdgcn_sched_barrier(0);           // keep every load above the arithmetic (one round trip, not two)

    const int t_pos = pos / KS;
    const bool owner = (t_pos % S) == s;
    const bool may_append = owner && !loop_done;
    // where the new K / V row goes (the owner only): contiguous slot `pos`, or slot pos % 64 of the block of tile t_pos
    size_t new_off = (size_t)pos * HD;
    if constexpr (PAGED) new_off = (size_t)block_of(t_pos) * a.blk_stride + (size_t)(pos - t_pos * KS) * HD;
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
            if (may_append) {
                T* cp = (vec == REP ? kc : vc) + new_off;
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
            sc = row16_sum(sc);
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
        if (tile != s) { blk = blk_next; issue_tile(tile, blk); }
        blk_next = block_of(tile + S);           // the next tile's table entry is in flight while this tile is multiplied
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
    // merge the four 16-lane key groups of the wave (lane-wise across rows, VALU permlane swaps)
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        const float M = xrow_max(m[h]);
        const float w = __expf(m[h] - M);
        l[h] = xrow_sum(l[h] * w);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[h][i] = xrow_sum(o[h][i] * w);
        m[h] = M;
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

template <typename T, int REP, bool PAGED>
__global__ __launch_bounds__(256) void attn_decode_kernel(AttnArgs a) { attn_decode_body<T, REP, PAGED>(a); }

// ================================================================================================
// Code-predictor attention (context <= 17 keys): ONE wave per q head, everything in registers, no LDS, no
// barrier, no partial slots -- the wave writes the final (normalised, T-rounded) head output, so the o_proj
// GEMV that follows needs no merge prologue.  Lane layout: 4 rows x 16 lanes; a row holds a full 128-dim
// vector (8 dims per lane); row r scores cached keys r, 4+r, 8+r, 12+r and row 0 also the new token's own key.
// Exact (non-online) softmax: wave-wide max, then one exp per key.  Both q heads of a GQA group recompute the
// new key's norm + RoPE; the group's first head appends K/V to the cache.
// ================================================================================================
template <typename T>
__device__ __forceinline__ void attn_pred_body(const AttnArgs& a, int head) {
    constexpr int HD = kHeadDim;
    const int g = head / a.rep, hh = head - g * a.rep;
    const int lane = threadIdx.x & 63, sub = lane >> 4, c = lane & 15;
    const int pos = a.pos_imm;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const int q_dim = a.n_kv * a.rep * HD, kv_dim = a.n_kv * HD;
    T* kc = reinterpret_cast<T*>(a.kcache) + (size_t)g * a.max_seq * HD;
    T* vc = reinterpret_cast<T*>(a.vcache) + (size_t)g * a.max_seq * HD;
    Raw8<T> qraw, knraw, vnraw, qwraw, kwraw, kr[4], vr[4];
    ldraw<false>(qraw, qkv + (size_t)head * HD + c * 8);
    ldraw<false>(knraw, qkv + q_dim + (size_t)g * HD + c * 8);
    ldraw<false>(vnraw, qkv + q_dim + kv_dim + (size_t)g * HD + c * 8);
    ldraw<false>(qwraw, reinterpret_cast<const T*>(a.q_norm_w) + c * 8);
    ldraw<false>(kwraw, reinterpret_cast<const T*>(a.k_norm_w) + c * 8);
    const f32x4 cs0 = *reinterpret_cast<const f32x4*>(a.cos_row + ((c * 8) & 63)), cs1 = *reinterpret_cast<const f32x4*>(a.cos_row + ((c * 8 + 4) & 63));
    const f32x4 sn0 = *reinterpret_cast<const f32x4*>(a.sin_row + ((c * 8) & 63)), sn1 = *reinterpret_cast<const f32x4*>(a.sin_row + ((c * 8 + 4) & 63));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int key = i * 4 + sub;
        key = key < a.max_seq ? key : a.max_seq - 1;
        ldraw<false>(kr[i], kc + (size_t)key * HD + c * 8);
        ldraw<false>(vr[i], vc + (size_t)key * HD + c * 8);
    }
    __builtin_amdgcn_sched_barrier(0);           // keep every load above the arithmetic (one round trip, not two)
    const float sgn = c < 8 ? -1.f : 1.f;        // rotate_half: dims < 64 take -x[d + 64], dims >= 64 take +x[d - 64]
    const float cs[8] = {cs0.x, cs0.y, cs0.z, cs0.w, cs1.x, cs1.y, cs1.z, cs1.w};
    const float sn[8] = {sgn * sn0.x, sgn * sn0.y, sgn * sn0.z, sgn * sn0.w, sgn * sn1.x, sgn * sn1.y, sgn * sn1.z, sgn * sn1.w};
    // head RMSNorm + RoPE; the rotate_half partner of dim d (d +- 64) lives in lane c ^ 8 of the same row
    auto norm_rope = [&](const Raw8<T>& raw, const Raw8<T>& wraw, float (&out)[8]) {
        float x[8], w[8];
        unpack(raw, x); unpack(wraw, w);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss = fmaf(x[i], x[i], ss);
        ss = row16_sum(ss);
        const float rs = 1.0f / sqrtf(ss / (float)HD + a.eps);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            float u = x[i] * rs, v = x[i + 1] * rs;
            DT<T>::rnd2(u, v);
            u *= w[i]; v *= w[i + 1];
            DT<T>::rnd2(u, v);
            x[i] = u; x[i + 1] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            float a0 = x[i] * cs[i], a1 = x[i + 1] * cs[i + 1];
            float b0 = row16_xor8(x[i]) * sn[i], b1 = row16_xor8(x[i + 1]) * sn[i + 1];
            DT<T>::rnd2(a0, a1); DT<T>::rnd2(b0, b1);
            float r0 = a0 + b0, r1 = a1 + b1;
            DT<T>::rnd2(r0, r1);
            out[i] = r0; out[i + 1] = r1;
        }
    };
    float qr[8], knew[8], vnew[8];
    norm_rope(qraw, qwraw, qr);
    norm_rope(knraw, kwraw, knew);
    unpack(vnraw, vnew);
    if (hh == 0 && sub == 0) {
        DT<T>::st8(kc + (size_t)pos * HD + c * 8, knew);
        DT<T>::st8(vc + (size_t)pos * HD + c * 8, vnew);
    }
    float sc[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float kf[8], s = 0.f;
        unpack(kr[i], kf);
#pragma unroll
        for (int d = 0; d < 8; ++d) s = fmaf(qr[d], kf[d], s);
        s = row16_sum(s);
        sc[i] = (i * 4 + sub < pos) ? s * a.scale : -INFINITY;
    }
    {
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) s = fmaf(qr[d], knew[d], s);
        s = row16_sum(s);
        sc[4] = sub == 0 ? s * a.scale : -INFINITY;
    }
    const float mx = wave_max(fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), sc[4]));
    float p[5], lsum = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 5; ++i) { p[i] = __expf(sc[i] - mx); lsum += p[i]; }
    lsum = wave_sum(lsum) * (1.0f / 16.0f);          // every lane of a row carries the row's sum
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = p[4] * vnew[d];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float vf[8];
        if (!(i * 4 + sub < pos)) zero(vr[i]);      // never-written cache slots may hold NaN bit patterns: 0 * NaN
        unpack(vr[i], vf);
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = fmaf(p[i], vf[d], o[d]);
    }
    const float inv = 1.0f / lsum;
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = xrow_sum(o[d]) * inv;
    if (sub == 0) DT<T>::st8(reinterpret_cast<T*>(a.out) + (size_t)head * HD + c * 8, o);
}

template <typename T>
__device__ __forceinline__ void attn_pred_body(const AttnArgs& a) { attn_pred_body<T>(a, (int)blockIdx.x); }

// The same attention for the lock-step batch (round 6): ONE wave per kv group serves the group's REP q heads from ONE read of the
// group's K / V rows, and only LIVE rows are requested -- a slot >= pos takes slot 0's address (its score is -inf and its V row is
// zeroed, exactly as for the unclamped read).  With one wave per q head (attn_pred_body) the two heads of a group sit in different
// workgroups -- on different XCDs -- and every wave requests all 16 slots whatever the position: 18.0 MB per launch at 128 lanes
// against 8.9 MB of cache and ~4.5 MB of live rows (profiles/r05_pmc_batch128_fetch.txt: 1.44 GB of the 9.9 GB of a frame).
// Every head keeps the instructions of attn_pred_body in their order (the new key's norm + RoPE is computed once: the same
// values), so the outputs are bit-identical (tests/test_gpu_batch.py::test_pred_attention_group_form_is_bit_identical).
template <typename T, int REP>
__device__ __forceinline__ void attn_pred_group_body(const AttnArgs& a, int g) {
    constexpr int HD = kHeadDim;
    const int lane = threadIdx.x & 63, sub = lane >> 4, c = lane & 15;
    const int pos = a.pos_imm;
    const T* qkv = reinterpret_cast<const T*>(a.qkv);
    const int q_dim = a.n_kv * REP * HD, kv_dim = a.n_kv * HD;
    T* kc = reinterpret_cast<T*>(a.kcache) + (size_t)g * a.max_seq * HD;
    T* vc = reinterpret_cast<T*>(a.vcache) + (size_t)g * a.max_seq * HD;
    Raw8<T> qraw[REP], knraw, vnraw, qwraw, kwraw, kr[4], vr[4];
#pragma unroll
    for (int h = 0; h < REP; ++h) ldraw<false>(qraw[h], qkv + (size_t)(g * REP + h) * HD + c * 8);
    ldraw<false>(knraw, qkv + q_dim + (size_t)g * HD + c * 8);
    ldraw<false>(vnraw, qkv + q_dim + kv_dim + (size_t)g * HD + c * 8);
    ldraw<false>(qwraw, reinterpret_cast<const T*>(a.q_norm_w) + c * 8);
    ldraw<false>(kwraw, reinterpret_cast<const T*>(a.k_norm_w) + c * 8);
    const f32x4 cs0 = *reinterpret_cast<const f32x4*>(a.cos_row + ((c * 8) & 63)), cs1 = *reinterpret_cast<const f32x4*>(a.cos_row + ((c * 8 + 4) & 63));
    const f32x4 sn0 = *reinterpret_cast<const f32x4*>(a.sin_row + ((c * 8) & 63)), sn1 = *reinterpret_cast<const f32x4*>(a.sin_row + ((c * 8 + 4) & 63));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int key = i * 4 + sub;
        key = key < pos ? key : 0;               // a dead slot: slot 0's line again (no new request), masked below
        ldraw<false>(kr[i], kc + (size_t)key * HD + c * 8);
        ldraw<false>(vr[i], vc + (size_t)key * HD + c * 8);
    }
    __builtin_amdgcn_sched_barrier(0);           // keep every load above the arithmetic (one round trip, not two)
    const float sgn = c < 8 ? -1.f : 1.f;
    const float cs[8] = {cs0.x, cs0.y, cs0.z, cs0.w, cs1.x, cs1.y, cs1.z, cs1.w};
    const float sn[8] = {sgn * sn0.x, sgn * sn0.y, sgn * sn0.z, sgn * sn0.w, sgn * sn1.x, sgn * sn1.y, sgn * sn1.z, sgn * sn1.w};
    auto norm_rope = [&](const Raw8<T>& raw, const Raw8<T>& wraw, float (&out)[8]) {
        float x[8], w[8];
        unpack(raw, x); unpack(wraw, w);
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) ss = fmaf(x[i], x[i], ss);
        ss = row16_sum(ss);
        const float rs = 1.0f / sqrtf(ss / (float)HD + a.eps);
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            float u = x[i] * rs, v = x[i + 1] * rs;
            DT<T>::rnd2(u, v);
            u *= w[i]; v *= w[i + 1];
            DT<T>::rnd2(u, v);
            x[i] = u; x[i + 1] = v;
        }
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            float a0 = x[i] * cs[i], a1 = x[i + 1] * cs[i + 1];
            float b0 = row16_xor8(x[i]) * sn[i], b1 = row16_xor8(x[i + 1]) * sn[i + 1];
            DT<T>::rnd2(a0, a1); DT<T>::rnd2(b0, b1);
            float r0 = a0 + b0, r1 = a1 + b1;
            DT<T>::rnd2(r0, r1);
            out[i] = r0; out[i + 1] = r1;
        }
    };
    float knew[8], vnew[8];
    norm_rope(knraw, kwraw, knew);
    unpack(vnraw, vnew);
    if (sub == 0) {
        DT<T>::st8(kc + (size_t)pos * HD + c * 8, knew);
        DT<T>::st8(vc + (size_t)pos * HD + c * 8, vnew);
    }
    float kf[4][8], vf[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unpack(kr[i], kf[i]);
        if (!(i * 4 + sub < pos)) zero(vr[i]);      // dead slots: another row's (or never-written) bits
        unpack(vr[i], vf[i]);
    }
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        float qr[8];
        norm_rope(qraw[h], qwraw, qr);
        float sc[5];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s = fmaf(qr[d], kf[i][d], s);
            s = row16_sum(s);
            sc[i] = (i * 4 + sub < pos) ? s * a.scale : -INFINITY;
        }
        {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s = fmaf(qr[d], knew[d], s);
            s = row16_sum(s);
            sc[4] = sub == 0 ? s * a.scale : -INFINITY;
        }
        const float mx = wave_max(fmaxf(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])), sc[4]));
        float p[5], lsum = 0.f, o[8];
#pragma unroll
        for (int i = 0; i < 5; ++i) { p[i] = __expf(sc[i] - mx); lsum += p[i]; }
        lsum = wave_sum(lsum) * (1.0f / 16.0f);
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = p[4] * vnew[d];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int d = 0; d < 8; ++d) o[d] = fmaf(p[i], vf[i][d], o[d]);
        const float inv = 1.0f / lsum;
#pragma unroll
        for (int d = 0; d < 8; ++d) o[d] = xrow_sum(o[d]) * inv;
        if (sub == 0) DT<T>::st8(reinterpret_cast<T*>(a.out) + (size_t)(g * REP + h) * HD + c * 8, o);
    }
}

template <typename T>
__global__ __launch_bounds__(64) void attn_pred_kernel(AttnArgs a) { attn_pred_body<T>(a); }

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
    int n_pad, rope_delta;        // copies of the context's generation state for the batched frame (fq3_batch.hip)
};

// Teacher forcing (parity-test hook, fq3_decode_set_forced) lives OUTSIDE the loop state: a context that was never asked for it
// has no such object and its samplers receive a null pointer (the captured graph holds that null).  When `forced` is set every
// sampler records its own decision in `decisions` and the loop continues with the forced id instead.  Both int[frames + 1][G];
// slot [f][0] = first-codebook id of frame f, [f][1 + cb] = predictor codebook cb of frame f.
struct TeacherForcing { const int* forced; int* decisions; };

// sampler epilogue of the teacher-forcing hook: returns the id the loop continues with (uniform across the block)
__device__ __forceinline__ int forced_or(const TeacherForcing* tf, int slot, int tok) {
    if (!tf) return tok;
    const int* forced = gptr(tf->forced); int* dec = gptr(tf->decisions);
    if (dec && threadIdx.x == 0) dec[slot] = tok;
    return forced ? forced[slot] : tok;
}

// frame prologue: EOS / limit test, record first-codebook id, history bitmap, predictor input
// [past_hidden ; embed(token)]  (generate.py:150-159)
template <typename T>
__device__ __forceinline__ void frame_begin_body(DecodeState* st, const T* codec_emb, const T* past_hidden,
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
template <typename T>
__global__ __launch_bounds__(256) void frame_begin_kernel(DecodeState* st, const T* codec_emb, const T* past_hidden,
                                                          T* pred_in, int* codes, unsigned char* seen, int H, int G) {
    frame_begin_body<T>(st, codec_emb, past_hidden, pred_in, codes, seen, H, G);
}

// 16-way embedding sum + text/pad embed -> talker input (generate.py:162-171); position limit test
// (generate.py:174-177) happens here, after the frame's codes have been recorded.
struct EmbTables { const void* t[32]; };      // [0] = talker codec embedding, [1..G-1] = predictor tables

template <typename T, int G>
__device__ __forceinline__ void embed_sum_body(DecodeState* st, const EmbTables& tabs, const int* codes, T* x, int H,
                                               const float* cos_tab, const float* sin_tab, int rope_len,
                                               int rope_delta, float* rope_now) {
    // one round trip for the state + this frame's 16 ids, one for the 16 embedding rows
    const int done = st->done, frame = st->frame, pos = st->pos, gen_step = st->gen_step;
    const int trailing_len = st->trailing_len, max_seq = st->max_seq;
    const T* trailing = gptr(reinterpret_cast<const T*>(st->trailing_text));     // (loaded pointers: global memory, fq3_common.cuh::gptr)
    const T* pad = gptr(reinterpret_cast<const T*>(st->tts_pad));
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
template <typename T, int G>
__global__ __launch_bounds__(256) void embed_sum_kernel(DecodeState* st, EmbTables tabs, const int* codes, T* x, int H,
                                                        const float* cos_tab, const float* sin_tab, int rope_len,
                                                        int rope_delta, float* rope_now) {
    embed_sum_body<T, G>(st, tabs, codes, x, H, cos_tab, sin_tab, rope_len, rope_delta, rope_now);
}

}  // namespace fq3
