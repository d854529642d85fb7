// Matrix-core prefill of the talker (replaces the upstream eager `talker.forward(inputs_embeds=...)`,
// /root/reference/faster_qwen3_tts/generate.py:107-122): all prompt rows go through each layer as GEMMs
// on MFMA (conv_gemm_kernel with one tap), K/V are normalised, rotated and written straight into the
// static cache layout the decode kernels read, attention is causal over the live (non-padded) keys.
// Rounding points are those of the decode path (one rounding to T per Linear / norm / RoPE / residual add),
// so prefill + decode match the oracle's matrix-form prefill.
#define FQ3_SKINNY_DEFINE           // skinny_gemm.cuh: the weight-stationary GEMM kernels are instantiated in fq3_prefill.hip only
#include "fq3_ctx.h"
#include <vector>
#include "codec_kernels.cuh"

using namespace fq3;

namespace {

// One layer of a context's paged talker cache (fq3_ctx.h): pool arrays [n_blocks][n_kv][64][128], the block table, elements per block.
// Row `key` of kv head g: base + table[key / 64] * blk_stride + (g * 64 + key % 64) * 128.
template <typename T> struct PagedKV { T* k; T* v; const int* table; int blk_stride; };
template <typename T> PagedKV<T> paged_kv(const fq3_ctx* c, int layer) {
    return PagedKV<T>{(T*)c->tk.k[layer], (T*)c->tk.v[layer], c->tk.d_table, (int)c->tk.pool->blk_elems};
}
template <typename T>
__device__ __forceinline__ size_t paged_row(const PagedKV<T>& kv, int g, int key) {
    return (size_t)kv.table[key / kKeysPerTile] * kv.blk_stride + ((size_t)g * kKeysPerTile + key % kKeysPerTile) * kHeadDim;
}

// per (token, head): RMSNorm over 128 + RoPE for q (in place) and k (-> cache); v copied to the cache
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kv_kernel(T* qkv, const T* qw, const T* kw, float eps, const float* cos_tab,
                                                              const float* sin_tab, int rope_len, int rope_delta, PagedKV<T> kv,
                                                              int L, int n_pad, int NH, int NKV) {
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
    T* dst = v < NH ? src : (v < NH + NKV ? kv.k + paged_row(kv, v - NH, t) : kv.v + paged_row(kv, v - NH - NKV, t));
    DT<T>::st(dst + lane, x0);
    DT<T>::st(dst + lane + 64, x1);
}

// causal attention for the prompt: one wave per (query row, q head); 16 lanes per key (8 dims each),
// 16 keys in flight per trip; fp32 online softmax, one rounding at the end.
template <typename T>
__global__ __launch_bounds__(256) void prefill_attn_kernel(const T* qkv, PagedKV<T> kv, T* out,
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
    float m = -1e30f, l = 0.f, o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = 0.f;
    for (int k0 = n_pad; k0 <= t; k0 += 16) {
        Raw8<T> kr[4], vr[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int key = k0 + i * 4 + sub;
            key = key <= t ? key : t;
            const size_t off = paged_row(kv, g, key) + c * 8;
            ldraw<false>(kr[i], kv.k + off);
            ldraw<false>(vr[i], kv.v + off);
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

// ---------------------------------------------------------------------------------------------------------------------
// Flash-style causal attention on the matrix cores (bf16 contexts; long prompts: BASELINE configs[4], 4k tokens).
// One workgroup = one q head x 64 queries (16 per wave); key tiles of 64 keys stream through LDS (K row-major, V
// TRANSPOSED so that both MFMAs read 16-byte fragments); S = Q K^T and O += P V on v_mfma_f32_16x16x32_bf16, online
// softmax in fp32 on the accumulator layout (row statistics by DPP inside the 16-lane groups).  The probabilities enter
// the second MFMA as bf16, which the oracle's fp32 softmax(QK^T) V does not round: P is therefore split into a bf16 high
// part and a bf16 residual (two MFMAs), leaving a 2^-16 relative error instead of 2^-9 -- below the one rounding to T of
// the output.  The next tile's K/V global loads are issued before the current tile's arithmetic (register staging).
// fp32 contexts keep prefill_attn_kernel (exact fp32 products).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kFaQ = 64, kFaK = 64, kFaKLd = kHeadDim + 8, kFaVLd = kFaK + 8, kFaPLd = kFaK + 8;

// One 64-key tile of the flash-style attention for ONE wave (16 queries): S = Q K^T, mask + online softmax on the accumulator layout, P
// (bf16 high part + residual) through the wave's private LDS region, O += P V.  Shared by flash_prefill_kernel (tiles streamed through one
// LDS stage) and flash_prefill_small_kernel (every tile resident): the same instructions on the same values in the same order.
__device__ __forceinline__ void flash_tile(const bf16_t* Ks, const bf16_t* Vt, bf16_t* ph, bf16_t* pl, const bf16x8_t (&qf)[4], f32x4_t (&o)[8],
                                           float (&m)[4], float (&l)[4], int tile, int q0, int wave, int fr, int fq, int n_pad, float sl2) {
        // ---- S = Q K^T: 4 key blocks of 16 ----
        f32x4_t sacc[4];
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) {
            sacc[nb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(&Ks[(nb * 16 + fr) * kFaKLd + ks * 32 + fq * 8]);
                sacc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kf, sacc[nb], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (accumulator layout: column = key nb*16 + fr, row = query fq*4 + r) ----
        float p[4][4], alpha[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qi = q0 + wave * 16 + fq * 4 + r;
            float mx = -1e30f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int key = tile * kFaK + nb * 16 + fr;
                const bool ok = key <= qi && key >= n_pad;
                const float v = ok ? sacc[nb][r] * sl2 : -1e30f;
                p[nb][r] = v;
                mx = fmaxf(mx, v);
            }
            mx = row16_max(mx);
            const float mn = fmaxf(m[r], mx);
            alpha[r] = exp2f(m[r] - mn);
            float rs = 0.f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const float e = p[nb][r] > -1e29f ? exp2f(p[nb][r] - mn) : 0.f;
                p[nb][r] = e;
                rs += e;
            }
            rs = row16_sum(rs);
            l[r] = l[r] * alpha[r] + rs;
            m[r] = mn;
        }
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[d][r] *= alpha[r];
        // ---- P -> LDS as bf16 high part + bf16 residual, [query][key] (this wave's private region) ----
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = p[nb][r];
                const bf16_t hi = f_to_bf16(v);
                ph[(fq * 4 + r) * kFaPLd + nb * 16 + fr] = hi;
                pl[(fq * 4 + r) * kFaPLd + nb * 16 + fr] = f_to_bf16(v - bf16_to_f(hi));
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);                               // lgkmcnt(0): the wave's own LDS writes have landed
        __builtin_amdgcn_wave_barrier();
        // ---- O += P V: A = P [query fr][keys fq*8 + 32 ks], B = V^T [dim db*16 + fr][keys fq*8 + 32 ks] ----
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t pah = *reinterpret_cast<const bf16x8_t*>(&ph[fr * kFaPLd + ks * 32 + fq * 8]);
            const bf16x8_t pal = *reinterpret_cast<const bf16x8_t*>(&pl[fr * kFaPLd + ks * 32 + fq * 8]);
#pragma unroll
            for (int d = 0; d < 8; ++d) {
                const bf16x8_t vf = *reinterpret_cast<const bf16x8_t*>(&Vt[(d * 16 + fr) * kFaVLd + ks * 32 + fq * 8]);
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pah, vf, o[d], 0, 0, 0);
                o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pal, vf, o[d], 0, 0, 0);
            }
        }
}

// NW waves = 16 * NW queries per block (4: 64 queries, the short-prompt shape; 8: 128 queries -- a staged K/V tile feeds twice the
// MFMA work).  PAIRED: the workgroup handles query block bx and then block nqb - 1 - bx, so that under the causal mask every
// workgroup walks the same number of key tiles (nqb + 1) instead of 1 .. nqb of them: a 4096-token prompt at 128 queries per block
// is 16 pairs x 16 heads = 256 equal workgroups, one per CU.
template <int NW, bool PAIRED>
__global__ __launch_bounds__(64 * NW) void flash_prefill_kernel(const bf16_t* qkv, PagedKV<bf16_t> kv, bf16_t* out,
                                                                int L, int n_pad, int NH, int NKV, float scale, int nqb) {
    constexpr int HD = kHeadDim, Q = 16 * NW, CPT = 16 / NW;             // CPT: 16-byte chunks of a K (and V) row staged per thread
    __shared__ __attribute__((aligned(16))) bf16_t Ks[kFaK * kFaKLd];            // [key][dim]
    __shared__ __attribute__((aligned(16))) bf16_t Vt[HD * kFaVLd];              // [dim][key]
    __shared__ __attribute__((aligned(16))) bf16_t Ps[NW][2][16 * kFaPLd];       // per wave: P high / residual, [query][key]
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, g = h / (NH / NKV), per = NH + 2 * NKV;
    // a key tile IS a block of the paged cache: tile t of kv head g = 64 contiguous rows at block table[t]
    const bf16_t* kc = kv.k + (size_t)g * kFaK * HD;
    const bf16_t* vc = kv.v + (size_t)g * kFaK * HD;
    const float sl2 = scale * 1.4426950408889634f;                        // softmax in base 2: exp(x) = exp2(x * log2 e)
    // staging: thread -> key (tid & 63), 16-byte chunks (tid >> 6) + NW j of that key's K and V rows
    const int skey = tid & 63, sch = tid >> 6;
    for (int pass = 0; pass < (PAIRED ? 2 : 1); ++pass) {
    const int qb = pass == 0 ? (int)blockIdx.x : nqb - 1 - (int)blockIdx.x;
    if (pass == 1 && qb <= (int)blockIdx.x) break;                        // odd block count: the middle block was pass 0
    const int q0 = qb * Q;
    if (q0 >= L) continue;
    // Q fragments of this wave's 16 rows (A operand: row = fr, dims fq*8 + 32*ks), kept in registers
    const int qrow = q0 + wave * 16 + fr;
    const int qrc = qrow < L ? qrow : L - 1;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8_t*>(qkv + (size_t)qrc * per * HD + (size_t)h * HD + ks * 32 + fq * 8);
    f32x4_t o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -1e30f; l[r] = 0.f; }
    const int q_hi = min(q0 + Q, L) - 1;                                  // last query of the block
    const int t_lo = n_pad / kFaK, t_hi = q_hi / kFaK;                    // key tiles [t_lo, t_hi]
    u32x4 kst[CPT], vst[CPT];
    auto issue = [&](int tile) {
        const size_t off = (size_t)kv.table[tile] * kv.blk_stride + (size_t)skey * HD;       // tile <= t_hi: a block this prompt owns
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            kst[j] = *reinterpret_cast<const u32x4*>(kc + off + (sch + NW * j) * 8);
            vst[j] = *reinterpret_cast<const u32x4*>(vc + off + (sch + NW * j) * 8);
        }
    };
    issue(t_lo);
    for (int tile = t_lo; tile <= t_hi; ++tile) {
        __syncthreads();                                                  // everyone is done reading the previous tile
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            *reinterpret_cast<u32x4*>(&Ks[skey * kFaKLd + (sch + NW * j) * 8]) = kst[j];
            // transposed store (dim-major image of V), two keys per 32-bit write: lanes 2i / 2i + 1 hold keys k / k + 1; they swap
            // words (DPP quad_perm [1,0,3,2]), the even lane writes the even dim of each pair for both keys, the odd lane the odd dim
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t mine = vst[j][w];
                const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);
                const bool odd = skey & 1;
                const uint32_t word = odd ? ((other >> 16) | (mine & 0xFFFF0000u)) : ((mine & 0xFFFFu) | (other << 16));
                const int dim = (sch + NW * j) * 8 + 2 * w + (odd ? 1 : 0);
                *reinterpret_cast<uint32_t*>(&Vt[dim * kFaVLd + (skey & ~1)]) = word;
            }
        }
        __syncthreads();
        if (tile < t_hi) issue(tile + 1);                                 // next tile's loads fly under the MFMAs
        flash_tile(Ks, Vt, Ps[wave][0], Ps[wave][1], qf, o, m, l, tile, q0, wave, fr, fq, n_pad, sl2);
    }
    // ---- normalise, one rounding, store (left-padded query rows are zeros, like prefill_attn_kernel) ----
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + wave * 16 + fq * 4 + r;
        if (qi >= L) continue;
        const float inv = (qi >= n_pad && l[r] > 0.f) ? 1.0f / l[r] : 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) out[((size_t)qi * NH + h) * HD + d * 16 + fr] = f_to_bf16(o[d][r] * inv);
    }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Short prompts (round 5): every sequence has at most 256 rows = four key tiles, and ALL of a query block's key tiles fit the LDS at
// once (4 x (K 17 KB + V^T 18 KB) + the waves' P regions = 158 KB).  flash_prefill_kernel walks the tiles one after the other through a
// single stage -- global loads, two workgroup barriers and the tile's arithmetic in series per tile: 14.8 us per layer at 200 rows
// (profiles/r03_prefill200_kernel_trace.txt), a four-deep latency chain.  Here the tiles are staged back to back (two register sets,
// the loads of tile i + 1 in flight while tile i is written; unconditional, clamped), ONE barrier, then every wave walks the resident
// tiles on its own (flash_tile: the same instructions in the same order, so the output is bit-identical to flash_prefill_kernel).
// The sequences of a PACKED prefill (fq3_prefill_batch) share the launch: blockIdx.z = sequence, its rows / pad / block table from a
// by-value table -- one launch per layer instead of one per prompt and layer (280 launches per 10-prompt group before).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int kMaxPack = 64;                        // fq3_prefill_batch takes at most 64 prompts
constexpr int kFsTiles = 4, kFsMaxRows = kFsTiles * kFaK;
struct PackSeq {
    const int* table[kMaxPack];                     // block table of each sequence's context (all of ONE pool)
    int off[kMaxPack + 1];                          // first packed row of each sequence
    int n_pad[kMaxPack];
    int rope_delta[kMaxPack];
    int n;
};
constexpr size_t kFsLdsBytes = (size_t)kFsTiles * (kFaK * kFaKLd + kHeadDim * kFaVLd) * 2 + (size_t)4 * 2 * 16 * kFaPLd * 2;

__global__ __launch_bounds__(256) void flash_prefill_small_kernel(const bf16_t* qkv_all, PagedKV<bf16_t> kv, bf16_t* out_all, PackSeq sq,
                                                                  int NH, int NKV, float scale) {
    constexpr int HD = kHeadDim, NW = 4, Q = 64, CPT = 16 / NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char fs_smem[];
    bf16_t* KsAll = reinterpret_cast<bf16_t*>(fs_smem);                                   // [tile][key][dim]
    bf16_t* VtAll = KsAll + (size_t)kFsTiles * kFaK * kFaKLd;                               // [tile][dim][key]
    bf16_t* PsAll = VtAll + (size_t)kFsTiles * HD * kFaVLd;                                 // [wave][hi | lo][query][key]
    const int tid = threadIdx.x, lane = tid & 63, fr = lane & 15, fq = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sqi = blockIdx.z;
    const int off = sq.off[sqi], L = sq.off[sqi + 1] - off, n_pad = sq.n_pad[sqi];
    const int q0 = (int)blockIdx.x * Q;
    if (q0 >= L) return;
    const int* table = sq.table[sqi];
    const int h = blockIdx.y, g = h / (NH / NKV), per = NH + 2 * NKV;
    const bf16_t* qkv = qkv_all + (size_t)off * per * HD;
    bf16_t* out = out_all + (size_t)off * NH * HD;
    const bf16_t* kc = kv.k + (size_t)g * kFaK * HD;
    const bf16_t* vc = kv.v + (size_t)g * kFaK * HD;
    const float sl2 = scale * 1.4426950408889634f;
    const int skey = tid & 63, sch = tid >> 6;
    const int qrow = q0 + wave * 16 + fr;
    const int qrc = qrow < L ? qrow : L - 1;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const bf16x8_t*>(qkv + (size_t)qrc * per * HD + (size_t)h * HD + ks * 32 + fq * 8);
    const int q_hi = min(q0 + Q, L) - 1;
    const int t_lo = n_pad / kFaK, t_hi = q_hi / kFaK;                    // key tiles [t_lo, t_hi]: at most kFsTiles of them
    // ---- stage every tile (slot i = tile t_lo + i; slots past t_hi repeat tile t_hi and are never read) ----
    u32x4 kst[2][CPT], vst[2][CPT];
    auto issue = [&](u32x4 (&kr)[CPT], u32x4 (&vr)[CPT], int i) {
        const int tile = t_lo + i < t_hi ? t_lo + i : t_hi;
        const size_t o_ = (size_t)table[tile] * kv.blk_stride + (size_t)skey * HD;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            kr[j] = *reinterpret_cast<const u32x4*>(kc + o_ + (sch + NW * j) * 8);
            vr[j] = *reinterpret_cast<const u32x4*>(vc + o_ + (sch + NW * j) * 8);
        }
    };
    auto stage = [&](const u32x4 (&kr)[CPT], const u32x4 (&vr)[CPT], int i) {
        bf16_t* Ks = KsAll + (size_t)i * kFaK * kFaKLd;
        bf16_t* Vt = VtAll + (size_t)i * HD * kFaVLd;
#pragma unroll
        for (int j = 0; j < CPT; ++j) {
            *reinterpret_cast<u32x4*>(&Ks[skey * kFaKLd + (sch + NW * j) * 8]) = kr[j];
            // transposed store, two keys per 32-bit write (see flash_prefill_kernel)
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t mine = vr[j][w];
                const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);
                const bool odd = skey & 1;
                const uint32_t word = odd ? ((other >> 16) | (mine & 0xFFFF0000u)) : ((mine & 0xFFFFu) | (other << 16));
                const int dim = (sch + NW * j) * 8 + 2 * w + (odd ? 1 : 0);
                *reinterpret_cast<uint32_t*>(&Vt[dim * kFaVLd + (skey & ~1)]) = word;
            }
        }
    };
    issue(kst[0], vst[0], 0);
    issue(kst[1], vst[1], 1);
    stage(kst[0], vst[0], 0);
    issue(kst[0], vst[0], 2);
    stage(kst[1], vst[1], 1);
    issue(kst[1], vst[1], 3);
    stage(kst[0], vst[0], 2);
    stage(kst[1], vst[1], 3);
    __syncthreads();
    f32x4_t o[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) o[d] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m[4], l[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { m[r] = -1e30f; l[r] = 0.f; }
    bf16_t* ph = PsAll + (size_t)(wave * 2 + 0) * 16 * kFaPLd;
    bf16_t* pl = PsAll + (size_t)(wave * 2 + 1) * 16 * kFaPLd;
    for (int tile = t_lo; tile <= t_hi; ++tile) {
        const int i = tile - t_lo;
        flash_tile(KsAll + (size_t)i * kFaK * kFaKLd, VtAll + (size_t)i * HD * kFaVLd, ph, pl, qf, o, m, l, tile, q0, wave, fr, fq, n_pad, sl2);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + wave * 16 + fq * 4 + r;
        if (qi >= L) continue;
        const float inv = (qi >= n_pad && l[r] > 0.f) ? 1.0f / l[r] : 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d) out[((size_t)qi * NH + h) * HD + d * 16 + fr] = f_to_bf16(o[d][r] * inv);
    }
}

// q / k head-norm + RoPE + K / V write for the packed rows of several sequences in one launch (see qk_norm_rope_kv_kernel)
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_kv_pack_kernel(T* qkv, const T* qw, const T* kw, float eps, const float* cos_tab,
                                                                   const float* sin_tab, int rope_len, PagedKV<T> kvp, PackSeq sq, int NH, int NKV) {
    constexpr int HD = kHeadDim;
    const int per = NH + 2 * NKV;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int Lt = sq.off[sq.n];
    if (w >= Lt * per) return;
    const int tg = w / per, v = w - tg * per;
    int qi = 0;                                          // the sequence of packed row tg (uniform per wave: scalar compares)
    for (int q = 1; q < sq.n; ++q) qi = tg >= sq.off[q] ? q : qi;
    const int t = tg - sq.off[qi];
    if (t < sq.n_pad[qi]) return;
    PagedKV<T> kv = kvp;
    kv.table = sq.table[qi];
    T* src = qkv + (size_t)tg * per * HD + (size_t)v * HD;
    float x0 = DT<T>::ld(src + lane), x1 = DT<T>::ld(src + lane + 64);
    if (v < NH + NKV) {
        const T* gw = v < NH ? qw : kw;
        const float ss = wave_sum(fmaf(x0, x0, x1 * x1));
        const float rs = 1.0f / sqrtf(ss / (float)HD + eps);
        const float n0 = DT<T>::rnd(DT<T>::ld(gw + lane) * DT<T>::rnd(x0 * rs));
        const float n1 = DT<T>::rnd(DT<T>::ld(gw + lane + 64) * DT<T>::rnd(x1 * rs));
        int rp = t + sq.rope_delta[qi];
        rp = rp < 0 ? 0 : (rp >= rope_len ? rope_len - 1 : rp);
        const float cs = cos_tab[(size_t)rp * 64 + lane], sn = sin_tab[(size_t)rp * 64 + lane];
        x0 = DT<T>::rnd(DT<T>::rnd(n0 * cs) + DT<T>::rnd(-n1 * sn));
        x1 = DT<T>::rnd(DT<T>::rnd(n1 * cs) + DT<T>::rnd(n0 * sn));
    }
    T* dst = v < NH ? src : (v < NH + NKV ? kv.k + paged_row(kv, v - NH, t) : kv.v + paged_row(kv, v - NH - NKV, t));
    DT<T>::st(dst + lane, x0);
    DT<T>::st(dst + lane + 64, x1);
}

// (the LDS limit of a kernel is a per-DEVICE setting: remembered per device, for a process that drives more than one)
static bool flash_small_prepare() {
    static int state[kSkMaxDevices] = {};                 // 0 = not asked yet, 1 = raised, -1 = failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSkMaxDevices) return false;
    if (!state[dev])
        state[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&flash_prefill_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)kFsLdsBytes) == hipSuccess ? 1 : -1;
    return state[dev] > 0;
}
// every sequence short enough, one pool, bf16: the packed attention kernels take the whole group (false: the per-prompt launches)
static bool pack_attention_ok(fq3_ctx* const* cs, int n, const int* L) {
    if (n < 1 || n > kMaxPack || cs[0]->cfg.dtype != FQ3_BF16 || !cs[0]->opt_flash_prefill || !cs[0]->opt_flash_small) return false;
    for (int q = 0; q < n; ++q)
        if (L[q] > kFsMaxRows || cs[q]->tk.pool != cs[0]->tk.pool || cs[q]->opt_flash_small == 0) return false;
    return flash_small_prepare();
}
static PackSeq pack_seq(fq3_ctx* const* cs, int n, const int* L, const int* n_pad) {
    PackSeq sq{};
    sq.n = n;
    for (int q = 0; q < n; ++q) {
        sq.table[q] = cs[q]->tk.d_table; sq.off[q + 1] = sq.off[q] + L[q]; sq.n_pad[q] = n_pad[q]; sq.rope_delta[q] = cs[q]->rope_delta;
    }
    return sq;
}
// one layer's q / k norm + RoPE + KV write and causal attention for every sequence of the pack: two launches
static bool pack_attention_layer(fq3_ctx* c0, int layer, const fq3_layer_weights& w, const PackSeq& sq, int Lmax, bf16_t* QKV, bf16_t* ATT,
                                 float scale, hipStream_t s) {
    const fq3_stack_dims& d = c0->cfg.talker;
    const int NH = d.n_heads, NKV = d.n_kv_heads, Lt = sq.off[sq.n];
    const PagedKV<bf16_t> kv = paged_kv<bf16_t>(c0, layer);
    hipLaunchKernelGGL((qk_norm_rope_kv_pack_kernel<bf16_t>), dim3((Lt * (NH + 2 * NKV) + 3) / 4), dim3(256), 0, s, QKV, (const bf16_t*)w.q_norm,
                       (const bf16_t*)w.k_norm, d.rms_eps, c0->wt.talker_cos, c0->wt.talker_sin, c0->wt.talker_rope_len, kv, sq, NH, NKV);
    hipLaunchKernelGGL(flash_prefill_small_kernel, dim3((Lmax + kFaQ - 1) / kFaQ, NH, sq.n), dim3(256), kFsLdsBytes, s, (const bf16_t*)QKV, kv, ATT, sq,
                       NH, NKV, scale);
    return hipGetLastError() == hipSuccess;                // a refused launch (e.g. the 158 KB of LDS) must not leave ATT stale silently
}

// shape choice: short prompts keep 64-query blocks, one per workgroup (parallelism first); from 1024 tokens the blocks are paired
// (equal work per workgroup under the causal mask), from 3072 tokens they hold 128 queries
static void flash_prefill_launch(const bf16_t* qkv, const PagedKV<bf16_t>& kv, bf16_t* out, int L, int n_pad, int NH,
                                 int NKV, float scale, hipStream_t s) {
    if (L >= 3072) {
        const int nqb = (L + 127) / 128;
        hipLaunchKernelGGL((flash_prefill_kernel<8, true>), dim3((nqb + 1) / 2, NH), dim3(512), 0, s, qkv, kv, out, L, n_pad, NH, NKV, scale, nqb);
    } else if (L >= 1024) {
        const int nqb = (L + 63) / 64;
        hipLaunchKernelGGL((flash_prefill_kernel<4, true>), dim3((nqb + 1) / 2, NH), dim3(256), 0, s, qkv, kv, out, L, n_pad, NH, NKV, scale, nqb);
    } else {
        const int nqb = (L + 63) / 64;
        hipLaunchKernelGGL((flash_prefill_kernel<4, false>), dim3(nqb, NH), dim3(256), 0, s, qkv, kv, out, L, n_pad, NH, NKV, scale, nqb);
    }
}

// RMSNorm of prompt rows, one wave per row, the whole row in registers: 16-byte loads, one reduction, 16-byte stores (the row-loop
// kernel the codec uses takes 10 us on 200 rows of 1024: 32 dependent 2-byte accesses per lane).  Per element the arithmetic is
// the same -- w * rnd(x * rs), rounded on store; only the order of the sum of squares differs.  C = NCH * 512.
template <typename T, int NCH>
__global__ __launch_bounds__(256) void rmsnorm_rows_vec_kernel(const T* x, const T* w, T* y, int rows, float eps) {
    constexpr int C = NCH * 512;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    Raw8<T> xr[NCH], wr[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        ldraw<false>(xr[j], x + (size_t)row * C + j * 512 + lane * 8);
        ldraw<false>(wr[j], w + j * 512 + lane * 8);
    }
    float xv[NCH][8], ss = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        unpack(xr[j], xv[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) ss = fmaf(xv[j][i], xv[j][i], ss);
    }
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)C + eps);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        float wv[8];
        unpack(wr[j], wv);
#pragma unroll
        for (int i = 0; i < 8; ++i) xv[j][i] = wv[i] * DT<T>::rnd(xv[j][i] * rs);
        DT<T>::st8(y + (size_t)row * C + j * 512 + lane * 8, xv[j]);
    }
}
template <typename T>
void rmsnorm_rows(const T* x, const T* w, T* y, int rows, int C, float eps, hipStream_t s) {
    const dim3 grid((rows + 3) / 4), block(256);
    if (C == 1024) hipLaunchKernelGGL((rmsnorm_rows_vec_kernel<T, 2>), grid, block, 0, s, x, w, y, rows, eps);
    else if (C == 2048) hipLaunchKernelGGL((rmsnorm_rows_vec_kernel<T, 4>), grid, block, 0, s, x, w, y, rows, eps);
    else hipLaunchKernelGGL((rmsnorm_rows_kernel<T>), grid, block, 0, s, x, w, y, 0, rows, C, eps);
}

constexpr long kPrefillWsFloats = 8L << 20;       // split-K partials of the short-prompt GEMMs (32 MB)
// every prefill GEMM lends the split-K workspace: that also marks it free to take the weight-stationary kernel (skinny_gemm.cuh)
// (kind: 0 = a plain matrix, 1 = [gate | up] halves -- which fragment-major copy the weight-stationary kernel would read, fq3_ctx.h)
template <typename T>
GemmArgs lin(fq3_ctx* c, const void* A, int M, int K, const void* W, int N, void* Y, int kind = 0) {
    GemmArgs a{}; a.A = A; a.lda = K; a.M = M; a.a_rows = M; a.n_taps = 1; a.tap_off[0] = 0; a.Cin = K; a.W = W; a.N = N;
    a.bias_mod = N; a.Y = Y; a.ldy = N; a.ws = (float*)c->pf_ws; a.ws_floats = kPrefillWsFloats; a.no_skinny = c->opt_no_skinny;
    if (sizeof(T) == 2 && c->opt_packed && M <= kSkinnyMaxRows) a.Wp = fq3_packed_find_(W, kind);
    if (sizeof(T) == 2 && kind == 1 && c->opt_swiglu_tile && M > kSkinnyMaxRows) a.Wi = fq3_packed_find_(W, 2);
    return a;
}
template <typename T> void gemm(const GemmArgs& a, hipStream_t s) { gemm_launch<T>(a, s); }

template <typename T>
int prefill_t(fq3_ctx* c, const void* embeds, int L, int n_pad, void* out_logits, void* out_hidden, hipStream_t s) {
    const fq3_stack_dims& d = c->cfg.talker;
    const int H = d.hidden, I = d.inter, NH = d.n_heads, NKV = d.n_kv_heads;
    const int QD = NH * kHeadDim, KVD = NKV * kHeadDim, per = QD + 2 * KVD;
    if (H % 32 || I % 32) return fq3_fail_(FQ3_EUNSUPPORTED, "MFMA prefill needs hidden and intermediate sizes that are multiples of 32");
    if (int r = fq3_prefill_reserve_(c)) return r;
    T *X = (T*)c->pf_x, *XN = (T*)c->pf_xn, *QKV = (T*)c->pf_qkv, *ATT = (T*)c->pf_att, *GU = (T*)c->pf_gu, *ACT = (T*)c->pf_act;
    if (hipMemcpyAsync(X, embeds, (size_t)L * H * c->esz, hipMemcpyDeviceToDevice, s) != hipSuccess)
        return fq3_fail_(FQ3_EHIP, "prefill: copy of the prompt embeddings failed");
    const float scale = 1.0f / sqrtf((float)kHeadDim);
    fq3_ctx* one[1] = {c};
    const bool small = pack_attention_ok(one, 1, &L);           // <= 256 rows: every key tile resident (flash_prefill_small_kernel)
    const PackSeq sq1 = small ? pack_seq(one, 1, &L, &n_pad) : PackSeq{};
    for (int i = 0; i < d.n_layers; ++i) {
        const fq3_layer_weights& w = c->tl[i];
        rmsnorm_rows<T>((const T*)X, (const T*)w.input_norm, XN, L, H, d.rms_eps, s);
        gemm<T>(lin<T>(c, XN, L, H, w.qkv, per, QKV), s);
        if constexpr (sizeof(T) == 2) {
            if (small) {
                if (!pack_attention_layer(c, i, w, sq1, L, (bf16_t*)QKV, (bf16_t*)ATT, scale, s)) return fq3_fail_(FQ3_EHIP, "prefill: the packed attention launch failed");
                { GemmArgs a = lin<T>(c, ATT, L, QD, w.o, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
                rmsnorm_rows<T>((const T*)X, (const T*)w.post_norm, XN, L, H, d.rms_eps, s);
                gemm_swiglu_halves<T>(lin<T>(c, XN, L, H, w.gate_up, 2 * I, GU, 1), ACT, s);
                { GemmArgs a = lin<T>(c, ACT, L, I, w.down, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
                continue;
            }
        }
        const PagedKV<T> kv = paged_kv<T>(c, i);
        hipLaunchKernelGGL((qk_norm_rope_kv_kernel<T>), dim3((L * (NH + 2 * NKV) + 3) / 4), dim3(256), 0, s, QKV, (const T*)w.q_norm,
                           (const T*)w.k_norm, d.rms_eps, c->wt.talker_cos, c->wt.talker_sin, c->wt.talker_rope_len, c->rope_delta,
                           kv, L, n_pad, NH, NKV);
        if constexpr (sizeof(T) == 2) {
            if (c->opt_flash_prefill) flash_prefill_launch((const bf16_t*)QKV, kv, (bf16_t*)ATT, L, n_pad, NH, NKV, scale, s);
            else hipLaunchKernelGGL((prefill_attn_kernel<T>), dim3((L * NH + 3) / 4), dim3(256), 0, s, (const T*)QKV, kv, ATT, L, n_pad, NH, NKV, scale);
        } else {
            hipLaunchKernelGGL((prefill_attn_kernel<T>), dim3((L * NH + 3) / 4), dim3(256), 0, s, (const T*)QKV, kv, ATT, L, n_pad, NH, NKV, scale);
        }
        { GemmArgs a = lin<T>(c, ATT, L, QD, w.o, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
        rmsnorm_rows<T>((const T*)X, (const T*)w.post_norm, XN, L, H, d.rms_eps, s);
        gemm_swiglu_halves<T>(lin<T>(c, XN, L, H, w.gate_up, 2 * I, GU, 1), ACT, s);
        { GemmArgs a = lin<T>(c, ACT, L, I, w.down, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
    }
    // final norm of the last row only -> past_hidden; logits through the decode-path head GEMV
    hipLaunchKernelGGL((rmsnorm_kernel<T>), dim3(1), dim3(256), 0, s, (const T*)X + (size_t)(L - 1) * H, (const T*)c->wt.talker_final_norm,
                       (T*)out_hidden, H, d.rms_eps);
    if (out_logits) return fq3_codec_head_launch_(c, out_hidden, out_logits, s);
    return 0;
}

// The same prefill for n prompts at once: everything row-wise (norms, the four GEMMs, SwiGLU) runs over the PACKED rows of all
// prompts -- one pass over the layer's weights instead of n -- and only the sequence-specific steps run per prompt on its
// slice of the packed rows: q/k norm + RoPE + KV write into that context's own cache, and causal attention over it.
// Workspaces are those of ctxs[0] (sum of the lengths <= its max_seq_len).  Per element the arithmetic is that of
// prefill_t; what can differ is the GEMM tile / split-K choice, which depends on the row count (bf16: last-bit level).
template <typename T>
int prefill_batch_t(fq3_ctx* const* cs, int n, const void* const* embeds, const int* L, const int* n_pad, void* const* out_logits,
                    void* const* out_hidden, hipStream_t s) {
    fq3_ctx* c = cs[0];
    const fq3_stack_dims& d = c->cfg.talker;
    const int H = d.hidden, I = d.inter, NH = d.n_heads, NKV = d.n_kv_heads;
    const int QD = NH * kHeadDim, KVD = NKV * kHeadDim, per = QD + 2 * KVD;
    if (H % 32 || I % 32) return fq3_fail_(FQ3_EUNSUPPORTED, "MFMA prefill needs hidden and intermediate sizes that are multiples of 32");
    if (int r = fq3_prefill_reserve_(c)) return r;
    T *X = (T*)c->pf_x, *XN = (T*)c->pf_xn, *QKV = (T*)c->pf_qkv, *ATT = (T*)c->pf_att, *GU = (T*)c->pf_gu, *ACT = (T*)c->pf_act;
    std::vector<int> off(n + 1, 0);
    for (int q = 0; q < n; ++q) off[q + 1] = off[q] + L[q];
    const int Lt = off[n];
    for (int q = 0; q < n; ++q)
        if (hipMemcpyAsync(X + (size_t)off[q] * H, embeds[q], (size_t)L[q] * H * c->esz, hipMemcpyDeviceToDevice, s) != hipSuccess)
            return fq3_fail_(FQ3_EHIP, "prefill: copy of the prompt embeddings failed");
    const float scale = 1.0f / sqrtf((float)kHeadDim);
    const bool small = pack_attention_ok(cs, n, L);             // every prompt <= 256 rows, one pool: the pack's attention in two launches per layer
    const PackSeq sqn = small ? pack_seq(cs, n, L, n_pad) : PackSeq{};
    int Lmax = 0;
    for (int q = 0; q < n; ++q) Lmax = std::max(Lmax, L[q]);
    for (int i = 0; i < d.n_layers; ++i) {
        const fq3_layer_weights& w = c->tl[i];
        rmsnorm_rows<T>((const T*)X, (const T*)w.input_norm, XN, Lt, H, d.rms_eps, s);
        gemm<T>(lin<T>(c, XN, Lt, H, w.qkv, per, QKV), s);
        if constexpr (sizeof(T) == 2) {
            if (small && !pack_attention_layer(c, i, w, sqn, Lmax, (bf16_t*)QKV, (bf16_t*)ATT, scale, s))
                return fq3_fail_(FQ3_EHIP, "prefill: the packed attention launch failed");
        }
        for (int q = 0; q < n && !small; ++q) {
            fq3_ctx* cq = cs[q];
            T* qkv = QKV + (size_t)off[q] * per;
            T* att = ATT + (size_t)off[q] * QD;
            const int Lq = L[q], pq = n_pad[q];
            const PagedKV<T> kv = paged_kv<T>(cq, i);
            hipLaunchKernelGGL((qk_norm_rope_kv_kernel<T>), dim3((Lq * (NH + 2 * NKV) + 3) / 4), dim3(256), 0, s, qkv, (const T*)w.q_norm,
                               (const T*)w.k_norm, d.rms_eps, c->wt.talker_cos, c->wt.talker_sin, c->wt.talker_rope_len, cq->rope_delta,
                               kv, Lq, pq, NH, NKV);
            bool flash = false;
            if constexpr (sizeof(T) == 2) flash = c->opt_flash_prefill != 0;
            if constexpr (sizeof(T) == 2) {
                if (flash) flash_prefill_launch((const bf16_t*)qkv, kv, (bf16_t*)att, Lq, pq, NH, NKV, scale, s);
            }
            if (!flash)
                hipLaunchKernelGGL((prefill_attn_kernel<T>), dim3((Lq * NH + 3) / 4), dim3(256), 0, s, (const T*)qkv, kv, att, Lq, pq, NH, NKV, scale);
        }
        { GemmArgs a = lin<T>(c, ATT, Lt, QD, w.o, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
        rmsnorm_rows<T>((const T*)X, (const T*)w.post_norm, XN, Lt, H, d.rms_eps, s);
        gemm_swiglu_halves<T>(lin<T>(c, XN, Lt, H, w.gate_up, 2 * I, GU, 1), ACT, s);
        { GemmArgs a = lin<T>(c, ACT, Lt, I, w.down, H, X); a.res = X; a.ldr = H; gemm<T>(a, s); }
    }
    for (int q = 0; q < n; ++q) {
        hipLaunchKernelGGL((rmsnorm_kernel<T>), dim3(1), dim3(256), 0, s, (const T*)X + (size_t)(off[q + 1] - 1) * H, (const T*)c->wt.talker_final_norm,
                           (T*)out_hidden[q], H, d.rms_eps);
        if (out_logits && out_logits[q])
            if (int r = fq3_codec_head_launch_(cs[q], out_hidden[q], out_logits[q], s)) return r;
    }
    return 0;
}

}  // namespace

// The matrix-core prefill's activation workspace (max_seq_len rows of x, norm(x), qkv, attention, gate|up, act + the split-K
// partials): allocated on first use, or up front through fq3_prefill_reserve -- a scheduler that prefills into spare contexts
// while other lanes decode must not meet a hipMalloc there.
int fq3_prefill_reserve_(fq3_ctx* c) {
    if (c->pf_x) return 0;
    const fq3_stack_dims& d = c->cfg.talker;
    const size_t H = d.hidden, I = d.inter, QD = (size_t)d.n_heads * kHeadDim, per = QD + 2 * (size_t)d.n_kv_heads * kHeadDim;
    const size_t rows = (size_t)c->cfg.max_seq_len;
    int r;
    void* x = nullptr;
    if ((r = fq3_dmalloc_(c, &x, rows * H * c->esz))) return r;
    if ((r = fq3_dmalloc_(c, &c->pf_xn, rows * H * c->esz))) return r;
    if ((r = fq3_dmalloc_(c, &c->pf_qkv, rows * per * c->esz))) return r;
    if ((r = fq3_dmalloc_(c, &c->pf_att, rows * QD * c->esz))) return r;
    if ((r = fq3_dmalloc_(c, &c->pf_gu, rows * 2 * I * c->esz))) return r;
    if ((r = fq3_dmalloc_(c, &c->pf_act, rows * I * c->esz))) return r;
    if ((r = fq3_dmalloc_(c, &c->pf_ws, (size_t)kPrefillWsFloats * sizeof(float)))) return r;
    c->pf_x = x;                                   // set last: marks the whole set as present
    return 0;
}

int fq3_prefill_batch_mfma_(fq3_ctx* const* cs, int n, const void* const* embeds, const int* L, const int* n_pad, void* const* out_logits,
                            void* const* out_hidden, hipStream_t s) {
    return cs[0]->cfg.dtype == FQ3_BF16 ? prefill_batch_t<bf16_t>(cs, n, embeds, L, n_pad, out_logits, out_hidden, s)
                                        : prefill_batch_t<float>(cs, n, embeds, L, n_pad, out_logits, out_hidden, s);
}

int fq3_prefill_mfma_(fq3_ctx* c, const void* embeds, int L, int n_pad, void* out_logits, void* out_hidden, hipStream_t s) {
    return c->cfg.dtype == FQ3_BF16 ? prefill_t<bf16_t>(c, embeds, L, n_pad, out_logits, out_hidden, s)
                                    : prefill_t<float>(c, embeds, L, n_pad, out_logits, out_hidden, s);
}
