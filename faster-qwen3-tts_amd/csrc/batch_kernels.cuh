// Batched decode (the reference fixes batch = 1, talker_graph.py:46 /
// predictor_graph.py:70): B utterances ("lanes") advance in lock-step through ONE launch chain and ONE pass over
// the weights per frame.  Every lane keeps its own single-stream context (KV caches, DecodeState, history, codes,
// noise rings -- prefill and fq3_decode_begin are the single-stream entry points); only the per-frame chain is
// replaced.  The lane-local kernels below are thin wrappers that point the single-stream bodies at lane `l`;
// gemv_batch_kernel is the M = B GEMV on the VALU: tokens prepared (RMSNorm, T-rounded) cooperatively into LDS in
// groups of up to 8, weight rows in registers, every wave walks the tokens.  Same arithmetic, order and rounding as
// gemv_kernel, so a lane's ids are bit-identical to the same utterance decoded alone (tests/test_gpu_batch.py).
// The bf16 default is the pair of matrix-core kernels further down (one 16-column token tile per MFMA: up to 16 lanes
// cost the same weight stream and the same MFMA count as 8).
#pragma once
#include "decode_kernels.cuh"
#include "sampler.cuh"
#include "sampler_wave.cuh"

namespace fq3 {

constexpr int kMaxLanes = 128;       // eight token tiles (round 4: 64, then 128; 32 = two tiles in round 3)
constexpr int kTokTile = 16;         // token columns of one v_mfma_f32_16x16x32_bf16 tile: lanes 0..15 / 16..31 of a batch are passes of the
                                     // same launch over register-resident weight fragments
constexpr int kGroupLanes = 8;       // VALU batch GEMV: tokens staged in LDS per pass over the register-resident weight rows

// The per-lane pointer tables live in DEVICE memory, one set per batch (written once in fq3_batch_create), and the kernels take
// their addresses: 128 lanes x (state, codes, history, past_hidden) alone are the whole 4 KB a launch may carry as arguments.  A
// lane-local kernel reads its own entries with scalar loads (uniform per workgroup).
struct LaneTab {
    DecodeState* st[kMaxLanes];
    int* codes[kMaxLanes];
    unsigned char* seen[kMaxLanes];
    void* past_hidden[kMaxLanes];
};
struct LaneSt { DecodeState* st[kMaxLanes]; };                    // the loop states alone (batch poll)
struct LaneForced { const TeacherForcing* tf[kMaxLanes]; };       // teacher-forcing objects of the lanes (parity tests; null in product use)
struct LaneKV { void* k[kMaxLanes]; void* v[kMaxLanes]; };       // per lane: the predictor's contiguous cache, or the base of the talker's block pool
struct LaneTabs { const int* t[kMaxLanes]; int blk_stride; };     // talker: every lane's block table (paged KV, decode_kernels.cuh)

template <typename T>
__global__ __launch_bounds__(256) void frame_begin_batch_kernel(const LaneTab* __restrict__ t, const T* codec_emb, T* pred_in, int H, int G) {
    const int l = blockIdx.x;
    frame_begin_body<T>(gptr(t->st[l]), codec_emb, gptr(reinterpret_cast<const T*>(t->past_hidden[l])), pred_in + (size_t)l * 2 * H,
                        gptr(t->codes[l]), gptr(t->seen[l]), H, G);
}

template <typename T, int G>
__global__ __launch_bounds__(256) void embed_sum_batch_kernel(const LaneTab* __restrict__ t, EmbTables tabs, T* x, int H, const float* cos_tab,
                                                              const float* sin_tab, int rope_len, float* rope_now) {
    const int l = blockIdx.x;
    DecodeState* st = gptr(t->st[l]);
    embed_sum_body<T, G>(st, tabs, gptr(t->codes[l]), x + (size_t)l * H, H, cos_tab, sin_tab, rope_len, st->rope_delta,
                         rope_now + (size_t)l * kHeadDim);
}

// code-predictor attention: grid (n_heads, B)
template <typename T>
__global__ __launch_bounds__(64) void attn_pred_batch_kernel(AttnArgs a, const LaneKV* __restrict__ kv, int qkv_stride, int out_stride) {
    const int l = blockIdx.y;
    a.qkv = reinterpret_cast<const T*>(a.qkv) + (size_t)l * qkv_stride;
    a.out = reinterpret_cast<T*>(a.out) + (size_t)l * out_stride;
    a.kcache = gptr(kv->k[l]); a.vcache = gptr(kv->v[l]);
    attn_pred_body<T>(a);
}
// the default since round 6: grid (n_kv_heads, B), one wave per (kv group, lane) -- both q heads of a group from one read of the live
// K / V rows (decode_kernels.cuh::attn_pred_group_body; bit-identical to the per-head form above, which stays as "pred_attn_group" 0)
template <typename T, int REP>
__global__ __launch_bounds__(64) void attn_pred_group_batch_kernel(AttnArgs a, const LaneKV* __restrict__ kv, int qkv_stride, int out_stride) {
    const int l = blockIdx.y;
    a.qkv = reinterpret_cast<const T*>(a.qkv) + (size_t)l * qkv_stride;
    a.out = reinterpret_cast<T*>(a.out) + (size_t)l * out_stride;
    a.kcache = gptr(kv->k[l]); a.vcache = gptr(kv->v[l]);
    attn_pred_group_body<T, REP>(a, (int)blockIdx.x);
}

// talker attention: grid (n_kv, workers, B); position, pad count and RoPE row are the lane's own
template <typename T, int REP>
__global__ __launch_bounds__(256) void attn_decode_batch_kernel(AttnArgs a, const LaneKV* __restrict__ kv, const LaneTabs* __restrict__ tabs,
                                                                const LaneTab* __restrict__ t, int qkv_stride,
                                                                const float* rope_now, size_t part_stride) {
    const int l = blockIdx.z;
    a.part = a.part + (size_t)l * part_stride;
    // A worker whose first key tile lies beyond the lane's position has nothing to read and is not the one that appends the new
    // K / V row.  The single-stream kernel lets it load a (clamped) tile anyway, so that the position need not be known before the
    // loads go out; across 32 lanes those idle workers were half of the launch's HBM traffic (workers are sized for max_seq_len,
    // the benchmarked caches hold 4 of 8 tiles: 33 of 66 MB per launch, profiles/r03_pmc_batch32_fetch.txt).  Here it leaves its
    // empty partial slot -- the very values the walk over no valid key produces: {0, m = -1e30, l = 0} -- and exits.
    // A lane that is DONE (EOS / limits reached, cancelled, or never armed) must not touch the cache at all: its block table may be
    // stale -- the scheduler returns a finished lane's blocks to the pool, and they may already belong to another context -- so the
    // append of the single-stream body would land in somebody else's rows.  It leaves neutral partials ({0, m = 0, l = 1}: the merge
    // yields zeros, not 0 / 0) and exits.
    const DecodeState* stl = gptr(t->st[l]);
    const int pos = stl->pos, lane_done = stl->done;
    const bool no_keys = (int)blockIdx.y * kKeysPerTile > pos;
    if (lane_done || no_keys) {
        for (int e = threadIdx.x; e < REP * kHeadDim; e += 256) {
            const int h = e / kHeadDim, d = e - h * kHeadDim;
            float* p = a.part + (((size_t)blockIdx.x * kMaxWorkers + blockIdx.y) * REP + h) * kPartStride;
            p[d] = 0.f;
            if (d == 0) { p[kHeadDim] = lane_done ? 0.f : -1e30f; p[kHeadDim + 1] = lane_done ? 1.f : 0.f; }
        }
        return;
    }
    a.qkv = reinterpret_cast<const T*>(a.qkv) + (size_t)l * qkv_stride;
    a.kcache = gptr(kv->k[l]); a.vcache = gptr(kv->v[l]);
    a.table = gptr(tabs->t[l]); a.blk_stride = tabs->blk_stride;
    a.pos_ptr = nullptr; a.pos_imm = pos;
    a.n_pad = stl->n_pad;
    a.cos_row = rope_now + (size_t)l * kHeadDim; a.sin_row = a.cos_row + 64;
    attn_decode_body<T, REP, true>(a);
}

// Talker attention of the lock-step batch from 64 lanes on (round 5): ONE workgroup per (kv head, lane) instead of one per
// (kv head, 64-key tile, lane) + a merge launch.  At 128 lanes the split kernel is 8 x 8 x 128 = 8192 workgroups of which the ~4100
// that have keys each repeat the lane's q / k normalisation + RoPE, chase lane state -> block table -> tile before their first K / V
// byte is requested, and live ~6 us: with 4-5 of them resident per CU the launch is three to four rounds of that latency chain
// (40.7 us per layer at 128 lanes x ~230 keys, profiles/r04_batch128_kernel_trace.txt, 19 % of the frame), and its partial slots need
// combine_batch_kernel afterwards.  Here every (kv head, lane) pair is resident at once (8 x 128 = 1024 workgroups of 4 waves), the
// prologue runs once per pair, wave w walks key tiles w, w + 4, ... in quarter tiles of 16 keys through two alternating register
// sets (the next quarter's 8 loads are in flight while this one is multiplied; all loads unconditional on clamped steps), the four
// waves meet once in LDS, and the workgroup writes the FINAL head outputs (fp32 softmax, one rounding to T -- the value
// combine_batch_kernel produces), so the merge launch disappears.  Only live keys are read; the new token's K / V row is appended by
// this workgroup (it is the only one that sees the pair); a lane that is done leaves zeros and touches no cache block (its block
// table may be stale, see attn_decode_batch_kernel).  The order in which keys enter the online softmax differs from the split
// kernel, so this form has a summation order of its own: it starts at 64 lanes (fq3_batch "attn_lane"), where the parity floors are
// the oracle-derived ones (tests/test_gpu_batch_fulldepth.py).
template <typename T, int REP, int NI = 4>
__global__ __launch_bounds__(256) void attn_decode_lane_kernel(AttnArgs a, const LaneKV* __restrict__ kv, const LaneTabs* __restrict__ tabs,
                                                               const LaneTab* __restrict__ t, int qkv_stride, const float* rope_now, int out_stride) {
    constexpr int HD = kHeadDim, KS = kKeysPerTile;
    constexpr int SPT = KS / (4 * NI);                 // steps per 64-key tile: a step = NI keys for each of the wave's four 16-lane groups
    static_assert(NI == 4 || NI == 2, "16- or 8-key steps");
    const int g = blockIdx.x, l = blockIdx.y;
    __shared__ float qs[REP][HD];
    __shared__ float knew[HD], vnew[HD];
    __shared__ float wo[4][REP][HD];
    __shared__ float wm[4][REP], wl[4][REP];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sub = lane >> 4, c = lane & 15;
    const DecodeState* stl = gptr(t->st[l]);
    const int pos = stl->pos, lane_done = stl->done, n_pad = stl->n_pad;
    T* out = reinterpret_cast<T*>(a.out) + (size_t)l * out_stride + (size_t)g * REP * HD;
    if (lane_done) {
        for (int e = tid; e < REP * HD; e += 256) DT<T>::st(out + e, 0.f);
        return;
    }
    const T* qkv = reinterpret_cast<const T*>(a.qkv) + (size_t)l * qkv_stride;
    const int q_dim = a.n_kv * REP * HD, kv_dim = a.n_kv * HD;
    T* kc = gptr(reinterpret_cast<T*>(kv->k[l])) + (size_t)g * KS * HD;        // this kv head's 64 rows inside block 0
    T* vc = gptr(reinterpret_cast<T*>(kv->v[l])) + (size_t)g * KS * HD;
    const int* table = gptr(tabs->t[l]);
    const int blk_stride = tabs->blk_stride;
    const int t_pos = pos / KS;                                                // the tile the new key falls into (keys < pos are read)
    // this wave's steps: part q of tile wave + 4 * (j / SPT) is step j; a wave without a tile walks (and discards) tile t_pos
    const int ntw = wave <= t_pos ? (t_pos - wave) / 4 + 1 : 0;
    // the tile the new key falls into holds only pos % 64 live keys: its steps beyond them would add exact no-ops (p = 0, rescale by
    // exp(0)), so the wave that owns it stops there (an even count: the two register sets alternate) -- up to a tile less of K / V per pair
    const int last_steps = ((pos - t_pos * KS + 4 * NI - 1) / (4 * NI) + 1) & ~1;
    const int nsteps = (ntw > 0 && (t_pos - wave) % 4 == 0) ? (ntw - 1) * SPT + (last_steps < SPT ? last_steps : SPT) : ntw * SPT;
    auto step_tile = [&](int j) { j = j < nsteps ? j : nsteps - 1; return ntw ? wave + 4 * (j / SPT) : t_pos; };
    Raw8<T> ka[NI], va[NI], kb[NI], vb[NI];
    auto issue = [&](Raw8<T> (&kr)[NI], Raw8<T> (&vr)[NI], int j, int blk) {
        const int jq = (j < nsteps ? j : nsteps - 1) & (SPT - 1);              // (nsteps = 0: the last part of tile t_pos, discarded)
        const size_t base = (size_t)blk * blk_stride + (size_t)(jq * 4 * NI + sub) * HD + c * 8;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            ldraw<false>(kr[i], kc + base + (size_t)i * 4 * HD);
            ldraw<false>(vr[i], vc + base + (size_t)i * 4 * HD);
        }
    };
    int blk_cur = table[step_tile(0)];
    // ---- token-side loads first (they fly while the table entry arrives), then the first two quarters ----
    constexpr int NV = (REP + 2 + 3) / 4;
    float px0[NV], px1[NV], pw0[NV], pw1[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int vec = wave + 4 * i < REP + 2 ? wave + 4 * i : REP + 1;
        const T* src = vec < REP ? qkv + (size_t)(g * REP + vec) * HD
                     : (vec == REP ? qkv + q_dim + (size_t)g * HD : qkv + q_dim + kv_dim + (size_t)g * HD);
        px0[i] = DT<T>::ld(src + lane); px1[i] = DT<T>::ld(src + lane + 64);
        const T* w = reinterpret_cast<const T*>(vec < REP ? a.q_norm_w : a.k_norm_w);
        pw0[i] = DT<T>::ld(w + lane); pw1[i] = DT<T>::ld(w + lane + 64);
    }
    const float* cos_row = rope_now + (size_t)l * kHeadDim;
    const float cs = cos_row[lane], sn = cos_row[64 + lane];
    issue(ka, va, 0, blk_cur);
    issue(kb, vb, 1, blk_cur);
    int blk_next = table[step_tile(SPT)];                                        // the next tile's entry, one tile ahead
    __builtin_amdgcn_sched_barrier(0);
    // ---- prologue: head RMSNorm + RoPE of the q heads and the new k; the new row goes to slot pos % 64 of the block of tile t_pos ----
    const size_t new_off = (size_t)table[t_pos] * blk_stride + (size_t)(pos - t_pos * KS) * HD;
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
            T* cp = (vec == REP ? kc : vc) + new_off;
            DT<T>::st(cp + lane, x0); DT<T>::st(cp + lane + 64, x1);
        }
    }
    __syncthreads();
    float qr[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i) qr[h][i] = qs[h][c * 8 + i];
    float m[REP], lsum[REP], o[REP][8];
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        m[h] = -1e30f; lsum[h] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) o[h][i] = 0.f;
    }
    // one step (4 NI keys: this lane's are key0 + 4 i + sub): scores first, then ONE rescale per head
    auto quarter = [&](const Raw8<T> (&kr)[NI], Raw8<T> (&vr)[NI], int key0) {
        float sc[REP][NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int key = key0 + i * 4 + sub;
            const bool ok = key >= n_pad && key < pos;
            float kf[8];
            unpack(kr[i], kf);
            if (!ok) zero(vr[i]);                                              // a dead slot may hold NaN bits: its V row becomes zeros (p is 0 there)
#pragma unroll
            for (int h = 0; h < REP; ++h) {
                float s = 0.f;
#pragma unroll
                for (int d = 0; d < 8; ++d) s = fmaf(qr[h][d], kf[d], s);
                s = row16_sum(s);
                sc[h][i] = ok ? s * a.scale : -INFINITY;
            }
        }
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float mn = fmaxf(m[h], fmaxf(sc[h][0], sc[h][1]));
            if constexpr (NI == 4) mn = fmaxf(mn, fmaxf(sc[h][2], sc[h][3]));
            const float al = __expf(m[h] - mn);
#pragma unroll
            for (int i = 0; i < NI; ++i) sc[h][i] = __expf(sc[h][i] - mn);      // the probabilities (0 for a masked key)
            float ps = sc[h][0] + sc[h][1];
            if constexpr (NI == 4) ps += sc[h][2] + sc[h][3];
            lsum[h] = fmaf(lsum[h], al, ps);
#pragma unroll
            for (int d = 0; d < 8; ++d) o[h][d] *= al;
            m[h] = mn;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            float vf[8];
            unpack(vr[i], vf);
#pragma unroll
            for (int h = 0; h < REP; ++h)
#pragma unroll
                for (int d = 0; d < 8; ++d) o[h][d] = fmaf(sc[h][i], vf[d], o[h][d]);
        }
    };
    for (int j = 0; j < nsteps; j += 2) {                                      // (uniform per wave; nsteps is a multiple of SPT, SPT is even)
        const int tile = wave + 4 * (j / SPT);
        quarter(ka, va, tile * KS + (j & (SPT - 1)) * 4 * NI);
        // step j + 2: the same tile's next part, or (j + 2 a multiple of SPT) the next tile's first
        const bool roll = ((j + 2) & (SPT - 1)) == 0;
        const int blk2 = roll ? blk_next : blk_cur;
        issue(ka, va, j + 2, j + 2 < nsteps ? blk2 : blk_cur);
        quarter(kb, vb, tile * KS + ((j + 1) & (SPT - 1)) * 4 * NI);
        issue(kb, vb, j + 3, j + 3 < nsteps ? blk2 : blk_cur);
        if (roll) { blk_cur = j + 2 < nsteps ? blk_next : blk_cur; blk_next = table[step_tile(j + 2 + SPT)]; }
    }
    {   // the new token's own key / value (LDS), by lane group 0 of wave 0
        float kf[8], vf[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { kf[i] = knew[c * 8 + i]; vf[i] = vnew[c * 8 + i]; }
        const bool valid = wave == 0 && sub == 0 && pos >= n_pad;
#pragma unroll
        for (int h = 0; h < REP; ++h) {
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d) s = fmaf(qr[h][d], kf[d], s);
            s = row16_sum(s);
            s = valid ? s * a.scale : -INFINITY;
            const float mn = fmaxf(m[h], s);
            const float al = __expf(m[h] - mn), p = __expf(s - mn);
            lsum[h] = fmaf(lsum[h], al, p);
#pragma unroll
            for (int d = 0; d < 8; ++d) o[h][d] = fmaf(o[h][d], al, valid ? p * vf[d] : 0.f);
            m[h] = mn;
        }
    }
    // merge the four 16-lane key groups of the wave, then the four waves (LDS), and write the final head outputs
#pragma unroll
    for (int h = 0; h < REP; ++h) {
        const float M = xrow_max(m[h]);
        const float w = __expf(m[h] - M);
        lsum[h] = xrow_sum(lsum[h] * w);
#pragma unroll
        for (int i = 0; i < 8; ++i) o[h][i] = xrow_sum(o[h][i] * w);
        m[h] = M;
    }
    if (sub == 0) {
#pragma unroll
        for (int h = 0; h < REP; ++h) {
#pragma unroll
            for (int i = 0; i < 8; ++i) wo[wave][h][c * 8 + i] = o[h][i];
            if (c == 0) { wm[wave][h] = m[h]; wl[wave][h] = lsum[h]; }
        }
    }
    __syncthreads();
    for (int e = tid; e < REP * HD; e += 256) {
        const int h = e / HD, d = e - h * HD;
        const float M = fmaxf(fmaxf(wm[0][h], wm[1][h]), fmaxf(wm[2][h], wm[3][h]));
        float num = 0.f, den = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float ww = __expf(wm[w][h] - M);
            num = fmaf(ww, wo[w][h][d], num);
            den = fmaf(ww, wl[w][h], den);
        }
        DT<T>::st(out + e, num * (1.0f / den));
    }
}

// Split-KV merge of the talker attention as its own launch (one thread per 8 head dims per lane, all slot loads in flight
// at once), so that the o_proj after it is a plain GEMV: the merge inside the batch GEMV prologue walks 8 chunk tasks per
// thread with one dependent round trip each (11.7-13.4 us per launch); merge + plain o_proj measured 8.2 us for the pair and
// is bit-identical (tools/microbench/kernel_chain.hip).
template <typename T>
__global__ __launch_bounds__(256) void combine_batch_kernel(const float* part, size_t part_stride, int n_part, int rep, int q_dim,
                                                            T* out, int out_stride) {
    const int e0 = (blockIdx.x * 256 + threadIdx.x) * 8, l = blockIdx.y;
    if (e0 >= q_dim) return;
    CombineRegs cr;
    combine_load(cr, part + (size_t)l * part_stride, e0, rep, n_part);
    float f[8];
    combine_finish<T>(cr, n_part, f);
    DT<T>::st8(out + (size_t)l * out_stride + e0, f);
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void sample_pred_batch_kernel(const LaneTab* __restrict__ t, const LaneForced* __restrict__ f, const T* logits,
                                                               size_t logit_stride, int V, int cb,
                                                               int G, const T* next_emb, T* next_in, int H) {
    const int l = blockIdx.x;
    SampleCfg c{};                       // policy comes from the lane's DecodeState
    c.rep_penalty = 1.0f; c.sup_lo = 0; c.sup_hi = 0; c.keep_id = -1; c.sup_extra = -1;
    // NUCLEUS = true: a lane whose policy asks for top_p < 1 takes the LDS sampler (sampler.cuh::sample_core) inside the same
    // launch; the register-resident path of the other lanes is unchanged
    sample_pred_wave_body<T, NC, true>(gptr(t->st[l]), logits + (size_t)l * logit_stride, V, cb, c, (const T*)nullptr, gptr(t->codes[l]), G,
                                       (int64_t*)nullptr, next_emb, next_in + (size_t)l * H, H, gptr(f->tf[l]));
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void sample_talker_batch_kernel(const LaneTab* __restrict__ t, const LaneForced* __restrict__ f, const T* logits, int V, int G) {
    const int l = blockIdx.x;
    sample_talker_wave_body<T, NC, true>(gptr(t->st[l]), logits + (size_t)l * V, V, gptr(t->seen[l]), G, gptr(f->tf[l]));
}

// ---------------------------------------------------------------------------------------------------------------
struct BatchGemvArgs {
    const void* W; int N; int K; int B;
    const void* x; int x_stride;                   // T[B][x_stride]
    const void* norm_w; float eps;
    const void* bias;
    void* y; int y_stride;                         // T[B][y_stride]
    const void* res; int res_stride;               // T[B][res_stride]        (EPI_RESIDUAL), may alias y
    int up_off;
    int group;                                     // VALU kernel: tokens per LDS pass (1..kGroupLanes; the launcher sizes the LDS for it)
    void* const* xn_out;                           // optional per-lane copy of the prepared token (codec_head -> past_hidden): a device table, or null
    int ntiles;                                    // token tiles of 16 lanes (the matrix-core kernels' runtime-count variants, NT = 0)
    void* xn_ws;                                   // [B][K] workspace for pre-normalised tokens (above 32 lanes: rmsnorm_batch_kernel + weight-stationary GEMM), or null
    // RMSNorm folded into the weight-stationary GEMM pair (round 5, skinny_gemm.cuh): a residual GEMV (o_proj, down) leaves the sum of
    // squares of every row it stores as N / 16 partials per row in `ssq_out`; the normalising GEMV that reads those rows next takes them as
    // `ssq_in` (null: the rows came from elsewhere -- layer 0 -- and are normalised by rmsnorm_batch_kernel as before)
    float* ssq_out; const float* ssq_in;
    // fragment-major copy of W in 16-row blocks (skinny_gemm.cuh: SkinnyArgs::Wp, kind 0), or null: the matrix-core GEMVs' A-operand
    // loads then read contiguous kilobytes instead of 16 rows x 64 B (round 6; every byte comes from HBM here, where half lines cost
    // 15-30 %, tools/microbench/l2_rate_bench.hip).  Needs N % 16 == 0 (and up_off % 16 == 0).  Same values, same registers: bit-identical.
    const void* Wp;
};

// One row per wave (4 rows per workgroup), weight rows loaded ONCE; the B <= kMaxLanes tokens pass through LDS in groups of
// `a.group` <= 8 (so that B x K never has to fit the 160 KB LDS at once: 16 lanes, fp32, K = 6144 would need 393 KB).  Loops
// are unrolled to the group maximum and guarded by uniform branches, so one instantiation serves every batch size.
// tokens per LDS pass at most: 8, or 4 where 8 rows of K = 6144 fp32 would not fit the LDS anyway (and their staging registers --
// 2 tokens x 12 chunks x 8 fp32 per lane next to 96 registers of weights -- sent the kernel to scratch)
template <typename T> constexpr int batch_group_max_nch(int nch) { return (sizeof(T) == 4 && nch >= 12) ? 4 : kGroupLanes; }
template <typename T> inline int batch_group_max(int need_chunks) { return batch_group_max_nch<T>(need_chunks > 6 ? 12 : need_chunks); }

template <typename T, int NCH, int PRO, int EPI>
__global__ __launch_bounds__(256) void gemv_batch_kernel(BatchGemvArgs a) {
    static_assert(PRO == PRO_PLAIN || PRO == PRO_NORM, "the split-KV merge is its own launch in the batch chain (combine_batch_kernel)");
    constexpr int NR = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr int GT = batch_group_max_nch<T>(NCH), TPW = GT / 4;               // tokens prepared per wave and group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* xs = reinterpret_cast<T*>(smem_raw);                     // [group][K], normalised + rounded
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int K = a.K, B = a.B, group = a.group;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int row = blockIdx.x * 4 + wave;
    const int rowc = row < a.N ? row : a.N - 1;

    // ---- 1. token loads of the first group: wave w prepares group tokens w and w + 4 (clamped, unconditional) ----
    Raw8<T> xraw[TPW][NCH], nraw[NCH];
    auto issue_tokens = [&](int g0, int nb) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const int m = g0 + (wave + 4 * t < nb ? wave + 4 * t : nb - 1);
                ldraw<false>(xraw[t][j], reinterpret_cast<const T*>(a.x) + (size_t)m * a.x_stride + offc);
            }
        }
    };
    issue_tokens(0, B < group ? B : group);
    if constexpr (PRO == PRO_NORM) {
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8;
            ldraw<false>(nraw[j], reinterpret_cast<const T*>(a.norm_w) + (off < K ? off : 0));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. this wave's weight row(s) ----
    Raw8<T> raw[NR][NCH];
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        const T* wr = W + (size_t)(rowc + h * a.up_off) * K;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8;
            ldraw<false>(raw[h][j], wr + (off < K ? off : 0));
        }
    }
    const T* bp = a.bias ? reinterpret_cast<const T*>(a.bias) + rowc : W;
    const float bv = DT<T>::ld(bp);
    const float biasv = a.bias ? bv : 0.f;
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int g0 = 0; g0 < B; g0 += group) {
        const int nb = B - g0 < group ? B - g0 : group;
        if (g0 > 0) issue_tokens(g0, nb);                       // later groups pay one exposed round trip (B > 8 only)
        float resv[GT];
#pragma unroll
        for (int m = 0; m < GT; ++m) {
            resv[m] = 0.f;
            if constexpr (EPI == EPI_RESIDUAL) {
                const int mc = g0 + (m < nb ? m : nb - 1);
                resv[m] = DT<T>::ld(reinterpret_cast<const T*>(a.res) + (size_t)mc * a.res_stride + rowc);
            }
        }
        // ---- 3. prepare this group's tokens (the first group: while the weights fly); stage them in LDS ----
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
            if (m < nb) {
                T* xo = (blockIdx.x == 0 && a.xn_out) ? gptr(reinterpret_cast<T*>(a.xn_out[g0 + m])) : nullptr;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int off = j * 512 + lane * 8;
                    if (off < K) {
                        DT<T>::st8(xs + (size_t)m * K + off, xr[j]);
                        if (xo) DT<T>::st8(xo + off, xr[j]);
                    }
                }
            }
        }
        __syncthreads();
        // ---- 4. walk the group's tokens: same chunk order and fma chain as gemv_kernel's dot8 ----
        float acc[GT][NR];
#pragma unroll
        for (int m = 0; m < GT; ++m) {
#pragma unroll
            for (int h = 0; h < NR; ++h) acc[m][h] = 0.f;
            if (m < nb) {
                float xr[NCH][8];
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int off = j * 512 + lane * 8;
                    Raw8<T> q;
                    ldraw<false>(q, xs + (size_t)m * K + (off < K ? off : 0));
                    if (off >= K) zero(q);
                    unpack(q, xr[j]);
                }
#pragma unroll
                for (int h = 0; h < NR; ++h) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < NCH; ++j) s = dot8<T>(raw[h][j], xr[j], s);
                    acc[m][h] = s;
                }
            }
        }
#pragma unroll
        for (int m = 0; m < GT; ++m)
#pragma unroll
            for (int h = 0; h < NR; ++h) acc[m][h] = wave_sum(acc[m][h]);
#pragma unroll
        for (int m = 0; m < GT; ++m) {
            float v;
            if constexpr (EPI == EPI_SWIGLU) {
                const float g = DT<T>::rnd(acc[m][0]);
                const float u = DT<T>::rnd(acc[m][NR - 1]);
                const float sg = DT<T>::rnd(g / (1.0f + expf(-g)));
                v = sg * u;
            } else {
                v = DT<T>::rnd(acc[m][0] + biasv);
                if constexpr (EPI == EPI_RESIDUAL) v = v + resv[m];
            }
            if (lane == 0 && row < a.N && m < nb) DT<T>::st(reinterpret_cast<T*>(a.y) + (size_t)(g0 + m) * a.y_stride + row, v);
        }
        __syncthreads();                                        // the next group overwrites the LDS tokens
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The same M = B GEMV on the matrix cores (bf16; the default).  One workgroup = one tile of 16 weight rows (x NR for
// SwiGLU); the waves split K; v_mfma_f32_16x16x32_bf16 with A = weights (lane: row = lane & 15, k group = lane >> 4: a
// 16-byte global load per lane IS the operand layout, no LDS hop), B = tokens (lane: token = lane & 15, same k group),
// C[row = (lane >> 4) * 4 + reg][token]; layouts as in conv_gemm_kernel.  The tile carries 16 token columns whatever B is,
// so 9..16 lanes cost the weight stream and the MFMA count of 1..8.  The fp32 accumulation order differs from the VALU
// kernels (<= ~1e-4 relative), so lanes are not bit-identical to single-stream decoding: parity for this path is against
// the oracle with the bf16 tolerances (tests/test_gpu_batch.py, tests/test_gpu_fulldepth.py).
//   gemv_batch_mfma_norm_kernel   RMSNorm prologue (qkv, gate|up, heads; K = hidden <= 2048): tokens are normalised
//                                 cooperatively (wave w: tokens w, w + 4, w + 8, w + 12) into a padded LDS panel
//   gemv_batch_mfma_plain_kernel  no prologue (o_proj, down, projection; K up to 6144): the token fragments are loaded
//                                 straight from global memory in operand layout -- no LDS panel (16 x 6144 bf16 would not
//                                 fit), no barrier before the MFMAs; NW = 4 or 8 waves split K
// Columns of tokens >= B hold a duplicate of token B - 1 and are never stored (an MFMA column depends on its own B column only).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));

// Above 32 lanes the normalising GEMVs (qkv, gate | up, heads) stop paying for themselves: every one of their 192-256 workgroups
// normalises ALL the batch's tokens and walks the token tiles one pair after the other (19.6 / 21.2 us per launch at 128 lanes,
// 54 % of the frame, profiles/r04_batch128_panel_kernel_trace.txt).  There the tokens are normalised ONCE by this kernel -- one wave per
// token, the very operations of gemv_batch_mfma_norm_kernel's prologue in the same order (a lane owns 8 consecutive elements of
// every 512-chunk, one sequential fma chain, wave_sum): the rows are bit-identical to the panels that kernel builds -- and the GEMM
// runs on the prefill's weight-stationary kernel (skinny_gemm.cuh: SK_STORE / SK_SWIGLU), which streams token tiles past
// register-resident weight rows on every CU.  Only the fp32 summation order of the products changes (8 waves x two chains instead
// of 4 waves x one), which is why the switch sits at a lane count of its own: up to 32 lanes -- the configurations whose parity floors
// were frozen in round 3 -- nothing moves.  Measured (profiles/r04_batch_norm_skinny.txt): 0.6B 48 lanes 5.18 -> 4.99 ms per frame, 64 lanes
// 5.79 -> 5.14, 128 lanes 8.39 -> 6.21; 1.7B 64 lanes 7.37 -> 5.93, 128 lanes 11.1 -> 7.48; at 16 / 32 lanes the panel kernels win (3.87 vs 4.54,
// 4.43 vs 4.72 ms).
template <int NCH>
__global__ __launch_bounds__(256) void rmsnorm_batch_kernel(const bf16_t* __restrict__ x, int x_stride, const bf16_t* __restrict__ norm_w, float eps,
                                                            int K, int B, bf16_t* __restrict__ y, int y_stride, void* const* xn_out) {
    typedef bf16_t T;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    if (m >= B) return;
    Raw8<T> xraw[NCH], nraw[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
        ldraw<false>(xraw[j], x + (size_t)m * x_stride + offc);
        ldraw<false>(nraw[j], norm_w + offc);
    }
    float xr[NCH][8];
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        if (j * 512 + lane * 8 >= K) zero(xraw[j]);
        unpack(xraw[j], xr[j]);
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) ss = fmaf(xr[j][i], xr[j][i], ss);
    ss = wave_sum(ss);
    const float rs = 1.0f / sqrtf(ss / (float)K + eps);
    T* xo = xn_out ? gptr(reinterpret_cast<T*>(xn_out[m])) : nullptr;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        float nw[8];
        unpack(nraw[j], nw);
        const u32x4 v = norm8_pack(xr[j], rs, nw);
        const int off = j * 512 + lane * 8;
        if (off < K) {
            *reinterpret_cast<u32x4*>(y + (size_t)m * y_stride + off) = v;
            if (xo) *reinterpret_cast<u32x4*>(xo + off) = v;
        }
    }
}

// NT = token tiles of 16 lanes the launch walks (1: B <= 16, 2: 17..32).  The weight fragments are loaded once and stay in
// registers; with NT = 2 the second tile's raw tokens are loaded up front as well when they fit (PRE2: K <= 1024), otherwise after
// the first tile has been multiplied (one exposed L2 round trip).
// DUAL (two token tiles, K <= 1024; round 4): BOTH tiles' tokens are normalised into their own LDS panels before the first MFMA --
// i.e. while the weight fragments are still in flight, where the first tile's normalisation already ran -- then both tiles are
// multiplied back to back, the K quarters of both meet in ONE exchange, and waves 0 / 1 run the two epilogues side by side.  The
// second tile's ~1 us of normalisation arithmetic, two of the five barriers and one exposed pass over the partial sums leave the
// critical path; every value is computed by the same instructions in the same order as in the one-panel form (bit-identical:
// tests/test_gpu_batch.py).  2 x 33 KB of LDS at K = 1024.
// Three and four token tiles (33..64 lanes, round 4) repeat the two-panel scheme per PAIR of tiles: the next pair's raw tokens are
// requested as soon as this pair's have been normalised and fly under its MFMAs and exchange.
template <int KSTEPS, int EPI, int NT, bool DUAL = false>            // K = KSTEPS * 128
__global__ __launch_bounds__(256) void gemv_batch_mfma_norm_kernel(BatchGemvArgs a) {
    typedef bf16_t T;
    constexpr int NR = (EPI == EPI_SWIGLU) ? 2 : 1;
    constexpr bool PRE2 = (NT == 2 || (DUAL && NT != 1)) && KSTEPS <= 8;
    static_assert(!DUAL || PRE2, "the two-panel form needs a pair of tiles' raw tokens in registers");
    constexpr int NXR = PRE2 ? 2 : 1;
    constexpr int K = KSTEPS * 128, KP = K + 8;                 // padded LDS row: 16 tokens x same k would share a bank
    constexpr int NCH = (K + 511) / 512;
    constexpr int TPW = kTokTile / 4;
    // K = 2048 with the two weight-row halves of SwiGLU (1.7B gate | up at <= 32 lanes): 128 registers of weight fragments + the raw
    // rows of four tokens (64) + gains do not fit 256 VGPRs (272-294: one wave per SIMD, i.e. one workgroup per CU for a grid of 384).
    // There a wave's four tokens are fetched and normalised in TWO rounds of two through the same registers (TPI tokens per round); the
    // second round's L2 round trip lies behind the weight rows that are in flight anyway.  Per token the same instructions: bit-identical.
    constexpr int TPI = (KSTEPS >= 16 && NR == 2) ? TPW / 2 : TPW, NRND = TPW / TPI;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int NPANEL = DUAL ? 2 : 1;
    T* xs = reinterpret_cast<T*>(smem_raw);                     // [NPANEL][16][KP]
    float* red = reinterpret_cast<float*>(smem_raw + (((size_t)NPANEL * kTokTile * KP * sizeof(T) + 15) & ~(size_t)15));   // [NPANEL][4][NR][64][4]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4, B = a.B;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int row0 = blockIdx.x * 16;

    // ---- 1. token loads of the first token tile (lanes 0..15 of the batch), and of the second when it fits ----
    Raw8<T> xraw[NXR][TPI][NCH], nraw[NCH];
    auto issue_tokens = [&](int slot, int t0, int nb, int rnd = 0) {       // tokens wave + 4 (rnd TPI + t), t < TPI, of the tile at t0
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int off = j * 512 + lane * 8, offc = off < K ? off : 0;
#pragma unroll
            for (int t = 0; t < TPI; ++t) {
                const int tk = wave + 4 * (rnd * TPI + t);
                const int m = t0 + (tk < nb ? tk : nb - 1);
                ldraw<false>(xraw[slot][t][j], reinterpret_cast<const T*>(a.x) + (size_t)m * a.x_stride + offc);
            }
        }
    };
    auto tile_nb = [&](int tile) { const int n = B - tile * kTokTile; return n < kTokTile ? (n < 1 ? 1 : n) : kTokTile; };
    issue_tokens(0, 0, tile_nb(0));
    if constexpr (PRE2) issue_tokens(1, kTokTile, tile_nb(1));
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int off = j * 512 + lane * 8;
        ldraw<false>(nraw[j], reinterpret_cast<const T*>(a.norm_w) + (off < K ? off : 0));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- 2. this wave's K quarter of the 16 (x NR) weight rows, in A-operand layout: loaded ONCE, kept for every token tile ----
    Raw8<T> wreg[NR][KSTEPS];
    const int rowc = row0 + fr < a.N ? row0 + fr : a.N - 1;
    const T* Wp = reinterpret_cast<const T*>(a.Wp);
#pragma unroll
    for (int h = 0; h < NR; ++h) {
        // row-major: row rowc (+ the up half), this wave's K quarter; fragment-major: block row0 / 16 (+ up_off / 16), K steps wave * KSTEPS + s
        const T* wp = Wp ? Wp + ((size_t)(blockIdx.x + h * (a.up_off >> 4)) * (K / 32) + wave * KSTEPS) * 512 + lane * 8
                         : W + (size_t)(rowc + h * a.up_off) * K + wave * (K / 4) + fq * 8;
        const int st_s = Wp ? 512 : 32;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) ldraw<false>(wreg[h][s], wp + s * st_s);
    }
    // epilogue operands of wave 0: token = fr, rows row0 + fq * 4 + i
    float biasv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = row0 + fq * 4 + i < a.N ? row0 + fq * 4 + i : a.N - 1;
        const T* bp = a.bias ? reinterpret_cast<const T*>(a.bias) + r : W;
        const float bv = DT<T>::ld(bp);
        biasv[i] = a.bias ? bv : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    // NT = 0: the tile count is a launch argument (a.ntiles; 5..8 tiles = 65..128 lanes) and the loop over pairs / tiles stays rolled --
    // one instantiation instead of four more per shape, same instructions per tile (NT > 0: a constant trip count, fully unrolled)
    const int ntl = NT ? NT : a.ntiles;
    constexpr int kTileUnroll = NT > 0 ? 8 : 1;
    if constexpr (DUAL) {
        const int npair = (ntl + 1) / 2;
#pragma unroll kTileUnroll
        for (int pr = 0; pr < npair; ++pr) {
            // ---- 3'. this pair's tokens -> the two panels (first pair: while the weights fly) ----
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (2 * pr + tt >= ntl) continue;                // (compile-time after unrolling: an odd tile count has a single last tile)
                const int t0 = (2 * pr + tt) * kTokTile;
                const int nb = tile_nb(2 * pr + tt);
                T* xp = xs + (size_t)tt * kTokTile * KP;
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
                    const int m = wave + 4 * t;
                    float xr[NCH][8];
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        if (j * 512 + lane * 8 >= K) zero(xraw[tt][t][j]);
                        unpack(xraw[tt][t][j], xr[j]);
                    }
                    float ss = 0.f;
#pragma unroll
                    for (int j = 0; j < NCH; ++j)
#pragma unroll
                        for (int i = 0; i < 8; ++i) ss = fmaf(xr[j][i], xr[j][i], ss);
                    ss = wave_sum(ss);
                    const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);
                    u32x4 xn[NCH];
#pragma unroll
                    for (int j = 0; j < NCH; ++j) {
                        float nw[8];
                        unpack(nraw[j], nw);
                        xn[j] = norm8_pack(xr[j], rs, nw);
                    }
                    if (m < nb) {
                        T* xo = (blockIdx.x == 0 && a.xn_out) ? gptr(reinterpret_cast<T*>(a.xn_out[t0 + m])) : nullptr;
#pragma unroll
                        for (int j = 0; j < NCH; ++j) {
                            const int off = j * 512 + lane * 8;
                            if (off < K) {
                                *reinterpret_cast<u32x4*>(xp + (size_t)m * KP + off) = xn[j];
                                if (xo) *reinterpret_cast<u32x4*>(xo + off) = xn[j];
                            }
                        }
                    }
                }
            }
            // the next pair's raw tokens: in flight under this pair's MFMAs and exchange
            if (pr + 1 < npair) {
                issue_tokens(0, (2 * pr + 2) * kTokTile, tile_nb(2 * pr + 2));
                if (2 * pr + 3 < ntl) issue_tokens(1, (2 * pr + 3) * kTokTile, tile_nb(2 * pr + 3));
            }
            __syncthreads();            // panels published; (from the second pair on) the previous pair's epilogues are done with `red`
            // ---- 4'. MFMA over this wave's K quarter, both tiles; 5'. one exchange, two epilogues side by side ----
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                if (2 * pr + tt >= ntl) continue;
                const int nb = tile_nb(2 * pr + tt);
                const T* xp = xs + (size_t)tt * kTokTile * KP;
                f32x4 acc[NR];
#pragma unroll
                for (int h = 0; h < NR; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int tokc = fr < nb ? fr : nb - 1;
#pragma unroll
                for (int s = 0; s < KSTEPS; ++s) {
                    const u32x4 bq = *reinterpret_cast<const u32x4*>(xp + (size_t)tokc * KP + wave * (K / 4) + s * 32 + fq * 8);
                    const mfma_bf16x8 bfrag = __builtin_bit_cast(mfma_bf16x8, bq);
#pragma unroll
                    for (int h = 0; h < NR; ++h)
                        acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, wreg[h][s].v), bfrag, acc[h], 0, 0, 0);
                }
#pragma unroll
                for (int h = 0; h < NR; ++h) *reinterpret_cast<f32x4*>(red + (((size_t)tt * 4 + wave) * NR + h) * 256 + (size_t)lane * 4) = acc[h];
            }
            __syncthreads();            // every wave is done with the panels and has written its partial sums
            if (wave < 2 && 2 * pr + wave < ntl) {
                const int tt = wave, t0 = (2 * pr + tt) * kTokTile;
                const int nb = tile_nb(2 * pr + tt);
                if (fr < nb) {
                    float tot[NR][4];
#pragma unroll
                    for (int h = 0; h < NR; ++h) {
                        f32x4 t = *reinterpret_cast<const f32x4*>(red + (((size_t)tt * 4 + 0) * NR + h) * 256 + (size_t)lane * 4);
#pragma unroll
                        for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + (((size_t)tt * 4 + w) * NR + h) * 256 + (size_t)lane * 4);
                        tot[h][0] = t.x; tot[h][1] = t.y; tot[h][2] = t.z; tot[h][3] = t.w;
                    }
                    T* yp = reinterpret_cast<T*>(a.y) + (size_t)(t0 + fr) * a.y_stride + row0 + fq * 4;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        float v;
                        if constexpr (EPI == EPI_SWIGLU) {
                            const float g = DT<T>::rnd(tot[0][i]);
                            const float u = DT<T>::rnd(tot[NR - 1][i]);
                            const float sg = DT<T>::rnd(g / (1.0f + expf(-g)));
                            v = sg * u;
                        } else {
                            v = DT<T>::rnd(tot[0][i] + biasv[i]);
                        }
                        if (row0 + fq * 4 + i < a.N) DT<T>::st(yp + i, v);
                    }
                }
            }
        }
        return;
    }
#pragma unroll kTileUnroll
    for (int tt = 0; tt < ntl; ++tt) {                          // one pass per token tile (a second one only above 16 lanes)
        const int t0 = tt * kTokTile;
        const int nb = B - t0 < kTokTile ? B - t0 : kTokTile;
        constexpr int kNoSlot = 0;
        const int slot = PRE2 ? tt : kNoSlot;
        if (tt > 0 && !PRE2) issue_tokens(0, t0, nb);           // one exposed round trip; the weights are already here
        // ---- 3. prepare the tile's tokens (the first tile: while the weights fly) ----
#pragma unroll
        for (int tq = 0; tq < TPW; ++tq) {
            const int t = tq % TPI;                             // the register set of token tq
            if (NRND > 1 && tq > 0 && t == 0) issue_tokens(slot, t0, nb, tq / TPI);      // the next round of tokens through the same registers
            const int m = wave + 4 * tq;
            float xr[NCH][8];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                if (j * 512 + lane * 8 >= K) zero(xraw[slot][t][j]);
                unpack(xraw[slot][t][j], xr[j]);
            }
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < NCH; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) ss = fmaf(xr[j][i], xr[j][i], ss);
            ss = wave_sum(ss);
            const float rs = 1.0f / sqrtf(ss / (float)K + a.eps);
            u32x4 xn[NCH];
#pragma unroll
            for (int j = 0; j < NCH; ++j) {
                float nw[8];
                unpack(nraw[j], nw);
                xn[j] = norm8_pack(xr[j], rs, nw);
            }
            if (m < nb) {
                T* xo = (blockIdx.x == 0 && a.xn_out) ? gptr(reinterpret_cast<T*>(a.xn_out[t0 + m])) : nullptr;
#pragma unroll
                for (int j = 0; j < NCH; ++j) {
                    const int off = j * 512 + lane * 8;
                    if (off < K) {
                        *reinterpret_cast<u32x4*>(xs + (size_t)m * KP + off) = xn[j];
                        if (xo) *reinterpret_cast<u32x4*>(xo + off) = xn[j];
                    }
                }
            }
        }
        __syncthreads();
        // ---- 4. MFMA over this wave's K quarter ----
        f32x4 acc[NR];
#pragma unroll
        for (int h = 0; h < NR; ++h) acc[h] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int tokc = fr < nb ? fr : nb - 1;
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const u32x4 bq = *reinterpret_cast<const u32x4*>(xs + (size_t)tokc * KP + wave * (K / 4) + s * 32 + fq * 8);
            const mfma_bf16x8 bfrag = __builtin_bit_cast(mfma_bf16x8, bq);
#pragma unroll
            for (int h = 0; h < NR; ++h)
                acc[h] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, wreg[h][s].v), bfrag, acc[h], 0, 0, 0);
        }
        // ---- 5. sum the four K quarters (fixed order), epilogue on wave 0 ----
#pragma unroll
        for (int h = 0; h < NR; ++h) *reinterpret_cast<f32x4*>(red + ((size_t)(wave * NR + h) * 64 + lane) * 4) = acc[h];
        __syncthreads();
        if (wave == 0 && fr < nb) {
            float tot[NR][4];
#pragma unroll
            for (int h = 0; h < NR; ++h) {
                f32x4 t = *reinterpret_cast<const f32x4*>(red + ((size_t)(0 * NR + h) * 64 + lane) * 4);
#pragma unroll
                for (int w = 1; w < 4; ++w) t += *reinterpret_cast<const f32x4*>(red + ((size_t)(w * NR + h) * 64 + lane) * 4);
                tot[h][0] = t.x; tot[h][1] = t.y; tot[h][2] = t.z; tot[h][3] = t.w;
            }
            T* yp = reinterpret_cast<T*>(a.y) + (size_t)(t0 + fr) * a.y_stride + row0 + fq * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v;
                if constexpr (EPI == EPI_SWIGLU) {
                    const float g = DT<T>::rnd(tot[0][i]);
                    const float u = DT<T>::rnd(tot[NR - 1][i]);
                    const float sg = DT<T>::rnd(g / (1.0f + expf(-g)));
                    v = sg * u;
                } else {
                    v = DT<T>::rnd(tot[0][i] + biasv[i]);
                }
                if (row0 + fq * 4 + i < a.N) DT<T>::st(yp + i, v);
            }
        }
        if (tt + 1 < ntl) __syncthreads();                      // the next tile overwrites the token panel and the partial sums
    }
}

template <int KSTEPS, int NW, int EPI, int NT>    // K = KSTEPS * 32 * NW; NW waves split K; NT token tiles (see the NORM kernel)
__global__ __launch_bounds__(64 * NW) void gemv_batch_mfma_plain_kernel(BatchGemvArgs a) {
    typedef bf16_t T;
    static_assert(EPI == EPI_STORE || EPI == EPI_RESIDUAL, "SwiGLU always follows an RMSNorm prologue");
    constexpr int K = KSTEPS * 32 * NW;
    constexpr bool PRE2 = NT == 2 && KSTEPS <= 12;              // both tiles' token fragments in registers from the start
    constexpr int NBR = PRE2 ? 2 : 1;
    __shared__ __attribute__((aligned(16))) float red[NW * 64 * 4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int fr = lane & 15, fq = lane >> 4, B = a.B;
    const T* W = reinterpret_cast<const T*>(a.W);
    const int row0 = blockIdx.x * 16;
    // ---- 1. token fragments (small, L2-resident) of the first token tile first, then the weight fragments: vmcnt retires in
    //         order, so the wait before MFMA step s (its weight fragment) also covers every token fragment ----
    // token fragments held at once: all of a tile's K share, except in the longest variant (K = 6144: 24 steps per wave), where two
    // halves take turns in the same registers (96 weight + 96 token registers plus the rest would not fit 256 VGPRs)
    constexpr int BCH = KSTEPS > 16 ? KSTEPS / 2 : KSTEPS, NBCH = KSTEPS / BCH;
    Raw8<T> breg[NBR][BCH], wreg[KSTEPS];
    auto issue_tokens = [&](int slot, int t0, int nb, int chunk) {
        const int tokc = t0 + (fr < nb ? fr : nb - 1);
        const T* xp = reinterpret_cast<const T*>(a.x) + (size_t)tokc * a.x_stride + wave * (K / NW) + fq * 8 + chunk * BCH * 32;
#pragma unroll
        for (int s = 0; s < BCH; ++s) ldraw<false>(breg[slot][s], xp + s * 32);
    };
    issue_tokens(0, 0, B < kTokTile ? B : kTokTile, 0);
    if constexpr (PRE2) issue_tokens(1, kTokTile, B - kTokTile, 0);
    __builtin_amdgcn_sched_barrier(0);
    const int rowc = row0 + fr < a.N ? row0 + fr : a.N - 1;
    const T* Wpk = reinterpret_cast<const T*>(a.Wp);
    const T* wp = Wpk ? Wpk + ((size_t)blockIdx.x * (K / 32) + wave * KSTEPS) * 512 + lane * 8        // fragment-major: block row0 / 16, K steps wave * KSTEPS + s
                      : W + (size_t)rowc * K + wave * (K / NW) + fq * 8;
    const int st_s = Wpk ? 512 : 32;
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) ldraw<false>(wreg[s], wp + s * st_s);
    // epilogue operands (bias, residual): issued up front so that they fly with the weights -- except in the longest variant
    // (K = 6144: 48 operand registers per K step pair already fill the 256-VGPR budget), where wave 0 fetches them at the end
    constexpr bool kHoist = KSTEPS <= 16;
    auto load_epi = [&](int tokc, float (&biasv)[4], float (&resv)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + fq * 4 + i < a.N ? row0 + fq * 4 + i : a.N - 1;
            const T* bp = a.bias ? reinterpret_cast<const T*>(a.bias) + r : W;
            const float bv = DT<T>::ld(bp);
            biasv[i] = a.bias ? bv : 0.f;
            resv[i] = 0.f;
            if constexpr (EPI == EPI_RESIDUAL) resv[i] = DT<T>::ld(reinterpret_cast<const T*>(a.res) + (size_t)tokc * a.res_stride + r);
        }
    };
    __builtin_amdgcn_sched_barrier(0);
    const int ntl = NT ? NT : a.ntiles;                         // NT = 0: runtime tile count, rolled loop (see the NORM kernel)
    constexpr int kTileUnroll = NT > 0 ? 8 : 1;
#pragma unroll kTileUnroll
    for (int tt = 0; tt < ntl; ++tt) {                          // one pass per token tile; the weight fragments stay in registers
        const int t0 = tt * kTokTile;
        const int nb = B - t0 < kTokTile ? B - t0 : kTokTile;
        const int slot = PRE2 ? tt : 0;
        if (tt > 0 && !PRE2) issue_tokens(0, t0, nb, 0);
        const int tokc = t0 + (fr < nb ? fr : nb - 1);
        float biasv[4], resv[4];
        if constexpr (kHoist) load_epi(tokc, biasv, resv);
        // ---- 2. MFMA over this wave's K share ----
        f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ch = 0; ch < NBCH; ++ch) {
            if (ch > 0) issue_tokens(slot, t0, nb, ch);
#pragma unroll
            for (int s = 0; s < BCH; ++s)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf16x8, wreg[ch * BCH + s].v), __builtin_bit_cast(mfma_bf16x8, breg[slot][s].v), acc, 0, 0, 0);
        }
        // ---- 3. sum the NW shares (fixed order), epilogue on wave 0 ----
        *reinterpret_cast<f32x4*>(red + ((size_t)wave * 64 + lane) * 4) = acc;
        __syncthreads();
        if (wave == 0 && fr < nb) {
            if constexpr (!kHoist) load_epi(tokc, biasv, resv);
            f32x4 t = *reinterpret_cast<const f32x4*>(red + (size_t)lane * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) t += *reinterpret_cast<const f32x4*>(red + ((size_t)w * 64 + lane) * 4);
            const float tot[4] = {t.x, t.y, t.z, t.w};
            T* yp = reinterpret_cast<T*>(a.y) + (size_t)(t0 + fr) * a.y_stride + row0 + fq * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = DT<T>::rnd(tot[i] + biasv[i]);
                if constexpr (EPI == EPI_RESIDUAL) v = v + resv[i];
                if (row0 + fq * 4 + i < a.N) DT<T>::st(yp + i, v);
            }
        }
        if (tt + 1 < ntl) __syncthreads();                      // the next tile's partial sums reuse `red`
    }
}

}  // namespace fq3
