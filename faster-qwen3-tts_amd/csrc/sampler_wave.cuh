// Register-resident sampler (default path, top_p >= 1): the vocabulary row lives in the registers of ONE
// 4-wave workgroup (8 contiguous ids per thread per 2048-id chunk); one global round trip for logits, the
// noise lands behind the k-th-value search; a handful of workgroup barriers in total.  Same arithmetic and rounding as sample_core (sampler.cuh) / sampling.py:32-66.
#pragma once
#include "sampler.cuh"

namespace fq3 {

// larger value wins, ties -> lower index (amax2), as two DPP reductions: value max, then min index among its holders
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
    ArgMax r;
    r.v = wave_max(a.v);
    r.i = wave_min_i32(a.v == r.v ? a.i : 0x7fffffff);
    return r;
}

// Scratch for the 4-wave sampler: tiny, so the kernels stay at high occupancy-irrelevant size.
struct BlkSmem { int cnt[2][4]; float f[4]; float av[4]; int ai[4]; };

__device__ __forceinline__ float blk_max4(float v, BlkSmem& sm) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm.f[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(sm.f[0], sm.f[1]), fmaxf(sm.f[2], sm.f[3]));
}
__device__ __forceinline__ float blk_sum4(float v, BlkSmem& sm) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm.f[threadIdx.x >> 6] = v;
    __syncthreads();
    return (sm.f[0] + sm.f[1]) + (sm.f[2] + sm.f[3]);
}
__device__ __forceinline__ int blk_argmax4(ArgMax a, BlkSmem& sm) {
    a = wave_argmax(a);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sm.av[threadIdx.x >> 6] = a.v; sm.ai[threadIdx.x >> 6] = a.i; }
    __syncthreads();
    ArgMax r; r.v = sm.av[0]; r.i = sm.ai[0];
#pragma unroll
    for (int i = 1; i < 4; ++i) { ArgMax b; b.v = sm.av[i]; b.i = sm.ai[i]; r = amax2(r, b); }
    return r.i;
}

// 256 threads, 8 contiguous ids per thread per 2048-id chunk (NC = ceil(V / 2048) chunks in registers).
// logits already issued into xraw by the caller (so that state loads overlap); returns token (all threads)
template <typename T, int NC>
__device__ int sample_wave_core(Raw8<T> (&xraw)[NC], int V, const SampleCfg& c, const unsigned char* seen,
                                const T* noise, BlkSmem& sm) {
    const int tid = threadIdx.x;
    float x[NC][8];
    Raw8<T> nz[NC];
    const bool samp = c.do_sample != 0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int off = j * 2048 + tid * 8;
        if (samp && off < V) ldraw<false>(nz[j], noise + off); else zero(nz[j]);
    }
    const bool pen = seen && c.rep_penalty != 1.0f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int off = j * 2048 + tid * 8;
        unpack(xraw[j], x[j]);
        unsigned long long sb = 0ull;
        if (pen && off < V) sb = *reinterpret_cast<const unsigned long long*>(seen + off);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = off + i;
            float v = x[j][i];
            if (pen && ((sb >> (8 * i)) & 0xffull)) v = v > 0.f ? DT<T>::rnd(v / c.rep_penalty) : DT<T>::rnd(v * c.rep_penalty);
            if (id >= V || (id >= c.sup_lo && id < c.sup_hi && id != c.keep_id) || id == c.sup_extra) v = -INFINITY;
            x[j][i] = v;
        }
    }
    if (!samp) {
        ArgMax a; a.v = -INFINITY; a.i = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) { ArgMax b; b.v = x[j][i]; b.i = j * 2048 + tid * 8 + i; a = amax2(a, b); }
        return blk_argmax4(a, sm);
    }
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[j][i] = DT<T>::rnd(x[j][i] / c.temperature);
    if (c.top_k > 0) {
        // k-th largest key by MSB-first bitwise search: count(key >= candidate) = ballot popcounts per wave,
        // the four wave counts meet in double-buffered LDS slots (one barrier per bit).  An LDS histogram
        // serialises badly here (the top byte of the keys is nearly constant -> 64-way atomic conflicts), and a
        // 4-bits-per-round variant (15 candidate digits, 120 ballots per round, 4 barriers) measured 15.2 us against
        // 6.9 us for this loop: a ballot + popcount costs more than the barrier it saves; per-lane counts + one DPP
        // reduction per round instead of the 8 ballots measured 7.2 us.
        const int kk = min(c.top_k, V);
        uint32_t key[NC][8];
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) key[j][i] = (j * 2048 + tid * 8 + i) < V ? okey(x[j][i]) : 0u;
        uint32_t prefix = 0;
        constexpr int kLowBit = sizeof(T) == 2 ? 16 : 0;       // bf16 values live in the top 16 bits
        const int wave = tid >> 6;
#pragma unroll 1
        for (int bit = 31; bit >= kLowBit; --bit) {
            const uint32_t cand = prefix | (1u << bit);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < NC; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) cnt += __popcll(__ballot(key[j][i] >= cand));
            int* slot = sm.cnt[bit & 1];
            if ((tid & 63) == 0) slot[wave] = cnt;
            __syncthreads();
            const int tot = (slot[0] + slot[1]) + (slot[2] + slot[3]);
            if (tot >= kk) prefix = cand;
            // exactly k keys at or above the candidate: they ARE the top k (the k-th largest is >= cand, the next one below it), and
            // the filter `key >= prefix` keeps the same set as the fully resolved threshold would -- the remaining bits need no rounds
            if (tot == kk) break;
        }
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (key[j][i] < prefix) x[j][i] = -INFINITY;
    }
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) mx = fmaxf(mx, x[j][i]);
    mx = blk_max4(mx, sm);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[j][i] = expf(x[j][i] - mx); se += x[j][i]; }
    se = blk_sum4(se, sm);
    ArgMax a; a.v = -INFINITY; a.i = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        float q[8];
        unpack(nz[j], q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = j * 2048 + tid * 8 + i;
            const float p = DT<T>::rnd(x[j][i] / se);
            ArgMax b; b.v = id < V ? DT<T>::rnd(p / q[i]) : -INFINITY; b.i = id;
            a = amax2(a, b);
        }
    }
    return blk_argmax4(a, sm);
}

template <typename T, int NC>
__device__ __forceinline__ void issue_logits(Raw8<T> (&xraw)[NC], const T* logits, int V) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int off = j * 2048 + threadIdx.x * 8;
        if (off < V) ldraw<false>(xraw[j], logits + off); else zero(xraw[j]);
    }
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void sample_api_wave_kernel(const T* logits, int V, SampleCfg c, const unsigned char* seen,
                                                             const T* noise, int64_t* out) {
    __shared__ BlkSmem sm;
    Raw8<T> xraw[NC];
    issue_logits<T, NC>(xraw, logits, V);
    const int tok = sample_wave_core<T, NC>(xraw, V, c, seen, noise, sm);
    if (threadIdx.x == 0) out[0] = tok;
}

// NUCLEUS (batched lanes): when the policy found in the loop state asks for top_p < 1 the logits already in registers are
// spilled to LDS and the workgroup sampler of sampler.cuh (stable sort + cumulative cut, sampling.py:57-65) takes over; the
// branch is uniform per workgroup (= per lane of the batch), so lanes with and without nucleus sampling share one launch.
template <typename T, int NC>
__device__ __forceinline__ int sample_nucleus_from_regs(Raw8<T> (&xraw)[NC], int V, const SampleCfg& c, const unsigned char* seen,
                                                        const T* noise, SampleSmem& big) {
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        float f[8];
        unpack(xraw[j], f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = j * 2048 + threadIdx.x * 8 + i;
            if (id < V) big.vals[id] = f[i];
        }
    }
    __syncthreads();
    const int tok = sample_core<T>(big, V, c, seen, noise);
    __syncthreads();
    return tok;
}

template <typename T, int NC, bool NUCLEUS = false>
__device__ __forceinline__ void sample_pred_wave_body(const DecodeState* st, const T* logits, int V, int cb,
                                                      const SampleCfg& c_imm, const T* noise_imm, int* codes, int G,
                                                      int64_t* out64, const T* next_emb, T* next_in, int H, const TeacherForcing* tf = nullptr) {
    __shared__ BlkSmem sm;
    Raw8<T> xraw[NC];
    issue_logits<T, NC>(xraw, logits, V);
    SampleCfg c = c_imm;
    const T* noise = noise_imm;
    int frame = 0;
    if (st) {
        if (st->done) return;
        c.temperature = st->p_temperature; c.top_k = st->p_top_k; c.top_p = st->p_top_p; c.do_sample = st->p_do_sample;
        frame = st->frame;
        if (st->pred_noise)
            noise = gptr(reinterpret_cast<const T*>(st->pred_noise)) + ((size_t)(frame % st->noise_frames) * (G - 1) + cb) * V;
    }
    int tok;
    if constexpr (NUCLEUS) {
        __shared__ SampleSmem big;
        if (c.do_sample && c.top_p < 1.0f) tok = sample_nucleus_from_regs<T, NC>(xraw, V, c, nullptr, noise, big);
        else tok = sample_wave_core<T, NC>(xraw, V, c, nullptr, noise, sm);
    } else {
        tok = sample_wave_core<T, NC>(xraw, V, c, nullptr, noise, sm);
    }
    if (st) tok = forced_or(tf, frame * G + 1 + cb, tok);
    if (threadIdx.x == 0) {
        if (codes) codes[(size_t)frame * G + 1 + cb] = tok;
        if (out64) out64[cb] = tok;
    }
    if (next_emb) {
        const T* src = next_emb + (size_t)tok * H;
        for (int e = threadIdx.x * 8; e < H; e += 256 * 8) {
            Raw8<T> r; ldraw<false>(r, src + e);
            if (sizeof(T) == 2) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(next_in) + (size_t)e * 2) = *reinterpret_cast<u32x4*>(&r);
            else { float f[8]; unpack(r, f);
#pragma unroll
                   for (int i = 0; i < 8; ++i) DT<T>::st(next_in + e + i, f[i]); }
        }
    }
}

template <typename T, int NC>
__global__ __launch_bounds__(256) void sample_pred_wave_kernel(const DecodeState* st, const T* logits, int V, int cb,
                                                              SampleCfg c_imm, const T* noise_imm, int* codes, int G,
                                                              int64_t* out64, const T* next_emb, T* next_in, int H, const TeacherForcing* tf) {
    sample_pred_wave_body<T, NC>(st, logits, V, cb, c_imm, noise_imm, codes, G, out64, next_emb, next_in, H, tf);
}

template <typename T, int NC, bool NUCLEUS = false>
__device__ __forceinline__ void sample_talker_wave_body(DecodeState* st, const T* logits, int V, const unsigned char* seen, int G,
                                                        const TeacherForcing* tf = nullptr) {
    __shared__ BlkSmem sm;
    Raw8<T> xraw[NC];
    issue_logits<T, NC>(xraw, logits, V);
    if (st->done) return;
    SampleCfg c;
    c.temperature = st->t_temperature; c.top_k = st->t_top_k; c.top_p = st->t_top_p; c.do_sample = st->t_do_sample;
    c.rep_penalty = st->t_rep_penalty;
    c.sup_lo = st->sup_lo; c.sup_hi = st->sup_hi; c.keep_id = st->eos_id;
    const int frame = st->frame;
    c.sup_extra = (frame + 1 < st->min_new) ? st->eos_id : -1;
    const T* noise = st->talker_noise
        ? gptr(reinterpret_cast<const T*>(st->talker_noise)) + (size_t)(frame % st->noise_frames) * V : nullptr;
    int tok;
    if constexpr (NUCLEUS) {
        __shared__ SampleSmem big;
        if (c.do_sample && c.top_p < 1.0f) tok = sample_nucleus_from_regs<T, NC>(xraw, V, c, seen, noise, big);
        else tok = sample_wave_core<T, NC>(xraw, V, c, seen, noise, sm);
    } else {
        tok = sample_wave_core<T, NC>(xraw, V, c, seen, noise, sm);
    }
    tok = forced_or(tf, (frame + 1) * G, tok);
    if (threadIdx.x == 0) { st->token = tok; st->frame = frame + 1; st->pos += 1; st->gen_step += 1; }
}
template <typename T, int NC>
__global__ __launch_bounds__(256) void sample_talker_wave_kernel(DecodeState* st, const T* logits, int V,
                                                                const unsigned char* seen, int G, const TeacherForcing* tf) {
    sample_talker_wave_body<T, NC>(st, logits, V, seen, G, tf);
}

}  // namespace fq3
