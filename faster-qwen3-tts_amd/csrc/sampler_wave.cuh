// Single-wavefront sampler (default path, top_p >= 1): the whole vocabulary row lives in the
// registers of ONE wave64 (8 contiguous ids per lane per 512-id chunk), so the only synchronisation
// is wave-level: no workgroup barriers, one global round trip for logits, noise lands behind the
// radix select.  Same arithmetic and rounding as sample_core (sampler.cuh) / sampling.py:32-66.
#pragma once
#include "sampler.cuh"

namespace fq3 {

__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b; b.v = __shfl_xor(a.v, o, 64); b.i = __shfl_xor(a.i, o, 64);
        a = amax2(a, b);
    }
    return a;
}

// logits already issued into xraw by the caller (so that state loads overlap); returns token (all lanes)
template <typename T, int NC>
__device__ int sample_wave_core(Raw8<T> (&xraw)[NC], int V, const SampleCfg& c, const unsigned char* seen,
                                const T* noise) {
    const int lane = threadIdx.x & 63;
    float x[NC][8];
    Raw8<T> nz[NC];
    const bool samp = c.do_sample != 0;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int off = j * 512 + lane * 8;
        if (samp && off < V) ldraw<false>(nz[j], noise + off); else zero(nz[j]);
    }
    const bool pen = seen && c.rep_penalty != 1.0f;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int off = j * 512 + lane * 8;
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
            for (int i = 0; i < 8; ++i) { ArgMax b; b.v = x[j][i]; b.i = j * 512 + lane * 8 + i; a = amax2(a, b); }
        return wave_argmax(a).i;
    }
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[j][i] = DT<T>::rnd(x[j][i] / c.temperature);
    if (c.top_k > 0) {
        // k-th largest key by MSB-first bitwise search: count(key >= candidate) over the wave.  (An LDS
        // histogram serialises badly here: the top byte of the keys is nearly constant.)
        const int kk = min(c.top_k, V);
        uint32_t key[NC][8];
#pragma unroll
        for (int j = 0; j < NC; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) key[j][i] = (j * 512 + lane * 8 + i) < V ? okey(x[j][i]) : 0u;
        uint32_t prefix = 0;
        constexpr int kLowBit = sizeof(T) == 2 ? 16 : 0;       // bf16 values live in the top 16 bits
        for (int bit = 31; bit >= kLowBit; --bit) {
            const uint32_t cand = prefix | (1u << bit);
            int cnt = 0;                 // wave-uniform: ballots + scalar popcounts, no cross-lane shuffles
#pragma unroll
            for (int j = 0; j < NC; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) cnt += __popcll(__ballot(key[j][i] >= cand));
            if (cnt >= kk) prefix = cand;
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
    mx = wave_max(mx);
    float se = 0.f;
#pragma unroll
    for (int j = 0; j < NC; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[j][i] = expf(x[j][i] - mx); se += x[j][i]; }
    se = wave_sum(se);
    ArgMax a; a.v = -INFINITY; a.i = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        float q[8];
        unpack(nz[j], q);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int id = j * 512 + lane * 8 + i;
            const float p = DT<T>::rnd(x[j][i] / se);
            ArgMax b; b.v = id < V ? DT<T>::rnd(p / q[i]) : -INFINITY; b.i = id;
            a = amax2(a, b);
        }
    }
    return wave_argmax(a).i;
}

template <typename T, int NC>
__device__ __forceinline__ void issue_logits(Raw8<T> (&xraw)[NC], const T* logits, int V) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int off = j * 512 + lane * 8;
        if (off < V) ldraw<false>(xraw[j], logits + off); else zero(xraw[j]);
    }
}

template <typename T, int NC>
__global__ __launch_bounds__(64) void sample_api_wave_kernel(const T* logits, int V, SampleCfg c, const unsigned char* seen,
                                                             const T* noise, int64_t* out) {
    Raw8<T> xraw[NC];
    issue_logits<T, NC>(xraw, logits, V);
    const int tok = sample_wave_core<T, NC>(xraw, V, c, seen, noise);
    if (threadIdx.x == 0) out[0] = tok;
}

template <typename T, int NC>
__global__ __launch_bounds__(64) void sample_pred_wave_kernel(const DecodeState* st, const T* logits, int V, int cb,
                                                              SampleCfg c_imm, const T* noise_imm, int* codes, int G,
                                                              int64_t* out64, const T* next_emb, T* next_in, int H) {
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
            noise = reinterpret_cast<const T*>(st->pred_noise) + ((size_t)(frame % st->noise_frames) * (G - 1) + cb) * V;
    }
    const int tok = sample_wave_core<T, NC>(xraw, V, c, nullptr, noise);
    if (threadIdx.x == 0) {
        if (codes) codes[(size_t)frame * G + 1 + cb] = tok;
        if (out64) out64[cb] = tok;
    }
    if (next_emb) {
        const T* src = next_emb + (size_t)tok * H;
        for (int e = threadIdx.x * 8; e < H; e += 64 * 8) {
            Raw8<T> r; ldraw<false>(r, src + e);
            if (sizeof(T) == 2) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(next_in) + (size_t)e * 2) = *reinterpret_cast<u32x4*>(&r);
            else { float f[8]; unpack(r, f);
#pragma unroll
                   for (int i = 0; i < 8; ++i) DT<T>::st(next_in + e + i, f[i]); }
        }
    }
}

template <typename T, int NC>
__global__ __launch_bounds__(64) void sample_talker_wave_kernel(DecodeState* st, const T* logits, int V,
                                                                const unsigned char* seen) {
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
        ? reinterpret_cast<const T*>(st->talker_noise) + (size_t)(frame % st->noise_frames) * V : nullptr;
    const int tok = sample_wave_core<T, NC>(xraw, V, c, seen, noise);
    if (threadIdx.x == 0) { st->token = tok; st->frame = frame + 1; st->pos += 1; st->gen_step += 1; }
}

}  // namespace fq3
