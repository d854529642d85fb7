// On-device sampler: reference faster_qwen3_tts/sampling.py:10-66 as ONE workgroup.
//   repetition penalty (history bitmap instead of unique(), sampling.py:24) -> suppress mask ->
//   greedy argmax | /T -> top-k threshold with ties kept (sampling.py:54-56, radix select instead of
//   topk) -> optional top-p (stable sort, cumulative cut, first kept; sampling.py:57-65) ->
//   softmax -> argmax(p / q) with host-drawn Exp(1) noise q  ==  torch.multinomial(p, 1).
// All intermediate values are rounded through T exactly where the Torch ops round.
#pragma once
#include "fq3_common.cuh"
#include "decode_kernels.cuh"

namespace fq3 {

constexpr int kMaxVocab = 4096;

__device__ __forceinline__ uint32_t okey(float f) {           // order-preserving float -> uint
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct ArgMax { float v; int i; };
__device__ __forceinline__ ArgMax amax2(ArgMax a, ArgMax b) {
    // larger value wins; ties -> lower index (torch.argmax returns the first maximal index)
    return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}
__device__ __forceinline__ int block_argmax(ArgMax a, float* redv, int* redi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        ArgMax b; b.v = __shfl_xor(a.v, o, 64); b.i = __shfl_xor(a.i, o, 64);
        a = amax2(a, b);
    }
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { redv[w] = a.v; redi[w] = a.i; }
    __syncthreads();
    ArgMax r; r.v = redv[0]; r.i = redi[0];
    for (int i = 1; i < 4; ++i) { ArgMax b; b.v = redv[i]; b.i = redi[i]; r = amax2(r, b); }
    return r.i;
}

struct SampleCfg {
    float temperature; int top_k; float top_p; int do_sample; float rep_penalty;
    int sup_lo, sup_hi, keep_id, sup_extra;      // suppress [sup_lo, sup_hi) except keep_id; plus sup_extra (>=0)
};

// Shared-memory image used by the sampler (static: 16 KB vals + 32 KB sort keys + small)
struct SampleSmem {
    float vals[kMaxVocab];
    unsigned long long keys[kMaxVocab];
    int hist[256];
    float redv[8]; int redi[8];
    int sel_bin, sel_k;
};

// vals[] must already hold the fp32 image of the T logits.  Returns the sampled id (all threads).
template <typename T>
__device__ int sample_core(SampleSmem& sm, int V, const SampleCfg& c, const unsigned char* seen, const T* noise) {
    const int tid = threadIdx.x;
    // 1. repetition penalty on seen ids, then suppression
    for (int i = tid; i < V; i += 256) {
        float x = sm.vals[i];
        if (seen && c.rep_penalty != 1.0f && seen[i])
            x = x > 0.f ? DT<T>::rnd(x / c.rep_penalty) : DT<T>::rnd(x * c.rep_penalty);
        if ((i >= c.sup_lo && i < c.sup_hi && i != c.keep_id) || i == c.sup_extra) x = -INFINITY;
        sm.vals[i] = x;
    }
    __syncthreads();
    if (!c.do_sample) {
        ArgMax a; a.v = -INFINITY; a.i = 0x7fffffff;
        for (int i = tid; i < V; i += 256) { ArgMax b; b.v = sm.vals[i]; b.i = i; a = amax2(a, b); }
        return block_argmax(a, sm.redv, sm.redi);
    }
    // 2. temperature
    for (int i = tid; i < V; i += 256) sm.vals[i] = DT<T>::rnd(sm.vals[i] / c.temperature);
    __syncthreads();
    // 3. top-k: k-th largest value by MSB-first radix select on order-preserving keys
    if (c.top_k > 0) {
        int kk = min(c.top_k, V);
        uint32_t prefix = 0, mask = 0;
        constexpr int kPasses = sizeof(T) == 2 ? 2 : 4;      // bf16 values live in the top 16 bits
        for (int pass = 0; pass < kPasses; ++pass) {
            const int shift = 24 - 8 * pass;
            sm.hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < V; i += 256) {
                const uint32_t k = okey(sm.vals[i]);
                if ((k & mask) == prefix) atomicAdd(&sm.hist[(k >> shift) & 255], 1);
            }
            __syncthreads();
            // suffix count: number of candidates in bins >= tid
            int cnt = sm.hist[tid];
            int incl = cnt;                       // inclusive suffix sum over bins [tid, 255]
            {
                // wave-level suffix scan then cross-wave fixup
                const int lane = tid & 63, w = tid >> 6;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_down(incl, o, 64);
                    if (lane + o < 64) incl += t;
                }
                __syncthreads();
                if (lane == 0) sm.redi[w] = incl;      // total of wave w
                __syncthreads();
                for (int ww = w + 1; ww < 4; ++ww) incl += sm.redi[ww];
            }
            const int excl = incl - cnt;          // candidates in bins > tid
            if (excl < kk && kk <= incl) { sm.sel_bin = tid; sm.sel_k = kk - excl; }
            __syncthreads();
            prefix |= ((uint32_t)sm.sel_bin) << shift;
            mask |= 0xFFu << shift;
            kk = sm.sel_k;
            __syncthreads();
        }
        const uint32_t kth = prefix;              // key of the k-th largest value (low bits zero for bf16)
        for (int i = tid; i < V; i += 256)
            if ((okey(sm.vals[i]) & mask) < kth) sm.vals[i] = -INFINITY;
        __syncthreads();
    }
    // 4. top-p (nucleus), only when requested: stable descending sort, cumulative cut
    if (c.top_p < 1.0f) {
        int n = 1;
        while (n < V) n <<= 1;
        for (int i = tid; i < n; i += 256)
            sm.keys[i] = i < V ? (((unsigned long long)(~okey(sm.vals[i]))) << 32) | (unsigned)i : ~0ull;
        __syncthreads();
        for (int k = 2; k <= n; k <<= 1)
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < n; i += 256) {
                    const int p = i ^ j;
                    if (p > i) {
                        const unsigned long long A = sm.keys[i], B = sm.keys[p];
                        const bool up = (i & k) == 0;
                        if ((A > B) == up) { sm.keys[i] = B; sm.keys[p] = A; }
                    }
                }
                __syncthreads();
            }
        float mx = -INFINITY;
        for (int i = tid; i < V; i += 256) mx = fmaxf(mx, sm.vals[i]);
        mx = block_max<4>(mx, sm.redv);
        float se = 0.f;
        for (int i = tid; i < V; i += 256) se += expf(sm.vals[i] - mx);
        se = block_sum<4>(se, sm.redv);
        if (tid == 0) {
            const float thr = DT<T>::rnd(c.top_p);
            float cum = 0.f;
            for (int j = 0; j < V; ++j) {
                const int idx = (int)(sm.keys[j] & 0xffffffffu);
                const float x = sm.vals[idx];
                if (x == -INFINITY) break;
                const float p = DT<T>::rnd(expf(x - mx) / se);
                cum += p;
                if (j > 0 && DT<T>::rnd(cum) > thr) sm.vals[idx] = -INFINITY;
            }
        }
        __syncthreads();
    }
    // 5. softmax + exponential-race draw
    float mx = -INFINITY;
    for (int i = tid; i < V; i += 256) mx = fmaxf(mx, sm.vals[i]);
    mx = block_max<4>(mx, sm.redv);
    float se = 0.f;
    for (int i = tid; i < V; i += 256) se += expf(sm.vals[i] - mx);
    se = block_sum<4>(se, sm.redv);
    ArgMax a; a.v = -INFINITY; a.i = 0x7fffffff;
    for (int i = tid; i < V; i += 256) {
        const float p = DT<T>::rnd(expf(sm.vals[i] - mx) / se);
        ArgMax b; b.v = DT<T>::rnd(p / DT<T>::ld(noise + i)); b.i = i;
        a = amax2(a, b);
    }
    return block_argmax(a, sm.redv, sm.redi);
}

// ---- standalone entry (fq3_sample): explicit history list ---------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sample_api_kernel(const T* logits, int V, SampleCfg c, const int64_t* history,
                                                         int n_hist, const T* noise, int64_t* out) {
    __shared__ SampleSmem sm;
    __shared__ unsigned char seen_l[kMaxVocab];
    for (int i = threadIdx.x; i < V; i += 256) { sm.vals[i] = DT<T>::ld(logits + i); seen_l[i] = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < n_hist; i += 256) { const int id = (int)history[i]; if (id >= 0 && id < V) seen_l[id] = 1; }
    __syncthreads();
    const int tok = sample_core<T>(sm, V, c, n_hist > 0 ? seen_l : nullptr, noise);
    if (threadIdx.x == 0) out[0] = tok;
}

// ---- predictor sampler inside the loop / predictor_loop API (predictor_graph.py:131-139,158-165) --
// Writes the id, and stages the NEXT pass's input row codec_embeds[cb][tok] (predictor_graph.py:144).
template <typename T>
__global__ __launch_bounds__(256) void sample_pred_kernel(const DecodeState* st, const T* logits, int V, int cb,
                                                          SampleCfg c_imm, const T* noise_imm, int* codes, int G,
                                                          int64_t* out64, const T* next_emb, T* next_in, int H, const TeacherForcing* tf) {
    if (st && st->done) return;
    __shared__ SampleSmem sm;
    __shared__ int s_tok;
    SampleCfg c = c_imm;
    const T* noise = noise_imm;
    int frame = 0;
    if (st) {
        c.temperature = st->p_temperature; c.top_k = st->p_top_k; c.top_p = st->p_top_p; c.do_sample = st->p_do_sample;
        frame = st->frame;
        if (st->pred_noise)
            noise = gptr(reinterpret_cast<const T*>(st->pred_noise)) + ((size_t)(frame % st->noise_frames) * (G - 1) + cb) * V;
    }
    for (int i = threadIdx.x; i < V; i += 256) sm.vals[i] = DT<T>::ld(logits + i);
    __syncthreads();
    int tok = sample_core<T>(sm, V, c, nullptr, noise);
    if (st) tok = forced_or(tf, frame * G + 1 + cb, tok);
    if (threadIdx.x == 0) {
        if (codes) codes[(size_t)frame * G + 1 + cb] = tok;
        if (out64) out64[cb] = tok;
    }
    if (next_emb)
        for (int e = threadIdx.x; e < H; e += 256) next_in[e] = next_emb[(size_t)tok * H + e];
}

// ---- talker sampler at the end of a frame (generate.py:184-199) -----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void sample_talker_kernel(DecodeState* st, const T* logits, int V,
                                                            const unsigned char* seen, int G, const TeacherForcing* tf) {
    if (st->done) return;
    __shared__ SampleSmem sm;
    SampleCfg c;
    c.temperature = st->t_temperature; c.top_k = st->t_top_k; c.top_p = st->t_top_p; c.do_sample = st->t_do_sample;
    c.rep_penalty = st->t_rep_penalty;
    c.sup_lo = st->sup_lo; c.sup_hi = st->sup_hi; c.keep_id = st->eos_id;
    const int frame = st->frame;
    c.sup_extra = (frame + 1 < st->min_new) ? st->eos_id : -1;      // len(all_codec_ids) < min_new_tokens
    const T* noise = st->talker_noise
        ? gptr(reinterpret_cast<const T*>(st->talker_noise)) + (size_t)(frame % st->noise_frames) * V : nullptr;
    for (int i = threadIdx.x; i < V; i += 256) sm.vals[i] = DT<T>::ld(logits + i);
    __syncthreads();
    int tok = sample_core<T>(sm, V, c, seen, noise);
    __syncthreads();
    tok = forced_or(tf, (frame + 1) * G, tok);
    if (threadIdx.x == 0) {
        st->token = tok; st->frame = frame + 1; st->pos += 1; st->gen_step += 1;
    }
}

}  // namespace fq3
