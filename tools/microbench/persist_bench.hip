// Micro-benchmark (measurement tool, not product code): how fast can a chain of dependent M=1 GEMVs run on MI355X
//   V0  one kernel per GEMV, replayed from a hipGraph           (what the decode path does today)
//   V1  ONE persistent kernel, device-wide counter barrier between GEMVs, release/acquire fences + plain loads
//   V2  same, activations moved with agent-scope relaxed atomics (sc1), counter barrier without L2 wb/inv fences
//   V3  same as V2 with a flag-array barrier (every block publishes its stage, wave 0 of every block polls all flags)
// The chain has the code predictor's shapes: 5 layers x (4096<-1024, 1024<-2048, 6144<-1024, 1024<-3072), 16 passes.
// Every spin loop is bounded: a lost arrival aborts the run (error flag) instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

struct Sync {
    unsigned* counter;      // V1/V2: monotonically increasing arrival counter
    unsigned* flags;        // V3: flags[b] = last stage block b completed (+1)
    unsigned* abort_flag;   // set when a bounded spin gives up
    unsigned nblocks;
};

constexpr unsigned kSpinLimit = 1u << 18;

template <int MODE>
__device__ __forceinline__ void arrive(const Sync& s, unsigned stage) {
    __syncthreads();                                    // all waves of this block stored their rows
    if (threadIdx.x == 0) {
        if (MODE == 1) {
            __hip_atomic_fetch_add(s.counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 2) {
            __hip_atomic_fetch_add(s.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(s.flags + blockIdx.x, stage + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <int MODE>
__device__ __forceinline__ void wait_prev(const Sync& s, unsigned stage) {
    if (stage == 0) return;
    if (MODE == 3) {
        if (threadIdx.x < 64) {
            const unsigned lane = threadIdx.x;
            unsigned spins = 0;
            for (;;) {
                bool ok = true;
                for (unsigned i = lane; i < s.nblocks; i += 64)
                    ok &= __hip_atomic_load(s.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= stage;
                if (__all(ok)) break;
                if (++spins > kSpinLimit || __hip_atomic_load(s.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    if (lane == 0) __hip_atomic_store(s.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    } else if (threadIdx.x == 0) {
        const unsigned target = stage * s.nblocks;
        unsigned spins = 0;
        for (;;) {
            const unsigned v = MODE == 1 ? __hip_atomic_load(s.counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)
                                         : __hip_atomic_load(s.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (v >= target) break;
            if (++spins > kSpinLimit || __hip_atomic_load(s.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(s.abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

// one GEMV stage: y[N] = W[N,K] x[K]; every wave owns RW consecutive rows, every lane CH 16-byte chunks of a row
template <int RW, int CH, int MODE>
__device__ __forceinline__ void stage_body(const unsigned short* __restrict__ W, int K, const float* xin, float* xout,
                                           const Sync& s, unsigned stage) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row0 = (blockIdx.x * 4 + wave) * RW;
    u32x4 w[RW][CH];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int c = 0; c < CH; ++c)
            w[r][c] = __builtin_nontemporal_load((const u32x4*)(W + (size_t)(row0 + r) * K + (size_t)(lane + 64 * c) * 8));
    if (MODE != 0) wait_prev<MODE>(s, stage);
    float acc[RW];
#pragma unroll
    for (int r = 0; r < RW; ++r) acc[r] = 0.f;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        float x[8];
        const float* xp = xin + (size_t)(lane + 64 * c) * 8;
        if (MODE >= 2) {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __hip_atomic_load(xp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const f32x4 a = *(const f32x4*)xp, b = *(const f32x4*)(xp + 4);
            x[0] = a.x; x[1] = a.y; x[2] = a.z; x[3] = a.w; x[4] = b.x; x[5] = b.y; x[6] = b.z; x[7] = b.w;
        }
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const u32x4 q = w[r][c];
            acc[r] += bf_lo(q.x) * x[0] + bf_hi(q.x) * x[1] + bf_lo(q.y) * x[2] + bf_hi(q.y) * x[3]
                    + bf_lo(q.z) * x[4] + bf_hi(q.z) * x[5] + bf_lo(q.w) * x[6] + bf_hi(q.w) * x[7];
        }
    }
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        float v = acc[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) {
            if (MODE >= 2) __hip_atomic_store(xout + row0 + r, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else xout[row0 + r] = v;
        }
    }
    if (MODE != 0) arrive<MODE>(s, stage);
}

struct Chain {
    const unsigned short* W[20];   // 5 layers x 4 matrices
    float* buf[2];                 // ping-pong activation vectors (6144 floats each)
    int passes;
};

template <int MODE>
__global__ void __launch_bounds__(256) persistent_kernel(Chain ch, Sync s, unsigned stage_base) {
    unsigned stage = stage_base;
    int cur = 0;
    for (int p = 0; p < ch.passes; ++p)
        for (int l = 0; l < 5; ++l) {
            stage_body<4, 2, MODE>(ch.W[l * 4 + 0], 1024, ch.buf[cur], ch.buf[cur ^ 1], s, stage++); cur ^= 1;
            stage_body<1, 4, MODE>(ch.W[l * 4 + 1], 2048, ch.buf[cur], ch.buf[cur ^ 1], s, stage++); cur ^= 1;
            stage_body<6, 2, MODE>(ch.W[l * 4 + 2], 1024, ch.buf[cur], ch.buf[cur ^ 1], s, stage++); cur ^= 1;
            stage_body<1, 6, MODE>(ch.W[l * 4 + 3], 3072, ch.buf[cur], ch.buf[cur ^ 1], s, stage++); cur ^= 1;
        }
}

template <int RW, int CH>
__global__ void __launch_bounds__(256) single_kernel(const unsigned short* W, int K, const float* xin, float* xout) {
    Sync s{};
    stage_body<RW, CH, 0>(W, K, xin, xout, s, 0);
}

__global__ void empty_kernel(float* p) { if (p == nullptr) __builtin_trap(); }

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const int passes = argc > 1 ? atoi(argv[1]) : 16;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int G = 256;
    const int NS[4] = {4096, 1024, 6144, 1024}, KS[4] = {1024, 2048, 1024, 3072};
    Chain ch{};
    ch.passes = passes;
    srand(1);
    for (int l = 0; l < 5; ++l)
        for (int m = 0; m < 4; ++m) {
            const size_t n = (size_t)NS[m] * KS[m];
            std::vector<unsigned short> h(n);
            const float sc = 1.7f / sqrtf((float)KS[m]);          // keeps the chained vector's norm roughly constant
            for (size_t i = 0; i < n; ++i) h[i] = f2bf(sc * ((rand() & 0xffff) / 32768.f - 1.f));
            void* d; CHK(hipMalloc(&d, n * 2)); CHK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
            ch.W[l * 4 + m] = (const unsigned short*)d;
        }
    std::vector<float> x0(6144);
    for (auto& v : x0) v = (rand() & 0xffff) / 32768.f - 1.f;
    for (int i = 0; i < 2; ++i) CHK(hipMalloc(&ch.buf[i], 6144 * 4));
    Sync s{};
    CHK(hipMalloc(&s.counter, 256)); CHK(hipMalloc(&s.flags, G * 4)); CHK(hipMalloc(&s.abort_flag, 256));
    s.nblocks = G;
    hipStream_t st; CHK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    const int stages = passes * 20;
    std::vector<float> ref(1024), out(1024);
    auto reset = [&]() {
        CHK(hipMemcpyAsync(ch.buf[0], x0.data(), 6144 * 4, hipMemcpyHostToDevice, st));
        CHK(hipMemsetAsync(ch.buf[1], 0, 6144 * 4, st));
        CHK(hipMemsetAsync(s.counter, 0, 256, st)); CHK(hipMemsetAsync(s.flags, 0, G * 4, st)); CHK(hipMemsetAsync(s.abort_flag, 0, 256, st));
        CHK(hipStreamSynchronize(st));
    };
    // ---- V0: graph of single kernels -------------------------------------------------------------
    reset();
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    {
        int cur = 0;
        for (int p = 0; p < passes; ++p)
            for (int l = 0; l < 5; ++l) {
                hipLaunchKernelGGL((single_kernel<4, 2>), dim3(G), dim3(256), 0, st, ch.W[l * 4 + 0], 1024, ch.buf[cur], ch.buf[cur ^ 1]); cur ^= 1;
                hipLaunchKernelGGL((single_kernel<1, 4>), dim3(G), dim3(256), 0, st, ch.W[l * 4 + 1], 2048, ch.buf[cur], ch.buf[cur ^ 1]); cur ^= 1;
                hipLaunchKernelGGL((single_kernel<6, 2>), dim3(G), dim3(256), 0, st, ch.W[l * 4 + 2], 1024, ch.buf[cur], ch.buf[cur ^ 1]); cur ^= 1;
                hipLaunchKernelGGL((single_kernel<1, 6>), dim3(G), dim3(256), 0, st, ch.W[l * 4 + 3], 3072, ch.buf[cur], ch.buf[cur ^ 1]); cur ^= 1;
            }
    }
    CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
    CHK(hipMemcpy(ref.data(), ch.buf[0], 1024 * 4, hipMemcpyDeviceToHost));     // even number of stages -> result in buf[0]
    {
        float ms = 0;
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CHK(hipGraphLaunch(ge, st));
        CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        double nrm = 0; for (float v : ref) nrm += (double)v * v;
        printf("V0 graph of %d kernels      : %8.3f us / stage   (%.3f ms / %d-pass chain)   |y|=%.4g\n", stages, 1e3 * ms / reps / stages, ms / reps, passes, sqrt(nrm));
    }
    // empty dependent kernels for reference
    {
        hipGraph_t g2; hipGraphExec_t ge2;
        CHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < stages; ++i) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, st, ch.buf[0]);
        CHK(hipStreamEndCapture(st, &g2)); CHK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
        CHK(hipGraphLaunch(ge2, st)); CHK(hipStreamSynchronize(st));
        float ms = 0;
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CHK(hipGraphLaunch(ge2, st));
        CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        printf("   graph of %d EMPTY kernels: %8.3f us / node\n", stages, 1e3 * ms / reps / stages);
    }
    // ---- V1..V3: persistent ------------------------------------------------------------------------------
    for (int mode = 1; mode <= 3; ++mode) {
        auto launch = [&](unsigned base) {
            if (mode == 1) hipLaunchKernelGGL((persistent_kernel<1>), dim3(G), dim3(256), 0, st, ch, s, base);
            if (mode == 2) hipLaunchKernelGGL((persistent_kernel<2>), dim3(G), dim3(256), 0, st, ch, s, base);
            if (mode == 3) hipLaunchKernelGGL((persistent_kernel<3>), dim3(G), dim3(256), 0, st, ch, s, base);
        };
        reset();
        launch(0);
        CHK(hipStreamSynchronize(st));
        unsigned ab = 0; CHK(hipMemcpy(&ab, s.abort_flag, 4, hipMemcpyDeviceToHost));
        CHK(hipMemcpy(out.data(), ch.buf[0], 1024 * 4, hipMemcpyDeviceToHost));
        int bad = 0; for (int i = 0; i < 1024; ++i) bad += memcmp(&out[i], &ref[i], 4) != 0;
        if (ab) { printf("V%d persistent: ABORTED (bounded spin expired) -- skipping timing\n", mode); continue; }
        float ms = 0;
        CHK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) launch((unsigned)(r + 1) * stages);       // counters keep counting up
        CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(&ab, s.abort_flag, 4, hipMemcpyDeviceToHost));
        const char* nm[4] = {"", "counter barrier, acq/rel fences ", "counter barrier, sc1 activations ", "flag-array barrier, sc1 activ.   "};
        printf("V%d %s: %8.3f us / stage   (%.3f ms / chain)   mismatching outputs vs V0: %d%s\n", mode, nm[mode],
               1e3 * ms / reps / stages, ms / reps, bad, ab ? "   [ABORTED during timing]" : "");
    }
    return 0;
}
