// Measurement tool (not product code): WHERE does a weight-stationary GEMM launch of the lock-step batch spend its 6-9 us?
// skinny_gemm_kernel is compiled here with FQ3_SK_TRACE: every wave records core-clock stamps at entry / first token unit landed / first
// MFMAs issued (weights landed) / first group multiplied / barrier passed / first epilogues done / exit, plus the chip-wide 100 MHz wall
// clock at entry and exit.  The tool replays a hipGraph of NLAY layers (norm + qkv, o_proj, norm + gate|up, down at the 0.6B or 1.7B
// shapes, B token rows) and prints, for the four GEMMs of the LAST layer: when the first / last workgroup started and ended relative to
// the first GEMM's first wave, the gap to the next launch, and the median / max of every phase stamp in microseconds.
// usage: skinny_trace [B=128] [size=0|1 (0.6B | 1.7B)] [unused] [reps=20] [packed=0|1: fragment-major weight copies]
#define FQ3_SK_TRACE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include "../../faster-qwen3-tts_amd/csrc/batch_kernels.cuh"
#include "../../faster-qwen3-tts_amd/csrc/skinny_gemm.cuh"
using namespace fq3;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static void* dev_bf16(size_t n, float scale, float offset = 0.f) {
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(offset + scale * ((rand() & 0xffff) / 32768.f - 1.f));
    void* d; CHK(hipMalloc(&d, n * 2)); CHK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    const int size = argc > 2 ? atoi(argv[2]) : 0;
    const int depth = argc > 3 ? atoi(argv[3]) : 2;
    const int reps = argc > 4 ? atoi(argv[4]) : 20;
    const int packed = argc > 5 ? atoi(argv[5]) : 0;
    const int H = size ? 2048 : 1024, I = size ? 6144 : 3072, QD = 2048, NQKV = 4096;
    hipStream_t st; CHK(hipStreamCreate(&st));
    if (!(skinny_prepare<SK_STORE>() && skinny_prepare<SK_SWIGLU>() && skinny_prepare<SK_RESIDUAL>())) { fprintf(stderr, "LDS limit\n"); return 2; }
    srand(5);
    const int NL = 5, NLAY = 20;
    void *Wqkv[NL], *Wo[NL], *Wgu[NL], *Wdn[NL];
    for (int l = 0; l < NL; ++l) {
        Wqkv[l] = dev_bf16((size_t)NQKV * H, 1.f / sqrtf((float)H)); Wo[l] = dev_bf16((size_t)H * QD, 1.f / sqrtf((float)QD));
        Wgu[l] = dev_bf16((size_t)2 * I * H, 1.f / sqrtf((float)H)); Wdn[l] = dev_bf16((size_t)H * I, 1.f / sqrtf((float)I));
    }
    void *Pqkv[NL], *Po[NL], *Pgu[NL], *Pdn[NL];
    for (int l = 0; l < NL; ++l) {
        CHK(hipMalloc(&Pqkv[l], (size_t)NQKV * H * 2)); CHK(hipMalloc(&Po[l], (size_t)H * QD * 2)); CHK(hipMalloc(&Pgu[l], (size_t)2 * I * H * 2)); CHK(hipMalloc(&Pdn[l], (size_t)H * I * 2));
        skinny_pack((const bf16_t*)Wqkv[l], (bf16_t*)Pqkv[l], NQKV, H, 0, st); skinny_pack((const bf16_t*)Wo[l], (bf16_t*)Po[l], H, QD, 0, st);
        skinny_pack((const bf16_t*)Wgu[l], (bf16_t*)Pgu[l], 2 * I, H, I, st); skinny_pack((const bf16_t*)Wdn[l], (bf16_t*)Pdn[l], H, I, 0, st);
    }
    CHK(hipStreamSynchronize(st));
    void* gain = dev_bf16(H, 0.05f, 1.f);
    void* attn = dev_bf16((size_t)B * QD, 1.f); void* h = dev_bf16((size_t)B * H, 1.f); void* xn = dev_bf16((size_t)B * H, 1.f);
    void* qkv = dev_bf16((size_t)B * NQKV, 1.f); void* act = dev_bf16((size_t)B * I, 1.f);
    const size_t TR = (size_t)512 * 8 * 8;                     // up to 512 workgroups
    unsigned long long* tr[4];
    for (int i = 0; i < 4; ++i) { CHK(hipMalloc(&tr[i], TR * 8)); CHK(hipMemset(tr[i], 0, TR * 8)); }
    int grid[4] = {0, 0, 0, 0};
    auto norm = [&]() {
        const dim3 g((B + 3) / 4);
        if (H <= 1024) hipLaunchKernelGGL((rmsnorm_batch_kernel<2>), g, dim3(256), 0, st, (const bf16_t*)h, H, (const bf16_t*)gain, 1e-6f, H, B, (bf16_t*)xn, H, (void* const*)nullptr);
        else hipLaunchKernelGGL((rmsnorm_batch_kernel<4>), g, dim3(256), 0, st, (const bf16_t*)h, H, (const bf16_t*)gain, 1e-6f, H, B, (bf16_t*)xn, H, (void* const*)nullptr);
    };
    auto layer = [&](int l) {
        SkinnyArgs k{};
        norm();
        k = SkinnyArgs{}; k.Wp = packed ? (const bf16_t*)Pqkv[l] : nullptr; k.trace = tr[0]; k.X = (const bf16_t*)xn; k.ldx = H; k.M = B; k.W = (const bf16_t*)Wqkv[l]; k.N = NQKV; k.Y = (bf16_t*)qkv; k.ldy = NQKV;
        skinny_launch<SK_STORE>(k, H, st);
        k = SkinnyArgs{}; k.Wp = packed ? (const bf16_t*)Po[l] : nullptr; k.trace = tr[1]; k.X = (const bf16_t*)attn; k.ldx = QD; k.M = B; k.W = (const bf16_t*)Wo[l]; k.N = H; k.res = (const bf16_t*)h; k.ldr = H; k.Y = (bf16_t*)h; k.ldy = H;
        skinny_launch<SK_RESIDUAL>(k, QD, st);
        norm();
        k = SkinnyArgs{}; k.Wp = packed ? (const bf16_t*)Pgu[l] : nullptr; k.trace = tr[2]; k.X = (const bf16_t*)xn; k.ldx = H; k.M = B; k.W = (const bf16_t*)Wgu[l]; k.N = 2 * I; k.Y = (bf16_t*)act; k.ldy = I;
        skinny_launch<SK_SWIGLU>(k, H, st);
        k = SkinnyArgs{}; k.Wp = packed ? (const bf16_t*)Pdn[l] : nullptr; k.trace = tr[3]; k.X = (const bf16_t*)act; k.ldx = I; k.M = B; k.W = (const bf16_t*)Wdn[l]; k.N = H; k.res = (const bf16_t*)h; k.ldr = H; k.Y = (bf16_t*)h; k.ldy = H;
        skinny_launch<SK_RESIDUAL>(k, I, st);
    };
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < NLAY; ++i) layer(i % NL);
    CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) CHK(hipGraphLaunch(ge, st));
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    printf("B=%d %s %s weights: %.3f us per layer (6 launches, traced build)\n", B, size ? "1.7B" : "0.6B", packed ? "fragment-major" : "row-major", 1e3 * ms / reps / NLAY);
    // the launch geometry: recompute what skinny_launch picked (non-zero trace rows)
    const char* names[4] = {"qkv   ", "o_proj", "gate|up", "down  "};
    std::vector<unsigned long long> t[4];
    unsigned long long base = ~0ull;
    for (int i = 0; i < 4; ++i) {
        t[i].resize(TR); CHK(hipMemcpy(t[i].data(), tr[i], TR * 8, hipMemcpyDeviceToHost));
        for (size_t w = 0; w < TR / 8; ++w) if (t[i][w * 8 + 7]) { grid[i] = (int)(w / 8) + 1; if (i == 0) base = std::min(base, t[i][w * 8]); }
    }
    double prev_end = 0;
    for (int i = 0; i < 4; ++i) {
        std::vector<double> start, end, ph[7], hz;
        for (size_t w = 0; w < (size_t)grid[i] * 8; ++w) {
            const unsigned long long* r = &t[i][w * 8];
            if (!r[7]) continue;
            start.push_back((double)(long long)(r[0] - base) * 0.01); end.push_back((double)(long long)(r[7] - base) * 0.01);
            const double wall_us = (double)(r[7] - r[0]) * 0.01;
            if (wall_us > 0.5) hz.push_back((double)r[6] / wall_us);           // core cycles per microsecond
            for (int p = 1; p < 7; ++p) ph[p].push_back((double)r[p]);
        }
        if (start.empty()) { printf("%s: no trace\n", names[i]); continue; }
        auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mx = [](const std::vector<double>& v) { return *std::max_element(v.begin(), v.end()); };
        auto mn = [](const std::vector<double>& v) { return *std::min_element(v.begin(), v.end()); };
        const double f = hz.empty() ? 2400.0 : med(hz);
        printf("%s %3d workgroups  clock %.0f MHz | first wave starts %+7.2f us (gap after previous kernel's last exit: %5.2f)  last wave starts %+7.2f  first exit %+7.2f  last exit %+7.2f  => span %.2f us\n",
               names[i], grid[i], f, mn(start), i ? mn(start) - prev_end : 0.0, mx(start), mn(end), mx(end), mx(end) - mn(start));
        const char* pn[7] = {"", "unit 0 landed", "first MFMAs issued", "group 0 multiplied", "barrier passed", "epilogues of group 0", "exit"};
        for (int p = 1; p < 7; ++p) printf("      %-22s median %6.2f us   max %6.2f us   (since the wave's entry)\n", pn[p], med(ph[p]) / f, mx(ph[p]) / f);
        prev_end = mx(end);
    }
    return 0;
}
