// Measurement + self-check tool (not product code): the RMSNorm folded into the weight-stationary GEMM pair of the lock-step batch
// (csrc/skinny_gemm.cuh, round 5) against the round-4 form (rmsnorm_batch_kernel + the same GEMM on pre-normalised rows).
//   1. checks, B = 48 / 128 / 256 rows, K = hidden 1024 and 2048:
//        * the sum-of-squares partials a residual GEMM leaves (SkinnyArgs::ssq_out) against a CPU sum over the rows it stored;
//        * the normalising GEMM (SkinnyArgs::ssq) against the two-launch form on the same rows (they differ only in the ORDER of the
//          sum of squares: a handful of 1-ulp flips of normalised values) and against a CPU double-precision reference;
//   2. hipGraph chains of one layer's four GEMMs (qkv, o_proj, gate | up, down) in both forms at the 0.6B and 1.7B shapes, weights
//      rotating over 5 copies: microseconds per layer.
// usage: normfuse_bench [reps]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <functional>
#include "../../faster-qwen3-tts_amd/csrc/batch_kernels.cuh"
#include "../../faster-qwen3-tts_amd/csrc/skinny_gemm.cuh"
using namespace fq3;

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short v) { unsigned u = (unsigned)v << 16; float f; memcpy(&f, &u, 4); return f; }
static float rbf(float f) { return bf2f(f2bf(f)); }
static void* dev_bf16(size_t n, float scale, float offset = 0.f) {
    std::vector<unsigned short> h(n);
    for (size_t i = 0; i < n; ++i) h[i] = f2bf(offset + scale * ((rand() & 0xffff) / 32768.f - 1.f));
    void* d; CHK(hipMalloc(&d, n * 2)); CHK(hipMemcpy(d, h.data(), n * 2, hipMemcpyHostToDevice));
    return d;
}
static std::vector<float> fetch_bf16(const void* d, size_t n) {
    std::vector<unsigned short> h(n); CHK(hipMemcpy(h.data(), d, n * 2, hipMemcpyDeviceToHost));
    std::vector<float> f(n); for (size_t i = 0; i < n; ++i) f[i] = bf2f(h[i]); return f;
}
static std::vector<float> fetch_f32(const void* d, size_t n) { std::vector<float> f(n); CHK(hipMemcpy(f.data(), d, n * 4, hipMemcpyDeviceToHost)); return f; }

static hipStream_t st;
static hipEvent_t e0, e1;
static int g_reps = 20, g_fail = 0;
static int g_packed = 0;       // 1 = the launches below read the fragment-major weight copies (SkinnyArgs::Wp, round 6)
static void report(const char* what, double err, double tol) {
    printf("check %-72s %.3e (tol %.1e) %s\n", what, err, tol, err <= tol ? "ok" : "FAIL");
    if (!(err <= tol)) ++g_fail;
}
static double chain(const char* name, int n, int per, const std::function<void(int)>& launch) {
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < n; ++i) launch(i);
    CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CHK(hipGraphLaunch(ge, st));
    CHK(hipStreamSynchronize(st));
    float ms = 0;
    CHK(hipEventRecord(e0, st));
    for (int r = 0; r < g_reps; ++r) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms, e0, e1));
    CHK(hipGetLastError());
    const double us = 1e3 * ms / g_reps / n * per;
    printf("%-86s %8.3f us per layer\n", name, us);
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
    return us;
}

static void rmsnorm_rows(const void* x, int ldx, const void* gain, int K, int B, void* y) {
    const dim3 grid((B + 3) / 4);
    if (K <= 1024) hipLaunchKernelGGL((rmsnorm_batch_kernel<2>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)gain, 1e-6f, K, B, (bf16_t*)y, K, (void* const*)nullptr);
    else hipLaunchKernelGGL((rmsnorm_batch_kernel<4>), grid, dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)gain, 1e-6f, K, B, (bf16_t*)y, K, (void* const*)nullptr);
}

struct Shapes { const char* name; int H, I, QD, NQKV; };

int main(int argc, char** argv) {
    g_reps = argc > 1 ? atoi(argv[1]) : 20;
    srand(11);
    CHK(hipStreamCreate(&st)); CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    if (!(skinny_prepare<SK_STORE>() && skinny_prepare<SK_SWIGLU>() && skinny_prepare<SK_RESIDUAL>())) { fprintf(stderr, "LDS limit\n"); return 2; }
    const Shapes shapes[2] = {{"0.6B", 1024, 3072, 2048, 4096}, {"1.7B", 2048, 6144, 2048, 4096}};
    const int MB = 256, NL = 5;
    for (const Shapes& sh : shapes) {
        const int H = sh.H, I = sh.I, QD = sh.QD, NQKV = sh.NQKV;
        void *Wqkv[NL], *Wo[NL], *Wgu[NL], *Wdn[NL];
        for (int l = 0; l < NL; ++l) {
            Wqkv[l] = dev_bf16((size_t)NQKV * H, 1.f / sqrtf((float)H));
            Wo[l] = dev_bf16((size_t)H * QD, 1.f / sqrtf((float)QD));
            Wgu[l] = dev_bf16((size_t)2 * I * H, 1.f / sqrtf((float)H));
            Wdn[l] = dev_bf16((size_t)H * I, 1.f / sqrtf((float)I));
        }
        void *Pqkv[NL], *Po[NL], *Pgu[NL], *Pdn[NL];
        for (int l = 0; l < NL; ++l) {
            CHK(hipMalloc(&Pqkv[l], (size_t)NQKV * H * 2)); CHK(hipMalloc(&Po[l], (size_t)H * QD * 2)); CHK(hipMalloc(&Pgu[l], (size_t)2 * I * H * 2)); CHK(hipMalloc(&Pdn[l], (size_t)H * I * 2));
            skinny_pack((const bf16_t*)Wqkv[l], (bf16_t*)Pqkv[l], NQKV, H, 0, st); skinny_pack((const bf16_t*)Wo[l], (bf16_t*)Po[l], H, QD, 0, st);
            skinny_pack((const bf16_t*)Wgu[l], (bf16_t*)Pgu[l], 2 * I, H, I, st); skinny_pack((const bf16_t*)Wdn[l], (bf16_t*)Pdn[l], H, I, 0, st);
        }
        CHK(hipStreamSynchronize(st));
        void* gain = dev_bf16(H, 0.05f, 1.f);
        void* attn = dev_bf16((size_t)MB * QD, 1.f);             // o_proj input
        void* res = dev_bf16((size_t)MB * H, 1.f);               // residual stream before o_proj
        void* h = dev_bf16((size_t)MB * H, 1.f);                 // residual stream after o_proj / down
        void* xn = dev_bf16((size_t)MB * H, 1.f);
        void* qkv_a = dev_bf16((size_t)MB * NQKV, 1.f); void* qkv_b = dev_bf16((size_t)MB * NQKV, 1.f);
        void* act_a = dev_bf16((size_t)MB * I, 1.f); void* act_b = dev_bf16((size_t)MB * I, 1.f);
        float* ssq; CHK(hipMalloc(&ssq, (size_t)MB * (H / 16) * 4)); CHK(hipMemset(ssq, 0xff, (size_t)MB * (H / 16) * 4));

        auto oproj = [&](int l, int B, bool with_ssq) {
            SkinnyArgs k{};
            k.X = (const bf16_t*)attn; k.ldx = QD; k.M = B; k.W = (const bf16_t*)Wo[l]; k.N = H; k.res = (const bf16_t*)res; k.ldr = H;
            k.Y = (bf16_t*)h; k.ldy = H; if (with_ssq) { k.ssq_out = ssq; k.ssq_ld = H / 16; }
            if (g_packed) k.Wp = (const bf16_t*)Po[l]; skinny_launch<SK_RESIDUAL>(k, QD, st);
        };
        auto down = [&](int l, int B, bool with_ssq, const void* a) {
            SkinnyArgs k{};
            k.X = (const bf16_t*)a; k.ldx = I; k.M = B; k.W = (const bf16_t*)Wdn[l]; k.N = H; k.res = (const bf16_t*)h; k.ldr = H;
            k.Y = (bf16_t*)h; k.ldy = H; if (with_ssq) { k.ssq_out = ssq; k.ssq_ld = H / 16; }
            if (g_packed) k.Wp = (const bf16_t*)Pdn[l]; skinny_launch<SK_RESIDUAL>(k, I, st);
        };
        auto qkvg = [&](int l, int B, bool fused, void* y) {
            SkinnyArgs k{};
            k.M = B; k.W = (const bf16_t*)Wqkv[l]; k.N = NQKV; k.Y = (bf16_t*)y; k.ldy = NQKV;
            if (fused) { k.X = (const bf16_t*)h; k.ldx = H; k.ssq = ssq; k.gain = (const bf16_t*)gain; k.eps = 1e-6f; }
            else { rmsnorm_rows(h, H, gain, H, B, xn); k.X = (const bf16_t*)xn; k.ldx = H; }
            if (g_packed) k.Wp = (const bf16_t*)Pqkv[l]; skinny_launch<SK_STORE>(k, H, st);
        };
        auto gateup = [&](int l, int B, bool fused, void* y) {
            SkinnyArgs k{};
            k.M = B; k.W = (const bf16_t*)Wgu[l]; k.N = 2 * I; k.Y = (bf16_t*)y; k.ldy = I;
            if (fused) { k.X = (const bf16_t*)h; k.ldx = H; k.ssq = ssq; k.gain = (const bf16_t*)gain; k.eps = 1e-6f; }
            else { rmsnorm_rows(h, H, gain, H, B, xn); k.X = (const bf16_t*)xn; k.ldx = H; }
            if (g_packed) k.Wp = (const bf16_t*)Pgu[l]; skinny_launch<SK_SWIGLU>(k, H, st);
        };

        // ---- 1. checks ----
        for (int B : {48, 128, 250}) {
            CHK(hipMemset(ssq, 0xff, (size_t)MB * (H / 16) * 4));
            oproj(0, B, true);
            CHK(hipStreamSynchronize(st));
            auto hh = fetch_bf16(h, (size_t)B * H);
            auto pp = fetch_f32(ssq, (size_t)B * (H / 16));
            double worst = 0;
            std::vector<double> rs(B);
            for (int m = 0; m < B; ++m) {
                double tot = 0, got = 0;
                for (int c = 0; c < H; ++c) tot += (double)hh[(size_t)m * H + c] * hh[(size_t)m * H + c];
                for (int j = 0; j < H / 16; ++j) {
                    double blk = 0;
                    for (int c = 0; c < 16; ++c) blk += (double)hh[(size_t)m * H + j * 16 + c] * hh[(size_t)m * H + j * 16 + c];
                    worst = fmax(worst, fabs(pp[(size_t)m * (H / 16) + j] - blk) / (blk + 1e-9));
                    got += pp[(size_t)m * (H / 16) + j];
                }
                worst = fmax(worst, fabs(got - tot) / tot);
                rs[m] = 1.0 / sqrt(tot / H + 1e-6);
            }
            char nm[128];
            snprintf(nm, sizeof nm, "%s B=%d: sum-of-squares partials of the o_proj epilogue vs CPU (rel)", sh.name, B); report(nm, worst, 1e-5);
            // consumer: fused vs two-launch form vs CPU
            qkvg(1, B, true, qkv_a); qkvg(1, B, false, qkv_b);
            gateup(1, B, true, act_a); gateup(1, B, false, act_b);
            CHK(hipStreamSynchronize(st));
            auto qa = fetch_bf16(qkv_a, (size_t)B * NQKV), qb = fetch_bf16(qkv_b, (size_t)B * NQKV);
            auto ga = fetch_bf16(act_a, (size_t)B * I), gb = fetch_bf16(act_b, (size_t)B * I);
            auto Wq = fetch_bf16(Wqkv[1], (size_t)NQKV * H); auto gw = fetch_bf16(gain, H);
            size_t ndiff = 0; double dmax = 0, cmax = 0;
            for (size_t i = 0; i < qa.size(); ++i) { if (qa[i] != qb[i]) ++ndiff; dmax = fmax(dmax, fabs(qa[i] - qb[i])); }
            for (int m = 0; m < B; m += 7)
                for (int n = 0; n < NQKV; n += 61) {
                    double acc = 0;
                    for (int c = 0; c < H; ++c) acc += (double)Wq[(size_t)n * H + c] * rbf(rbf((float)(hh[(size_t)m * H + c] * rs[m])) * gw[c]);
                    cmax = fmax(cmax, fabs(acc - qa[(size_t)m * NQKV + n]));
                }
            snprintf(nm, sizeof nm, "%s B=%d: normalising qkv GEMM vs CPU double (abs; values O(1))", sh.name, B); report(nm, cmax, 3e-2);
            snprintf(nm, sizeof nm, "%s B=%d: ... vs the two-launch form (max abs diff)", sh.name, B); report(nm, dmax, 3e-2);
            snprintf(nm, sizeof nm, "%s B=%d: ... share of elements that differ from the two-launch form", sh.name, B); report(nm, (double)ndiff / qa.size(), 0.05);
            ndiff = 0; dmax = 0;
            for (size_t i = 0; i < ga.size(); ++i) { if (ga[i] != gb[i]) ++ndiff; dmax = fmax(dmax, fabs(ga[i] - gb[i])); }
            snprintf(nm, sizeof nm, "%s B=%d: normalising gate|up + SwiGLU vs the two-launch form (max abs diff)", sh.name, B); report(nm, dmax, 3e-2);
            snprintf(nm, sizeof nm, "%s B=%d: ... share of elements that differ", sh.name, B); report(nm, (double)ndiff / ga.size(), 0.05);
        }
        // ---- 1b. fragment-major weight copies against the row-major matrices: the same values into the same registers -> bit for bit ----
        for (int B : {1, 16, 33, 48, 64, 128, 200, 256}) {
            size_t bad = 0;
            auto snap = [&](const void* d, size_t n) { std::vector<unsigned short> v(n); CHK(hipMemcpy(v.data(), d, n * 2, hipMemcpyDeviceToHost)); return v; };
            std::vector<unsigned short> ref[6], got[6];
            for (int pk : {0, 1}) {
                g_packed = pk;
                std::vector<unsigned short>* out = pk ? got : ref;
                CHK(hipMemcpy(h, res, (size_t)MB * H * 2, hipMemcpyDeviceToDevice));
                CHK(hipMemset(ssq, 0, (size_t)MB * (H / 16) * 4));
                oproj(2, B, true); CHK(hipStreamSynchronize(st)); out[0] = snap(h, (size_t)B * H);
                qkvg(2, B, false, qkv_a); CHK(hipStreamSynchronize(st)); out[1] = snap(qkv_a, (size_t)B * NQKV);
                qkvg(2, B, true, qkv_a); CHK(hipStreamSynchronize(st)); out[2] = snap(qkv_a, (size_t)B * NQKV);       // (the normalising form)
                gateup(2, B, false, act_a); CHK(hipStreamSynchronize(st)); out[3] = snap(act_a, (size_t)B * I);
                gateup(2, B, true, act_b); CHK(hipStreamSynchronize(st)); out[4] = snap(act_b, (size_t)B * I);
                down(2, B, false, act_a); CHK(hipStreamSynchronize(st)); out[5] = snap(h, (size_t)B * H);             // (in place: res = Y = h)
            }
            for (int q = 0; q < 6; ++q) for (size_t i = 0; i < ref[q].size(); ++i) bad += ref[q][i] != got[q][i];
            char nm[160];
            snprintf(nm, sizeof nm, "%s B=%d: fragment-major == row-major weights, o_proj / qkv / gate|up / down (elements that differ)", sh.name, B); report(nm, (double)bad, 0.5);
            g_packed = 0;
        }
        // ---- 2. chains: one layer = qkv, o_proj, gate | up, down ----
        const int NLAY = 40;
        for (int B : {32, 64, 128}) {
            char nm[160];
            double tp[2] = {0, 0};
            for (int pk : {0, 1}) {
                g_packed = pk;
                const char* wn = pk ? "fragment-major" : "row-major     ";
                snprintf(nm, sizeof nm, "%s B=%3d  %s: norm + qkv, o_proj, norm + gate|up, down (6 launches)", sh.name, B, wn);
                tp[pk] = chain(nm, NLAY, 1, [&](int j) { const int l = j % NL; qkvg(l, B, false, qkv_b); oproj(l, B, false); gateup(l, B, false, act_b); down(l, B, false, act_b); });
                snprintf(nm, sizeof nm, "%s B=%3d  %s: qkv alone: norm + GEMM", sh.name, B, wn);
                chain(nm, NLAY, 1, [&](int j) { qkvg(j % NL, B, false, qkv_b); });
                snprintf(nm, sizeof nm, "%s B=%3d  %s: o_proj alone", sh.name, B, wn);
                chain(nm, NLAY, 1, [&](int j) { oproj(j % NL, B, false); });
                snprintf(nm, sizeof nm, "%s B=%3d  %s: gate|up alone: norm + GEMM", sh.name, B, wn);
                chain(nm, NLAY, 1, [&](int j) { gateup(j % NL, B, false, act_b); });
                snprintf(nm, sizeof nm, "%s B=%3d  %s: down alone", sh.name, B, wn);
                chain(nm, NLAY, 1, [&](int j) { down(j % NL, B, false, act_b); });
            }
            printf("   -> fragment-major weights save %.2f us per layer (%.1f %%)\n", tp[0] - tp[1], 100.0 * (tp[0] - tp[1]) / tp[0]);
            g_packed = 0;
            snprintf(nm, sizeof nm, "%s B=%3d  round-4 form: norm + qkv, o_proj, norm + gate|up, down (6 launches)", sh.name, B);
            const double t0 = chain(nm, NLAY, 1, [&](int j) { const int l = j % NL; qkvg(l, B, false, qkv_b); oproj(l, B, false); gateup(l, B, false, act_b); down(l, B, false, act_b); });
            snprintf(nm, sizeof nm, "%s B=%3d  fused form:   qkv(norm), o_proj(+ssq), gate|up(norm), down(+ssq) (4 launches)", sh.name, B);
            const double t1 = chain(nm, NLAY, 1, [&](int j) { const int l = j % NL; qkvg(l, B, true, qkv_a); oproj(l, B, true); gateup(l, B, true, act_a); down(l, B, true, act_a); });
            printf("   -> %.2f us per layer saved (%.1f %%)\n", t0 - t1, 100.0 * (t0 - t1) / t0);
            snprintf(nm, sizeof nm, "%s B=%3d  qkv alone: norm + GEMM", sh.name, B);
            chain(nm, NLAY, 1, [&](int j) { qkvg(j % NL, B, false, qkv_b); });
            snprintf(nm, sizeof nm, "%s B=%3d  qkv alone: normalising GEMM", sh.name, B);
            chain(nm, NLAY, 1, [&](int j) { qkvg(j % NL, B, true, qkv_a); });
            snprintf(nm, sizeof nm, "%s B=%3d  gate|up alone: norm + GEMM", sh.name, B);
            chain(nm, NLAY, 1, [&](int j) { gateup(j % NL, B, false, act_b); });
            snprintf(nm, sizeof nm, "%s B=%3d  gate|up alone: normalising GEMM", sh.name, B);
            chain(nm, NLAY, 1, [&](int j) { gateup(j % NL, B, true, act_a); });
            snprintf(nm, sizeof nm, "%s B=%3d  o_proj alone, without / with the partials", sh.name, B);
            chain(nm, NLAY, 1, [&](int j) { oproj(j % NL, B, false); });
            chain(nm, NLAY, 1, [&](int j) { oproj(j % NL, B, true); });
            snprintf(nm, sizeof nm, "%s B=%3d  down alone", sh.name, B);
            chain(nm, NLAY, 1, [&](int j) { down(j % NL, B, false, act_b); });
            snprintf(nm, sizeof nm, "%s B=%3d  rmsnorm_batch_kernel alone (one wave per token)", sh.name, B);
            chain(nm, NLAY, 1, [&](int) { rmsnorm_rows(h, H, gain, H, B, xn); });
        }
        for (int l = 0; l < NL; ++l) { CHK(hipFree(Wqkv[l])); CHK(hipFree(Wo[l])); CHK(hipFree(Wgu[l])); CHK(hipFree(Wdn[l])); CHK(hipFree(Pqkv[l])); CHK(hipFree(Po[l])); CHK(hipFree(Pgu[l])); CHK(hipFree(Pdn[l])); }
    }
    printf("%s\n", g_fail ? "SELF-CHECK FAILED" : "self-checks ok");
    return g_fail ? 1 : 0;
}
