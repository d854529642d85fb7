// Micro-benchmark of the product's implicit-GEMM kernel (csrc/codec_kernels.cuh, through gemm_launch's own tile choice) on the
// shapes the path runs: prefill GEMMs (200 and 4096 tokens), codec convs (full 370-frame decode and a streaming chunk), and
// 4096^3 for comparison with published ladders.  Development aid; prints TFLOP/s per shape.  usage: gemm_bench [reps]
#include "../../faster-qwen3-tts_amd/csrc/codec_kernels.cuh"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
using namespace fq3;

static uint16_t f_to_bf16_host(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float bf16_to_f_host(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }

struct Shape { const char* name; int M, N, Cin, taps, dil; bool ws; };

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 10;
    std::vector<Shape> shapes = {
        {"square 4096^3", 4096, 4096, 4096, 1, 1},
        {"big-tile overhead probe", 4096, 4096, 128, 1, 1},
        {"big-tile odd step count", 4096, 4096, 96 * 7, 1, 1},
        {"prefill4k qkv  (1.7B)", 4096, 4096, 2048, 1, 1},
        {"prefill4k gate_up", 4096, 12288, 2048, 1, 1},
        {"prefill4k down", 4096, 2048, 6144, 1, 1},
        {"prefill4k o_proj", 4096, 2048, 2048, 1, 1},
        {"prefill200 qkv (0.6B)", 200, 4096, 1024, 1, 1},
        {"prefill200 gate_up", 200, 6144, 1024, 1, 1},
        {"prefill200 down", 200, 1024, 3072, 1, 1},
        {"prefill200 o     +splitK", 200, 1024, 2048, 1, 1, true},
        {"prefill200 down  +splitK", 200, 1024, 3072, 1, 1, true},
        {"codec370 dec.0 k7", 1480, 1536, 1024, 7, 1},
        {"codec370 b1 conv1 k7", 11832, 768, 768, 7, 1},
        {"codec370 b2 conv1 k7", 59155, 384, 384, 7, 3},
        {"codec370 b3 conv1 k7", 236616, 192, 192, 7, 9},
        {"codec370 b4 conv1 k7", 709845, 96, 96, 7, 1},
        {"codec370 b4 conv2 k1", 709845, 96, 96, 1, 1},
        {"chunk13 dec.0 k7", 52, 1536, 1024, 7, 1},
        {"chunk13 b1 conv1 k7", 416, 768, 768, 7, 1},
        {"chunk13 b4 conv1 k7", 24960, 96, 96, 7, 1},
    };
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    if (argc > 2) {        // variant sweep on 4096 x 4096 x 4096: tile shape x prefetch depth
        const int M = 4096, N = 4096, K = 4096;
        void *A, *W, *Y;
        hipMalloc(&A, (size_t)M * K * 2); hipMalloc(&W, (size_t)N * K * 2); hipMalloc(&Y, (size_t)M * N * 2);
        hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(W, 0x3c, (size_t)N * K * 2);
        GemmArgs a{};
        a.A = A; a.lda = K; a.M = M; a.a_rows = M; a.n_taps = 1; a.Cin = K; a.W = W; a.N = N; a.bias_mod = N; a.Y = Y; a.ldy = N;
        auto run = [&](const char* name, auto kern, int bm, int bn) {
            dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm);
            hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
            hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("%-28s %9.3f us  %8.1f TFLOP/s\n", name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        };
        auto rung = [&](const char* name, auto go) {
            go(a, s); hipStreamSynchronize(s);
            hipEventRecord(e0, s);
            for (int r = 0; r < reps; ++r) go(a, s);
            hipEventRecord(e1, s); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("%-28s %9.3f us  %8.1f TFLOP/s\n", name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        };
        {   // big tile vs the register-staged kernel on random operands: must agree bit for bit (same MFMA chain per element)
            std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
            uint32_t x = 777;
            auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 16; };
            for (auto& v : ha) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 128.f);
            for (auto& v : hw) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 1024.f);
            hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
            void* Y2; hipMalloc(&Y2, (size_t)M * N * 2);
            glds_go<64, 2>(a, s);
            GemmArgs b = a; b.Y = Y2;
            big_go(b, s);
            hipStreamSynchronize(s);
            std::vector<uint16_t> y1((size_t)M * N), y2((size_t)M * N);
            hipMemcpy(y1.data(), Y, y1.size() * 2, hipMemcpyDeviceToHost); hipMemcpy(y2.data(), Y2, y2.size() * 2, hipMemcpyDeviceToHost);
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < y1.size(); ++i) if (y1[i] != y2[i]) { if (!bad) first = i; ++bad; }
            printf("big 256x256 vs glds 128x64 on random operands: %zu / %zu elements differ%s\n", bad, y1.size(), bad ? "  <-- MISMATCH" : " (bit-identical)");
            if (bad) printf("  first at row %zu col %zu: %04x vs %04x\n", first / N, first % N, y1[first], y2[first]);
            {   // the 256 x 128 variant on the same operands
                big_go_t<true, 128>(b, s);
                hipStreamSynchronize(s);
                hipMemcpy(y2.data(), Y2, y2.size() * 2, hipMemcpyDeviceToHost);
                size_t bad2 = 0;
                for (size_t i = 0; i < y1.size(); ++i) bad2 += y1[i] != y2[i];
                printf("big 256x128 vs glds 128x64 on random operands: %zu / %zu elements differ%s\n", bad2, y1.size(), bad2 ? "  <-- MISMATCH" : " (bit-identical)");
            }
            hipFree(Y2);
            rung("big 256x128 ring-4 (random)", big_go_t<true, 128>);
            rung("big 256x256 ring-4 (random)", big_go);
            rung("glds 128x64 2 st (random)", glds_go<64, 2>);
            hipMemset(A, 0x3c, (size_t)M * K * 2); hipMemset(W, 0x3c, (size_t)N * K * 2);
        }
        rung("big 256x256 ring-4", big_go);
        rung("glds 128x64 2 stages", glds_go<64, 2>);
        rung("glds 128x64 3 stages", glds_go<64, 3>);
        rung("glds 128x128 2 stages", glds_go<128, 2>);
        rung("glds 128x128 3 stages", glds_go<128, 3>);
        run("128x64 PF=2", conv_gemm_kernel<bf16_t, 128, 64, 2>, 128, 64);
        run("128x64 PF=4", conv_gemm_kernel<bf16_t, 128, 64, 4>, 128, 64);
        run("128x64 PF=8", conv_gemm_kernel<bf16_t, 128, 64, 8>, 128, 64);
        run("64x64 PF=4", conv_gemm_kernel<bf16_t, 64, 64, 4>, 64, 64);
        run("64x64 PF=8", conv_gemm_kernel<bf16_t, 64, 64, 8>, 64, 64);
        run("128x128 PF=2", conv_gemm_kernel<bf16_t, 128, 128, 2>, 128, 128);
        run("128x128 PF=4", conv_gemm_kernel<bf16_t, 128, 128, 4>, 128, 128);
        return 0;
    }
    for (auto& sh : shapes) {
        const size_t K = (size_t)sh.taps * sh.Cin;
        const size_t na = (size_t)sh.M * sh.Cin, nw = (size_t)sh.N * K, ny = (size_t)sh.M * sh.N;
        std::vector<uint16_t> ha(na), hw(nw);
        uint32_t x = 12345;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 16; };
        for (auto& v : ha) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 128.f);
        for (auto& v : hw) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 1024.f);
        void *A, *W, *Y;
        hipMalloc(&A, na * 2); hipMalloc(&W, nw * 2); hipMalloc(&Y, ny * 2);
        hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); hipMemcpy(W, hw.data(), nw * 2, hipMemcpyHostToDevice);
        GemmArgs a{};
        a.A = A; a.lda = sh.Cin; a.M = sh.M; a.a_rows = sh.M; a.n_taps = sh.taps; a.Cin = sh.Cin; a.W = W; a.N = sh.N;
        for (int i = 0; i < sh.taps; ++i) a.tap_off[i] = -(sh.taps - 1 - i) * sh.dil;
        a.bias_mod = sh.N; a.Y = Y; a.ldy = sh.N;
        void* ws = nullptr;
        if (sh.ws) { hipMalloc(&ws, (size_t)(8 << 20) * 4); a.ws = (float*)ws; a.ws_floats = 8 << 20; }
        gemm_launch<bf16_t>(a, s);
        hipStreamSynchronize(s);
        hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) gemm_launch<bf16_t>(a, s);
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        // spot check of 16 outputs against a host dot product
        std::vector<uint16_t> hy(ny);
        hipMemcpy(hy.data(), Y, ny * 2, hipMemcpyDeviceToHost);
        double maxerr = 0;
        for (int t = 0; t < 16; ++t) {
            const int m = (int)(((uint64_t)t * 7919 + 13) % sh.M), n = (int)(((uint64_t)t * 104729 + 7) % sh.N);
            double acc = 0;
            for (int tap = 0; tap < sh.taps; ++tap) {
                const int ar = m + a.tap_off[tap];
                if (ar < 0 || ar >= sh.M) continue;
                for (int c = 0; c < sh.Cin; ++c) acc += (double)bf16_to_f_host(ha[(size_t)ar * sh.Cin + c]) * bf16_to_f_host(hw[(size_t)n * K + (size_t)tap * sh.Cin + c]);
            }
            const double got = bf16_to_f_host(hy[(size_t)m * sh.N + n]);
            const double err = fabs(got - acc) / (fabs(acc) + 1.0);
            if (err > maxerr) maxerr = err;
        }
        const double fl = 2.0 * sh.M * sh.N * K;
        printf("%-24s M=%7d N=%5d K=%5zu  %9.3f us  %8.1f TFLOP/s  check %.2e %s\n", sh.name, sh.M, sh.N, K, ms * 1e3, fl / (ms * 1e-3) / 1e12,
               maxerr, maxerr < 2e-2 ? "ok" : "MISMATCH");
        hipFree(A); hipFree(W); hipFree(Y); if (ws) hipFree(ws);
    }
    return 0;
}
