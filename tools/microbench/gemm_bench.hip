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
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (argc > 2 && !strcmp(argv[2], "codec")) {
        // The codec's decoder convs of one 300-frame piece WITH their real epilogues (bias, residual, SnakeBeta second output),
        // through the product's own tile choice: LDS-parked row-contiguous epilogue (product) vs the register-layout one
        // (epi_legacy = 1), outputs compared bit for bit.  kind: T = transposed conv (2 taps, Y + Y2), 1 = conv1 (k7, Y2 only),
        // 2 = conv2 (1x1, residual + Y + Y2).
        struct CS { const char* name; char kind; int M, Cin, N, co, dil; };
        const std::vector<CS> cs = {
            {"b1 convT  1536 -> 8 x 768", 'T', 1199, 1536, 6144, 768, 1}, {"b1 conv1 k7  768", '1', 9592, 768, 768, 768, 1}, {"b1 conv2 1x1 768", '2', 9592, 768, 768, 768, 1},
            {"b2 convT   768 -> 5 x 384", 'T', 9591, 768, 1920, 384, 1}, {"b2 conv1 k7  384 d3", '1', 47955, 384, 384, 384, 3}, {"b2 conv2 1x1 384", '2', 47955, 384, 384, 384, 1},
            {"b3 convT   384 -> 4 x 192", 'T', 47954, 384, 768, 192, 1}, {"b3 conv1 k7  192 d9", '1', 191816, 192, 192, 192, 9}, {"b3 conv2 1x1 192", '2', 191816, 192, 192, 192, 1},
            {"b4 convT   192 -> 3 x 96", 'T', 191815, 192, 288, 96, 1}, {"b4 conv1 k7   96", '1', 575445, 96, 96, 96, 1}, {"b4 conv2 1x1  96", '2', 575445, 96, 96, 96, 1},
            {"chunk b1 conv2 1x1 768", '2', 416, 768, 768, 768, 1}, {"chunk b4 conv1 k7 96", '1', 24960, 96, 96, 96, 1}, {"chunk b4 conv2 1x1 96", '2', 24960, 96, 96, 96, 1},
        };
        double tot_new = 0, tot_old = 0;
        for (auto& c : cs) {
            const int taps = c.kind == 'T' ? 2 : (c.kind == '1' ? 7 : 1);
            const size_t K = (size_t)taps * c.Cin, arows = (size_t)c.M + (c.kind == 'T' ? 1 : 0);
            const size_t na = arows * c.Cin, nw = (size_t)c.N * K, ny = (size_t)c.M * c.N;
            std::vector<uint16_t> ha(na), hw(nw), hr(ny), hc(4 * (size_t)c.co);
            uint32_t x = 4242;
            auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 16; };
            for (auto& v : ha) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 128.f);
            for (auto& v : hw) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 1024.f);
            for (auto& v : hr) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 128.f);
            for (size_t i = 0; i < hc.size(); ++i) hc[i] = f_to_bf16_host(i < (size_t)c.co ? (float)((int)(rnd() & 0xff) - 128) / 512.f : 0.75f + (float)(rnd() & 0xff) / 512.f);
            void *A, *W, *R, *C, *Y[2], *Y2[2];
            (void)hipMalloc(&A, na * 2); (void)hipMalloc(&W, nw * 2); (void)hipMalloc(&R, ny * 2); (void)hipMalloc(&C, hc.size() * 2);
            for (int v = 0; v < 2; ++v) { (void)hipMalloc(&Y[v], ny * 2); (void)hipMalloc(&Y2[v], ny * 2); (void)hipMemset(Y[v], 0, ny * 2); (void)hipMemset(Y2[v], 0, ny * 2); }
            (void)hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); (void)hipMemcpy(W, hw.data(), nw * 2, hipMemcpyHostToDevice);
            (void)hipMemcpy(R, hr.data(), ny * 2, hipMemcpyHostToDevice); (void)hipMemcpy(C, hc.data(), hc.size() * 2, hipMemcpyHostToDevice);
            GemmArgs a{};
            a.A = A; a.lda = c.Cin; a.M = c.M; a.a_rows = (int)arows; a.n_taps = taps; a.Cin = c.Cin; a.W = W; a.N = c.N;
            if (c.kind == 'T') { a.tap_off[0] = 1; a.tap_off[1] = 0; } else for (int i = 0; i < taps; ++i) a.tap_off[i] = -(taps - 1 - i) * c.dil;
            a.bias = C; a.bias_mod = c.co; a.ldy = c.N;
            a.sn_a = (const uint16_t*)C + 2 * c.co; a.sn_ib = (const uint16_t*)C + 3 * c.co;
            if (c.kind == '2') { a.res = R; a.ldr = c.N; }
            float ms[2];
            for (int v = 0; v < 2; ++v) {
                GemmArgs b = a; b.epi_legacy = v; b.Y = c.kind == '1' ? nullptr : Y[v]; b.Y2 = Y2[v];
                gemm_launch<bf16_t>(b, s); (void)hipStreamSynchronize(s);
                (void)hipEventRecord(e0, s);
                for (int r = 0; r < reps; ++r) gemm_launch<bf16_t>(b, s);
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms[v], e0, e1); ms[v] /= reps;
            }
            std::vector<uint16_t> y0(ny), y1(ny);
            size_t bad = 0;
            (void)hipMemcpy(y0.data(), Y2[0], ny * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(y1.data(), Y2[1], ny * 2, hipMemcpyDeviceToHost);
            for (size_t i = 0; i < ny; ++i) bad += y0[i] != y1[i];
            if (c.kind != '1') {
                (void)hipMemcpy(y0.data(), Y[0], ny * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(y1.data(), Y[1], ny * 2, hipMemcpyDeviceToHost);
                for (size_t i = 0; i < ny; ++i) bad += y0[i] != y1[i];
            }
            const double fl = 2.0 * c.M * c.N * K;
            const double bytes = 2.0 * ((double)na + (c.kind == '2' ? 3.0 : (c.kind == 'T' ? 2.0 : 1.0)) * ny);      // A + [residual] + outputs
            printf("%-28s M=%7d N=%5d K=%5zu  parked %8.2f us (%6.1f TFLOP/s, %5.2f TB/s)   register-layout %8.2f us   outputs %s\n", c.name, c.M, c.N, K,
                   ms[0] * 1e3, fl / (ms[0] * 1e-3) / 1e12, bytes / (ms[0] * 1e-3) / 1e12, ms[1] * 1e3, bad ? "DIFFER  <-- MISMATCH" : "bit-identical");
            if (c.name[0] == 'b') { tot_new += ms[0]; tot_old += ms[1]; }
            (void)hipFree(A); (void)hipFree(W); (void)hipFree(R); (void)hipFree(C);
            for (int v = 0; v < 2; ++v) { (void)hipFree(Y[v]); (void)hipFree(Y2[v]); }
        }
        printf("one pass over the 12 decoder-block convs of a 300-frame piece (x1; each residual unit runs 3x): parked %.3f ms, register-layout %.3f ms\n", tot_new, tot_old);
        return 0;
    }
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
