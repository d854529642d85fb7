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
        {"packed10 qkv (0.6B)", 2000, 4096, 1024, 1, 1},
        {"packed10 gate_up", 2000, 6144, 1024, 1, 1},
        {"packed10 o_proj", 2000, 1024, 2048, 1, 1},
        {"packed10 down", 2000, 1024, 3072, 1, 1},
        {"packed6 o_proj", 1200, 1024, 2048, 1, 1},
        {"packed6 down", 1200, 1024, 3072, 1, 1},
        {"packed10 o_proj (1.7B)", 2000, 2048, 2048, 1, 1},
        {"packed10 down (1.7B)", 2000, 2048, 6144, 1, 1},
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
        {"chunk13 dec.0 as ONE tap", 52, 1536, 7168, 1, 1},
        {"chunk13 dec.0 k7 bf16x2", 52, 1536, 2048, 7, 1},
        {"chunk13 b1 conv1 bf16x2", 416, 768, 1536, 7, 3},
        {"chunk13 b1 up k2 bf16x2", 52, 6144, 3072, 2, 1},
        {"chunk front GEMM", 33, 512, 2048, 1, 1},
        {"chunk13 b4 conv1 k7", 24960, 96, 96, 7, 1},
    };
    hipStream_t s; (void)hipStreamCreate(&s);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (argc > 2 && !strcmp(argv[2], "skinny1")) {
        // one shape of the weight-stationary kernel, a few launches: the workload of the rocprofv3 --pmc passes
        // usage: gemm_bench <reps> skinny1 <M> <N> <K> [RB] [mt]
        const int M = atoi(argv[3]), N = atoi(argv[4]), K = atoi(argv[5]), rbf = argc > 6 ? atoi(argv[6]) : 0, mt = argc > 7 ? atoi(argv[7]) : 0;
        void *X, *W, *Y;
        (void)hipMalloc(&X, (size_t)M * K * 2); (void)hipMalloc(&W, (size_t)N * K * 2); (void)hipMalloc(&Y, (size_t)M * N * 2);
        (void)hipMemset(X, 0x3c, (size_t)M * K * 2); (void)hipMemset(W, 0x3c, (size_t)N * K * 2);
        SkinnyArgs q{}; q.X = (const bf16_t*)X; q.ldx = K; q.M = M; q.W = (const bf16_t*)W; q.N = N; q.Y = (bf16_t*)Y; q.ldy = N; q.mt = mt;
        for (int r = 0; r < reps; ++r) skinny_launch<SK_STORE>(q, K, s, rbf);
        (void)hipStreamSynchronize(s);
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "skinny")) {
        // The short-prompt prefill's four GEMMs: the tiled / split-K kernels (what gemm_launch picked before) against the
        // weight-stationary kernel of skinny_gemm.cuh, over a ROTATION of weight copies larger than the Infinity Cache (as in the real
        // prefill, where every layer's weights come from HBM).  Outputs compared element by element (the accumulation order differs:
        // a few bf16 ulps at most) and against a host double dot product on a sample.
        struct SS { const char* name; int M, N, K; int epi; };        // epi: 0 store, 1 residual, 2 [gate|up] + SwiGLU
        std::vector<SS> ss;
        // usage: gemm_bench <reps> skinny [M,M,...]   (default 200,52,416: the prefill's row counts; 64,128: the lock-step batch's lane counts)
        std::vector<int> Ms = {200, 52, 416};
        if (argc > 3) { Ms.clear(); for (char* t = strtok(argv[3], ","); t; t = strtok(nullptr, ",")) Ms.push_back(atoi(t)); }
        for (int M : Ms) {
            ss.push_back({"0.6B qkv", M, 4096, 1024, 0}); ss.push_back({"0.6B gate_up+silu", M, 6144, 1024, 2});
            ss.push_back({"0.6B o_proj", M, 1024, 2048, 1}); ss.push_back({"0.6B down", M, 1024, 3072, 1});
            if (M == 200 || argc > 3) {
                ss.push_back({"1.7B qkv", M, 4096, 2048, 0}); ss.push_back({"1.7B gate_up+silu", M, 12288, 2048, 2});
                ss.push_back({"1.7B o_proj", M, 2048, 2048, 1}); ss.push_back({"1.7B down", M, 2048, 6144, 1});
            }
        }
        void* ws; (void)hipMalloc(&ws, (size_t)(8 << 20) * 4);
        for (auto& c : ss) {
            const size_t nx = (size_t)c.M * c.K, nw = (size_t)c.N * c.K, nyf = (size_t)c.M * c.N, ny = c.epi == 2 ? nyf / 2 : nyf;
            const int copies = (int)((320ull << 20) / (nw * 2)) + 1;
            std::vector<uint16_t> hx(nx), hw(nw), hr(ny);
            uint32_t x = 99 + c.K;
            auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 16; };
            for (auto& v : hx) v = f_to_bf16_host((float)((int)(rnd() & 0xfff) - 2048) / 1931.f);
            for (auto& v : hw) v = f_to_bf16_host((float)((int)(rnd() & 0xfff) - 2048) / 16411.f);
            for (auto& v : hr) v = f_to_bf16_host((float)((int)(rnd() & 0xfff) - 2048) / 1931.f);
            void *X, *W, *R, *GU, *Y[2];
            (void)hipMalloc(&X, nx * 2); (void)hipMalloc(&W, nw * 2 * copies); (void)hipMalloc(&R, ny * 2); (void)hipMalloc(&GU, nyf * 2);
            for (int v = 0; v < 2; ++v) { (void)hipMalloc(&Y[v], ny * 2); (void)hipMemset(Y[v], 0, ny * 2); }
            (void)hipMemcpy(X, hx.data(), nx * 2, hipMemcpyHostToDevice); (void)hipMemcpy(R, hr.data(), ny * 2, hipMemcpyHostToDevice);
            for (int k = 0; k < copies; ++k) (void)hipMemcpy((char*)W + (size_t)k * nw * 2, hw.data(), nw * 2, hipMemcpyHostToDevice);
            auto args = [&](int k, int v) {
                GemmArgs a{};
                a.A = X; a.lda = c.K; a.M = c.M; a.a_rows = c.M; a.n_taps = 1; a.Cin = c.K; a.W = (char*)W + (size_t)k * nw * 2; a.N = c.N; a.bias_mod = c.N;
                a.Y = c.epi == 2 ? GU : Y[v]; a.ldy = c.N; a.ws = (float*)ws; a.ws_floats = 8 << 20; a.no_skinny = v == 0;
                if (c.epi == 1) { a.res = R; a.ldr = c.N; }
                return a;
            };
            auto go = [&](int k, int v) { GemmArgs a = args(k, v); if (c.epi == 2) gemm_swiglu_halves<bf16_t>(a, Y[v], s); else gemm_launch<bf16_t>(a, s); };
            float ms[2];
            for (int v = 0; v < 2; ++v) {
                if (getenv("SK_TRACE")) { printf("  M=%d %s: %s ...\n", c.M, c.name, v ? "skinny" : "tiled"); fflush(stdout); }
                go(0, v); (void)hipStreamSynchronize(s);
                (void)hipEventRecord(e0, s);
                for (int r = 0; r < reps; ++r) go(r % copies, v);
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms[v], e0, e1); ms[v] /= reps;
            }
            std::vector<uint16_t> y0(ny), y1(ny);
            (void)hipMemcpy(y0.data(), Y[0], ny * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(y1.data(), Y[1], ny * 2, hipMemcpyDeviceToHost);
            size_t differ = 0; double maxd = 0, maxref = 0;
            for (size_t i = 0; i < ny; ++i) {
                const double a0 = bf16_to_f_host(y0[i]), a1 = bf16_to_f_host(y1[i]);
                differ += y0[i] != y1[i];
                maxd = fmax(maxd, fabs(a0 - a1)); maxref = fmax(maxref, fabs(a0));
            }
            double maxerr = 0;                      // host double reference on a sample (residual / SwiGLU applied as the kernels do)
            const int ycols = c.epi == 2 ? c.N / 2 : c.N;
            for (int t = 0; t < 24; ++t) {
                const int m = (int)(((uint64_t)t * 7919 + 13) % c.M), n = t < 2 ? ycols - 1 - t : (int)(((uint64_t)t * 104729 + 7) % ycols);
                auto dot = [&](int wr) { double acc = 0; for (int k = 0; k < c.K; ++k) acc += (double)bf16_to_f_host(hx[(size_t)m * c.K + k]) * bf16_to_f_host(hw[(size_t)wr * c.K + k]); return acc; };
                double want;
                if (c.epi == 2) { const double g = dot(n), u = dot(c.N / 2 + n); want = g / (1.0 + exp(-g)) * u; }
                else want = dot(n) + (c.epi == 1 ? bf16_to_f_host(hr[(size_t)m * c.N + n]) : 0.0);
                const double got = bf16_to_f_host(y1[(size_t)m * ycols + n]);
                maxerr = fmax(maxerr, fabs(got - want) / (fabs(want) + 1.0));
            }
            // (row blocks per wave, workgroups per row group) sweep of the weight-stationary kernel
            char sweep[512]; int sp = 0;
            for (int rbf : {1, 2, 3, -1}) for (int mt : {1, 2, 4}) {
                if (rbf > 0 && ((c.N / 16) % rbf || (c.K > 3072 && rbf > 1) || (c.K > 2048 && rbf > 2))) continue;
                if (rbf < 0 && mt > 1) continue;           // -1: the launcher's own choice WITHOUT the staggered tile walk
                auto gs = [&](int k) {
                    SkinnyArgs q{}; q.X = (const bf16_t*)X; q.ldx = c.K; q.M = c.M; q.W = (const bf16_t*)((char*)W + (size_t)k * nw * 2); q.N = c.N;
                    q.res = (const bf16_t*)R; q.ldr = c.N; q.Y = (bf16_t*)Y[1]; q.ldy = c.epi == 2 ? c.N / 2 : c.N; q.mt = rbf < 0 ? 0 : mt; q.no_stagger = rbf < 0; const int rbf_ = rbf < 0 ? 0 : rbf;
                    if (c.epi == 2) skinny_launch<SK_SWIGLU>(q, c.K, s, rbf_); else if (c.epi == 1) skinny_launch<SK_RESIDUAL>(q, c.K, s, rbf_); else skinny_launch<SK_STORE>(q, c.K, s, rbf_);
                };
                if (getenv("SK_TRACE")) { printf("  sweep RB %d mt %d ...\n", rbf, mt); fflush(stdout); }
                gs(0); (void)hipStreamSynchronize(s);
                (void)hipEventRecord(e0, s);
                for (int r = 0; r < reps; ++r) gs(r % copies);
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                float t; (void)hipEventElapsedTime(&t, e0, e1);
                if (rbf < 0) sp += snprintf(sweep + sp, sizeof(sweep) - sp, " unstaggered:%.1f", t / reps * 1e3); else sp += snprintf(sweep + sp, sizeof(sweep) - sp, " %dx%d:%.1f", rbf, mt, t / reps * 1e3);
            }
            const double wbytes = (double)nw * 2;
            printf("M=%3d %-20s N=%5d K=%4d  tiled %7.2f us   skinny %7.2f us (%5.2f TB/s of weights)   %zu / %zu outputs differ, max |d| %.3g of %.3g   host check %.2e %s   [RBxmt us:%s]\n",
                   c.M, c.name, c.N, c.K, ms[0] * 1e3, ms[1] * 1e3, wbytes / (ms[1] * 1e-3) / 1e12, differ, ny, maxd, maxref, maxerr,
                   maxerr < 2e-2 && maxd <= 0.02 * maxref ? "ok" : "MISMATCH", sweep);
            (void)hipFree(X); (void)hipFree(W); (void)hipFree(R); (void)hipFree(GU); (void)hipFree(Y[0]); (void)hipFree(Y[1]);
        }
        return 0;
    }
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
    if (argc > 2 && !strcmp(argv[2], "glds")) {
        // LDS-DMA 128 x 64 tile by ring depth on one shape (random operands; every depth must agree bit for bit with two stages)
        // usage: gemm_bench <reps> glds <M> <N> <K>
        const int M = atoi(argv[3]), N = atoi(argv[4]), K = atoi(argv[5]);
        std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
        uint32_t x = 4242;
        auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 16; };
        for (auto& v : ha) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 128.f);
        for (auto& v : hw) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 1024.f);
        void *A, *W, *Y;
        (void)hipMalloc(&A, ha.size() * 2); (void)hipMalloc(&W, hw.size() * 2); (void)hipMalloc(&Y, (size_t)M * N * 2);
        (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        GemmArgs a{};
        a.A = A; a.lda = K; a.M = M; a.a_rows = M; a.n_taps = 1; a.Cin = K; a.W = W; a.N = N; a.bias_mod = N; a.Y = Y; a.ldy = N;
        std::vector<uint16_t> y0((size_t)M * N), y1((size_t)M * N);
        auto rung = [&](const char* name, auto go, bool ref) {
            (void)hipMemset(Y, 0, (size_t)M * N * 2);
            go(a, s); (void)hipStreamSynchronize(s);
            (void)hipMemcpy((ref ? y0 : y1).data(), Y, y0.size() * 2, hipMemcpyDeviceToHost);
            size_t bad = 0; if (!ref) for (size_t i = 0; i < y0.size(); ++i) bad += y0[i] != y1[i];
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < reps; ++r) go(a, s);
            (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("M=%d N=%d K=%d  %-26s %9.3f us  %8.1f TFLOP/s  %s\n", M, N, K, name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12,
                   ref ? "(reference)" : bad ? "DIFFERS  <-- MISMATCH" : "bit-identical");
        };
        a.xcd_map = -1;
        rung("glds 128x64 2 st, plain order", glds_go<64, 2>, true);
        rung("glds 128x64 3 st, plain order", glds_go<64, 3>, false);
        rung("glds 128x64 4 st, plain order", glds_go<64, 4>, false);
        rung("glds 128x64 6 st, plain order", glds_go<64, 6>, false);
        rung("64x64 PF=4, plain order", gemm_go<bf16_t, 64, 64>, false);
        rung("gemm_launch, plain order", gemm_launch<bf16_t>, false);
        a.xcd_map = 0;
        rung("glds 128x64 2 st, XCD order", glds_go<64, 2>, false);
        rung("glds 128x64 3 st, XCD order", glds_go<64, 3>, false);
        rung("glds 128x64 4 st, XCD order", glds_go<64, 4>, false);
        rung("glds 128x64 6 st, XCD order", glds_go<64, 6>, false);
        rung("glds 128x64 8 waves 2 st", glds_go<64, 2, bf16_t, 512>, false);
        rung("glds 128x64 8 waves 3 st", glds_go<64, 3, bf16_t, 512>, false);
        rung("glds 128x64 8 waves 4 st", glds_go<64, 4, bf16_t, 512>, false);
        rung("glds 128x128 8 waves 3 st", glds_go<128, 3, bf16_t, 512>, false);
        rung("glds 64x64 8 waves 3 st", glds_go<64, 3, bf16_t, 512, 64>, false);
        rung("glds 64x64 8 waves 4 st", glds_go<64, 4, bf16_t, 512, 64>, false);
        rung("glds 64x128 8 waves 3 st", glds_go<128, 3, bf16_t, 512, 64>, false);
        rung("dbg 64x128 8w: no MFMAs", glds_go<128, 3, bf16_t, 512, 64, 1>, false);
        rung("dbg 64x128 8w: copies + barriers", glds_go<128, 3, bf16_t, 512, 64, 2>, false);
        rung("dbg 64x128 8w: copies only", glds_go<128, 3, bf16_t, 512, 64, 3>, false);
        rung("dbg 64x128 8w 4st: copies+barriers", glds_go<128, 4, bf16_t, 512, 64, 2>, false);
        rung("dbg 128x64 4w 2st: copies+barriers", glds_go<64, 2, bf16_t, 256, 128, 2>, false);
        rung("dbg 128x64 4w 2st: no MFMAs", glds_go<64, 2, bf16_t, 256, 128, 1>, false);
        rung("chain 32x32, 8 waves x 8 steps", chain_go<8, 8>, false);
        rung("chain 32x32, 16 waves x 4 steps", chain_go<16, 4>, false);
        rung("chain 32x32, 8 waves x 4 steps", chain_go<8, 4>, false);
        rung("chain 32x32, 4 waves x 8 steps", chain_go<4, 8>, false);
        rung("64x64 PF=4, XCD order", gemm_go<bf16_t, 64, 64>, false);
        rung("gemm_launch, XCD order", gemm_launch<bf16_t>, false);
        return 0;
    }
    if (argc > 2) {        // variant sweep on 4096 x 4096 x 4096: tile shape x prefetch depth
        const int M = 4096, N = 4096, K = 4096;
        void *A, *W, *Y;
        (void)hipMalloc(&A, (size_t)M * K * 2); (void)hipMalloc(&W, (size_t)N * K * 2); (void)hipMalloc(&Y, (size_t)M * N * 2);
        (void)hipMemset(A, 0x3c, (size_t)M * K * 2); (void)hipMemset(W, 0x3c, (size_t)N * K * 2);
        GemmArgs a{};
        a.A = A; a.lda = K; a.M = M; a.a_rows = M; a.n_taps = 1; a.Cin = K; a.W = W; a.N = N; a.bias_mod = N; a.Y = Y; a.ldy = N;
        auto run = [&](const char* name, auto kern, int bm, int bn) {
            dim3 grid((N + bn - 1) / bn, (M + bm - 1) / bm);
            hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
            (void)hipStreamSynchronize(s);
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a);
            (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("%-28s %9.3f us  %8.1f TFLOP/s\n", name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        };
        auto rung = [&](const char* name, auto go) {
            go(a, s); (void)hipStreamSynchronize(s);
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < reps; ++r) go(a, s);
            (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
            printf("%-28s %9.3f us  %8.1f TFLOP/s\n", name, ms * 1e3, 2.0 * M * N * K / (ms * 1e-3) / 1e12);
        };
        {   // big tile vs the register-staged kernel on random operands: must agree bit for bit (same MFMA chain per element)
            std::vector<uint16_t> ha((size_t)M * K), hw((size_t)N * K);
            uint32_t x = 777;
            auto rnd = [&]() { x = x * 1664525u + 1013904223u; return x >> 16; };
            for (auto& v : ha) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 128.f);
            for (auto& v : hw) v = f_to_bf16_host((float)((int)(rnd() & 0xff) - 128) / 1024.f);
            (void)hipMemcpy(A, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); (void)hipMemcpy(W, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
            void* Y2; (void)hipMalloc(&Y2, (size_t)M * N * 2);
            glds_go<64, 2>(a, s);
            GemmArgs b = a; b.Y = Y2;
            big_go(b, s);
            (void)hipStreamSynchronize(s);
            std::vector<uint16_t> y1((size_t)M * N), y2((size_t)M * N);
            (void)hipMemcpy(y1.data(), Y, y1.size() * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(y2.data(), Y2, y2.size() * 2, hipMemcpyDeviceToHost);
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < y1.size(); ++i) if (y1[i] != y2[i]) { if (!bad) first = i; ++bad; }
            printf("big 256x256 vs glds 128x64 on random operands: %zu / %zu elements differ%s\n", bad, y1.size(), bad ? "  <-- MISMATCH" : " (bit-identical)");
            if (bad) printf("  first at row %zu col %zu: %04x vs %04x\n", first / N, first % N, y1[first], y2[first]);
            {   // the 256 x 128 variant on the same operands
                big_go_t<true, 128>(b, s);
                (void)hipStreamSynchronize(s);
                (void)hipMemcpy(y2.data(), Y2, y2.size() * 2, hipMemcpyDeviceToHost);
                size_t bad2 = 0;
                for (size_t i = 0; i < y1.size(); ++i) bad2 += y1[i] != y2[i];
                printf("big 256x128 vs glds 128x64 on random operands: %zu / %zu elements differ%s\n", bad2, y1.size(), bad2 ? "  <-- MISMATCH" : " (bit-identical)");
            }
            (void)hipFree(Y2);
            rung("big 256x128 whole lines (random)", big_go_t<true, 128>);
            rung("big 256x256 whole lines (random)", big_go<bf16_t>);
            a.big_pair = -1;
            rung("big 256x128 ring-4 half lines (random)", big_go_t<true, 128>);
            rung("big 256x256 ring-4 half lines (random)", big_go<bf16_t>);
            {   // the two copy forms of the big tile against each other
                GemmArgs c2 = a; void* Y3; (void)hipMalloc(&Y3, (size_t)M * N * 2); c2.Y = Y3;
                big_go(c2, s); c2.big_pair = 0; c2.Y = Y; big_go(c2, s); (void)hipStreamSynchronize(s);
                std::vector<uint16_t> q1((size_t)M * N), q2((size_t)M * N);
                (void)hipMemcpy(q1.data(), Y, q1.size() * 2, hipMemcpyDeviceToHost); (void)hipMemcpy(q2.data(), Y3, q2.size() * 2, hipMemcpyDeviceToHost);
                size_t bd = 0; for (size_t i = 0; i < q1.size(); ++i) bd += q1[i] != q2[i];
                printf("big 256x256 whole lines vs half lines: %zu elements differ%s\n", bd, bd ? "  <-- MISMATCH" : " (bit-identical)");
                (void)hipFree(Y3);
            }
            a.big_pair = 0;
            rung("glds 128x64 2 st (random)", glds_go<64, 2>);
            (void)hipMemset(A, 0x3c, (size_t)M * K * 2); (void)hipMemset(W, 0x3c, (size_t)N * K * 2);
        }
        rung("big 256x256 whole lines", big_go<bf16_t>);
        a.big_pair = -1;
        rung("big 256x256 ring-4 half lines", big_go<bf16_t>);
        a.big_pair = 0;
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
        (void)hipMalloc(&A, na * 2); (void)hipMalloc(&W, nw * 2); (void)hipMalloc(&Y, ny * 2);
        (void)hipMemcpy(A, ha.data(), na * 2, hipMemcpyHostToDevice); (void)hipMemcpy(W, hw.data(), nw * 2, hipMemcpyHostToDevice);
        GemmArgs a{};
        a.A = A; a.lda = sh.Cin; a.M = sh.M; a.a_rows = sh.M; a.n_taps = sh.taps; a.Cin = sh.Cin; a.W = W; a.N = sh.N;
        for (int i = 0; i < sh.taps; ++i) a.tap_off[i] = -(sh.taps - 1 - i) * sh.dil;
        a.bias_mod = sh.N; a.Y = Y; a.ldy = sh.N;
        void* ws = nullptr;
        if (sh.ws) { (void)hipMalloc(&ws, (size_t)(8 << 20) * 4); a.ws = (float*)ws; a.ws_floats = 8 << 20; }
        // the half-line ring of four first (where the big tile is chosen at all), its output kept for the comparison
        float ms_half = 0; std::vector<uint16_t> hy_half(ny);
        {
            GemmArgs h = a; h.big_pair = -1;
            gemm_launch<bf16_t>(h, s); (void)hipStreamSynchronize(s);
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < reps; ++r) gemm_launch<bf16_t>(h, s);
            (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms_half, e0, e1); ms_half /= reps;
            (void)hipMemcpy(hy_half.data(), Y, ny * 2, hipMemcpyDeviceToHost);
            (void)hipMemset(Y, 0, ny * 2);
        }
        gemm_launch<bf16_t>(a, s);
        (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s);
        for (int r = 0; r < reps; ++r) gemm_launch<bf16_t>(a, s);
        (void)hipEventRecord(e1, s);
        (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
        // spot check of 16 outputs against a host dot product
        std::vector<uint16_t> hy(ny);
        (void)hipMemcpy(hy.data(), Y, ny * 2, hipMemcpyDeviceToHost);
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
        float ms_g = 0; size_t ndg = 0;
        {   // round 5's choices: the plain blockIdx order of the (N tiles x M tiles) grids, the LDS-DMA tile by four waves whatever the grid
            GemmArgs h = a; h.xcd_map = -1; h.glds_waves = -1; h.chain = -1;
            (void)hipMemset(Y, 0, ny * 2);
            gemm_launch<bf16_t>(h, s); (void)hipStreamSynchronize(s);
            (void)hipEventRecord(e0, s);
            for (int r = 0; r < reps; ++r) gemm_launch<bf16_t>(h, s);
            (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms_g, e0, e1); ms_g /= reps;
            std::vector<uint16_t> hg(ny); (void)hipMemcpy(hg.data(), Y, ny * 2, hipMemcpyDeviceToHost);
            if (!sh.ws) for (size_t i = 0; i < ny; ++i) ndg += hy[i] != hg[i];
        }
        if (sh.M <= 2100 && !sh.ws && chain_ok(a)) {   // the chain kernel on the few-row shapes (taps and dilations included): must agree bit for bit
            for (int form = 0; form < 2; ++form) {
                (void)hipMemset(Y, 0, ny * 2);
                auto go = [&]() { if (form) chain_go<16, 4>(a, s); else chain_go<8, 8>(a, s); };
                go(); (void)hipStreamSynchronize(s);
                (void)hipEventRecord(e0, s);
                for (int r = 0; r < reps; ++r) go();
                (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
                float ms_c; (void)hipEventElapsedTime(&ms_c, e0, e1); ms_c /= reps;
                std::vector<uint16_t> hc(ny); (void)hipMemcpy(hc.data(), Y, ny * 2, hipMemcpyDeviceToHost);
                size_t ndc = 0; for (size_t i = 0; i < ny; ++i) ndc += hy[i] != hc[i];
                printf("%-24s    chain kernel, %s: %9.3f us (%+.1f %%), outputs %s\n", "", form ? "16 waves x 4 steps" : " 8 waves x 8 steps", ms_c * 1e3,
                       100.0 * (ms - ms_c) / ms, ndc ? "DIFFER <-- MISMATCH" : "bit-identical");
            }
        }
        size_t nd = 0; for (size_t i = 0; i < ny; ++i) nd += hy[i] != hy_half[i];
        printf("%-24s M=%7d N=%5d K=%5zu  %9.3f us  %8.1f TFLOP/s  check %.2e %s | big tile on half lines: %9.3f us (%+.1f %%), outputs %s\n", sh.name, sh.M, sh.N, K, ms * 1e3,
               fl / (ms * 1e-3) / 1e12, maxerr, maxerr < 2e-2 ? "ok" : "MISMATCH", ms_half * 1e3, 100.0 * (ms_half - ms) / ms_half, nd ? "DIFFER <-- MISMATCH" : "bit-identical");
        printf("%-24s    round 5's choices (see source) : %9.3f us (%+.1f %%), outputs %s\n", "", ms_g * 1e3, 100.0 * (ms_g - ms) / ms_g, ndg ? "DIFFER" : "bit-identical");
        (void)hipFree(A); (void)hipFree(W); (void)hipFree(Y); if (ws) (void)hipFree(ws);
    }
    return 0;
}
