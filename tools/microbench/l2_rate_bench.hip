// Measurement tool (not product code): how fast can ONE workgroup per CU (512 threads, as skinny_gemm_kernel) pull bytes into registers,
// by access pattern and by where the bytes live?  Every wave issues NL 16-byte-per-lane loads back to back (all in flight), xors them
// and repeats ROUNDS times over its share; total bytes per workgroup = 8 waves x ROUNDS x NL x 1 KB.
//   pattern 0  contiguous      one instruction = 1 KB contiguous                                   (the M = 1 GEMV's weight rows)
//   pattern 1  token rows      one instruction = 4 rows x 256 B, row stride 2 KB                   (skinny_gemm_kernel's token units)
//   pattern 2  operand rows    one instruction = 16 rows x 64 B, row stride 2 KB (K = 1024)        (skinny_gemm_kernel's weight fragments)
//   pattern 3  operand rows, row stride 6 KB (K = 3072)
//   pattern 4  8 rows x 128 B (whole cache lines; a K step of 64)      pattern 5  16 rows x 64 B, both halves of a line by the same wave back to back
//   source  0  shared: every workgroup reads the SAME region (tokens: L2 hits after the first touch per XCD)
//           1  private: every workgroup its own region of a buffer much larger than the caches, rotating (weights: HBM)
// Prints GB/s per CU and TB/s aggregate.   usage: l2_rate_bench [KB per workgroup = 128] [NL = 8] [reps = 50]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int PAT, int NL>
__global__ __launch_bounds__(512) void pull_kernel(const unsigned char* base, size_t wg_stride, int rounds, unsigned* sink) {
    extern __shared__ unsigned char smem[];          // sized by the launch so that one workgroup fills a CU
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char* p = base + (size_t)blockIdx.x * wg_stride;
    u32x4 acc = {0, 0, 0, 0};
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int q = (r * NL + i) * 8 + wave;                       // the q-th 1-KB piece of the workgroup's share
            size_t off;
            if (PAT == 0) off = (size_t)q * 1024 + lane * 16;
            else if (PAT == 1) off = (size_t)(q >> 3) * (4 * 2048) + (size_t)(lane >> 4) * 2048 + (q & 7) * 256 + (lane & 15) * 16;     // 4 rows x 256 B
            else if (PAT == 2) off = (size_t)(q >> 5) * (16 * 2048) + (size_t)(lane & 15) * 2048 + (q & 31) * 64 + (lane >> 4) * 16;  // 16 rows x 64 B
            else if (PAT == 3) off = (size_t)(q / 96) * (16 * 6144) + (size_t)(lane & 15) * 6144 + (q % 96) * 64 + (lane >> 4) * 16;
            else if (PAT == 4) off = (size_t)(q >> 4) * (8 * 2048) + (size_t)(lane >> 3) * 2048 + (q & 15) * 128 + (lane & 7) * 16;   // 8 rows x 128 B (whole lines)
            else {  // PAT 5: 16 rows x 64 B, the SAME wave takes both halves of a line in consecutive instructions
                const int pr = r * NL + i, qq = (pr >> 1) * 8 + wave;                     // the pair's index; i even / odd = first / second half
                off = (size_t)(qq >> 4) * (16 * 2048) + (size_t)(lane & 15) * 2048 + (qq & 15) * 128 + (pr & 1) * 64 + (lane >> 4) * 16;
            }
            v[i] = *reinterpret_cast<const u32x4*>(p + off);
        }
#pragma unroll
        for (int i = 0; i < NL; ++i) acc ^= v[i];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc.x;      // (never true: keeps the loads)
    if (threadIdx.x == 0) smem[0] = 0;
}

template <int PAT, int NL>
static void run(const char* name, int src, int kb, int reps, unsigned char* buf, size_t buf_bytes, unsigned* sink, hipStream_t st) {
    const int rounds = kb / (8 * NL);
    const size_t share = (size_t)kb * 1024 * 2;                          // (patterns 1-3 span up to 2 x their bytes)
    const size_t wg_stride = src ? share : 0;
    const int nrot = src ? (int)(buf_bytes / (share * 256)) : 1;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pull_kernel<PAT, NL>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((pull_kernel<PAT, NL>), dim3(256), dim3(512), 150 * 1024, st, buf, wg_stride, rounds, sink);
    CHK(hipStreamSynchronize(st));
    // a chain of dependent launches inside a graph, like the product
    hipGraph_t g; hipGraphExec_t ge;
    CHK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 40; ++i) hipLaunchKernelGGL((pull_kernel<PAT, NL>), dim3(256), dim3(512), 150 * 1024, st, buf + (size_t)(i % nrot) * share * 256, wg_stride, rounds, sink);
    CHK(hipStreamEndCapture(st, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CHK(hipGraphLaunch(ge, st)); CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int r = 0; r < reps; ++r) CHK(hipGraphLaunch(ge, st));
    CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps / 40;
    printf("%-16s %-8s %4d KB per workgroup, %2d loads in flight per wave: %7.2f us per launch -> %6.1f GB/s per CU incl. the launch, %6.1f GB/s over (launch - 1.5 us)  [%5.2f TB/s]\n",
           name, src ? "private" : "shared", kb, NL, us, kb * 1.024 / us, kb * 1.024 / (us - 1.5), 256.0 * kb * 1.024e-3 / (us - 1.5));
    CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
}

int main(int argc, char** argv) {
    const int kb = argc > 1 ? atoi(argv[1]) : 128;
    const int reps = argc > 3 ? atoi(argv[3]) : 50;
    hipStream_t st; CHK(hipStreamCreate(&st));
    const size_t buf_bytes = (size_t)3 << 30;
    unsigned char* buf; CHK(hipMalloc(&buf, buf_bytes)); CHK(hipMemset(buf, 1, buf_bytes));
    unsigned* sink; CHK(hipMalloc(&sink, 4096));
    for (int src = 0; src < 2; ++src) {
        run<0, 8>("contiguous", src, kb, reps, buf, buf_bytes, sink, st);
        run<1, 8>("token rows", src, kb, reps, buf, buf_bytes, sink, st);
        run<2, 8>("operand rows 2K", src, kb, reps, buf, buf_bytes, sink, st);
        run<3, 8>("operand rows 6K", src, kb, reps, buf, buf_bytes, sink, st);
        run<4, 8>("8 rows x 128 B", src, kb, reps, buf, buf_bytes, sink, st);
        run<5, 8>("16 x 64 B paired", src, kb, reps, buf, buf_bytes, sink, st);
        run<0, 16>("contiguous", src, kb, reps, buf, buf_bytes, sink, st);
        run<1, 16>("token rows", src, kb, reps, buf, buf_bytes, sink, st);
        run<2, 16>("operand rows 2K", src, kb, reps, buf, buf_bytes, sink, st);
    }
    return 0;
}
