// Measurement tool (not product code): STEADY-STATE rate at which one workgroup per CU takes delivery of operand tiles out of the L2,
// by the way the bytes travel:
//   mode 0  registers          global_load_dwordx4 -> VGPRs (xor-reduced)
//   mode 1  registers + LDS    global_load_dwordx4 -> VGPRs -> ds_write_b128 (the register-prefetch tiles' path)
//   mode 2  LDS-DMA            global_load_lds_dwordx4 (the LDS-DMA tiles' path: glds_gemm_kernel, big_gemm_kernel)
// Every workgroup walks the same L2-resident window (`window_kb`, shared by all workgroups: the tiles of a GEMM whose operands fit the
// L2) in steps of `step_kb` (one "K step" of a tile), `iters` steps per launch (long enough that the launch cost vanishes), with
// DEPTH steps in flight.  Access pattern: 8 lanes per 128-byte row (whole cache lines), rows `row_stride` bytes apart -- a K step of 64
// bf16 columns of a row-major operand; with --half: 4 lanes per 64-byte half row, 16 rows per wave instruction (a K step of 32).
// Workgroups of 4 / 8 / 16 waves, one to four per CU.  Result on MI355X (profiles/r06_dma_rate.txt): ~3.6 B/clk per WAVE whatever the path
// and the depth; a CU saturates near 80-90 GB/s from 16 waves on.
// usage: dma_rate_bench [iters = 4000] [row_stride = 4096] [window_kb = 1024]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr;
typedef const __attribute__((address_space(1))) void* gbl_ptr;

// NP = 16-byte pieces per thread and step; DEPTH steps in flight (modes 0 / 1: register sets; mode 2: LDS stages)
template <int MODE, int NT, int NP, int DEPTH, bool HALF>
__global__ __launch_bounds__(NT) void pull_kernel(const unsigned char* base, int window_rows, int row_stride, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(128))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int LPR = HALF ? 4 : 8;                    // lanes per row piece
    constexpr int RB = HALF ? 64 : 128;                  // bytes per row piece
    constexpr int STEP_BYTES = NT * NP * 16;
    constexpr int ROWS_PER_STEP = STEP_BYTES / RB;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 v[DEPTH][NP];
    auto src = [&](int step, int p) -> const unsigned char* {
        const int slot = p * NT + tid, r = slot / LPR, c = slot % LPR;
        // step s reads rows [s * ROWS_PER_STEP ...) of the window, wrapping; HALF: the two halves of a line are consecutive steps' columns
        const int row = (int)(((long)(HALF ? step >> 1 : step) * ROWS_PER_STEP + r) % window_rows);
        return base + (size_t)row * row_stride + (HALF ? (step & 1) * 64 : 0) + c * 16;
    };
    auto issue = [&](int step, int d) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            if constexpr (MODE == 2) {
                __builtin_amdgcn_global_load_lds((gbl_ptr)src(step, p), (lds_ptr)(smem + d * STEP_BYTES + (p * NT + wave * 64) * 16), 16, 0, 0);
            } else v[d][p] = *reinterpret_cast<const u32x4*>(src(step, p));
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH - 1; ++d) issue(d, d);
    for (int s0 = 0; s0 < iters; s0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int s = s0 + d;
            issue(s + DEPTH - 1, (d + DEPTH - 1) % DEPTH);
            // wait for step s: (DEPTH - 1) * NP loads may stay in flight
            if constexpr (MODE == 2) {
                constexpr int n = (DEPTH - 1) * NP;
                __builtin_amdgcn_s_waitcnt((n & 15) | (7 << 4) | (15 << 8) | ((n >> 4) << 14));
                __builtin_amdgcn_s_barrier();
            } else if constexpr (MODE == 1) {
#pragma unroll
                for (int p = 0; p < NP; ++p) *reinterpret_cast<u32x4*>(smem + (d & 1) * STEP_BYTES + (p * NT + tid) * 16) = v[d][p];
                __builtin_amdgcn_s_barrier();
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p) acc ^= v[d][p];
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();
    if (MODE != 0) acc.x = *reinterpret_cast<unsigned*>(smem + tid * 4);
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[tid] = acc.x;
}

template <int MODE, int NT, int NP, int DEPTH, bool HALF>
static void run(const char* name, unsigned char* buf, int window_kb, int row_stride, int iters, unsigned* sink, hipStream_t st, int wgs) {
    constexpr int STEP_BYTES = NT * NP * 16;
    const int window_rows = window_kb * 1024 / (HALF ? 128 : 128);
    const size_t shm = MODE == 2 ? (size_t)DEPTH * STEP_BYTES : (MODE == 1 ? 2 * STEP_BYTES : 1024);
    auto kern = pull_kernel<MODE, NT, NP, DEPTH, HALF>;
    CHK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const size_t shm_fill = wgs <= 256 ? 100 * 1024 : (wgs <= 512 ? 60 * 1024 : 36 * 1024);   // pad the LDS request: exactly wgs / 256 workgroups per CU
    const size_t shm_use = shm > shm_fill ? shm : shm_fill;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), shm_use, st, buf, window_rows, row_stride, iters, sink);
    CHK(hipStreamSynchronize(st));
    CHK(hipEventRecord(e0, st));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), shm_use, st, buf, window_rows, row_stride, iters, sink);
    CHK(hipEventRecord(e1, st)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double us_step = 1e3 * ms / iters, gbs = STEP_BYTES / (us_step * 1e3);
    printf("%-34s %3d threads x %d pieces, depth %d, %s, %d wgs: %6.3f us per %2d-KB step -> %6.1f GB/s per workgroup = %5.1f B/clk @2.4GHz  [%5.2f TB/s aggregate]\n", name, NT, NP,
           DEPTH, HALF ? "half lines " : "whole lines", wgs, us_step, STEP_BYTES / 1024, gbs, gbs / 2.4, gbs * wgs * 1e-3);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 4000;
    const int row_stride = argc > 2 ? atoi(argv[2]) : 4096;
    const int window_kb = argc > 3 ? atoi(argv[3]) : 1024;          // bytes actually touched: window_kb (rows of 128 B)
    hipStream_t st; CHK(hipStreamCreate(&st));
    const size_t buf_bytes = (size_t)window_kb * 1024 / 128 * row_stride + (1 << 20);
    unsigned char* buf; CHK(hipMalloc(&buf, buf_bytes)); CHK(hipMemset(buf, 1, buf_bytes));
    unsigned* sink; CHK(hipMalloc(&sink, 4096));
    printf("# window %d KB of whole lines (row stride %d B) shared by all workgroups, %d steps per launch\n", window_kb, row_stride, iters);
    // waves per CU: how far does the per-CU rate scale?  (LDS request of a workgroup sized so that exactly wgs / 256 fit a CU)
    for (int wgs : {256, 512, 1024}) {
        run<0, 512, 4, 4, false>("registers", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<2, 512, 4, 4, false>("LDS-DMA (global_load_lds x4)", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<0, 1024, 2, 4, false>("registers", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<2, 1024, 2, 4, false>("LDS-DMA (global_load_lds x4)", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<0, 1024, 4, 2, false>("registers", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<2, 1024, 4, 2, false>("LDS-DMA (global_load_lds x4)", buf, window_kb, row_stride, iters, sink, st, wgs);
    }
    for (int wgs : {256, 512}) {
        run<0, 256, 6, 2, false>("registers", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<0, 256, 6, 4, false>("registers", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<1, 256, 6, 2, false>("registers + ds_write", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<1, 256, 6, 4, false>("registers + ds_write", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<2, 256, 6, 2, false>("LDS-DMA (global_load_lds x4)", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<2, 256, 6, 4, false>("LDS-DMA (global_load_lds x4)", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<0, 256, 6, 4, true>("registers", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<1, 256, 6, 4, true>("registers + ds_write", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<2, 256, 6, 4, true>("LDS-DMA (global_load_lds x4)", buf, window_kb, row_stride, iters, sink, st, wgs);
        run<0, 512, 4, 4, false>("registers", buf, window_kb, row_stride, iters, sink, st, wgs / 2 < 256 ? 256 : wgs / 2);
        run<1, 512, 4, 4, false>("registers + ds_write", buf, window_kb, row_stride, iters, sink, st, wgs / 2 < 256 ? 256 : wgs / 2);
        run<2, 512, 4, 4, false>("LDS-DMA (global_load_lds x4)", buf, window_kb, row_stride, iters, sink, st, wgs / 2 < 256 ? 256 : wgs / 2);
    }
    return 0;
}
