// Weight-stationary GEMM for FEW rows (the short-prompt prefill: M <= kSkinnyMaxRows tokens against a [N][K] weight), bf16.
//
// The tiled kernels of codec_kernels.cuh need M x N / (64 x 64) workgroups to fill the chip and walk K behind one barrier per
// 32-wide step; at M = 200 that is a latency-bound K loop on a quarter of the matrix cores' issue slots (16-33 us per GEMM,
// profiles/r03_gemm_bench.txt).  Here the roles are those of the batch-decode GEMV (batch_kernels.cuh): one workgroup owns
// 16 * RB WEIGHT rows, its 8 waves split K, and every wave keeps its K share of those rows in registers for the whole launch
// (v_mfma_f32_16x16x32_bf16: A = weights, lane (row = lane & 15, k group = lane >> 4) -- a 16-byte global load per lane IS the
// operand layout).  The tokens stream past in units of 16 rows x 1024 columns; a wave handles ITS 128 columns of a unit alone:
// coalesced 16-byte loads (a quarter-wave reads 256 contiguous bytes of one token row) into staging registers, a unit later a
// swizzled ds_write into the wave's own 4 KB LDS slot, a unit later b128 reads in B-operand layout, a unit later the MFMAs -- no
// barrier anywhere on that path, the compiler's vmcnt / lgkmcnt bookkeeping is exact (straight-line, branch-free loads).
// Only the K split meets across waves: per group of two token tiles the 8 partial 16 x 16 products per row block go through LDS
// (double-buffered: one barrier per group) and each (tile, row block) epilogue runs on a different wave while the others go on.
// Each weight byte is fetched from HBM once (mt = 1); the token matrix is re-read from L2 by every workgroup, N / (16 RB) times
// in all -- that L2 -> CU stream (~30 B/clk per CU measured) is what bounds a unit, hence RB up to 3 where registers allow.
// What did NOT work, measured (profiles/r03_skinny_gemm.txt): B fragments loaded straight from global memory in operand layout
// (64 separate 16-byte requests per instruction: 14 B/clk per CU); the LDS-DMA path (global_load_lds) in a shared or per-wave
// ring (same ~25 B/clk, SQ_LDS_DATA_FIFO_FULL 11 % of the time, profiles/r03_pmc_skinny_dma_*.txt); one epilogue wave per tile
// (the others wait for it at the next barrier: +40 % on the SwiGLU variant).
//
//   SK_STORE      y = rnd(acc)                                   (qkv)
//   SK_RESIDUAL   y = rnd(rnd(acc) + res)                        (o_proj, down: the epilogue order of the tiled kernels)
//   SK_SWIGLU     [gate | up] weight halves: a row block is gate rows j0 .. j0+7 and up rows I+j0 .. I+j0+7, so both factors of
//                 an output sit in the same 16 x 16 tile: y = rnd(rnd(silu(rnd(g))) * rnd(u)) -- the values the two-launch form
//                 (GEMM, then silu_mul_kernel on the stored bf16 halves) produces -- and the [M][2I] image is never written.
// The accumulation order differs from the tiled kernels (8 partial sums per element, two chains each), so this path is for
// callers that lend a workspace (GemmArgs::ws != null marks "order-free"): the prefill.  The codec never takes it (a tail decode
// must stay bit-identical to a full decode, which runs on the tiled kernels).
#pragma once
#include "fq3_common.cuh"

namespace fq3 {

enum { SK_STORE = 0, SK_RESIDUAL = 1, SK_SWIGLU = 2 };
constexpr int kSkinnyMaxRows = 416;

struct SkinnyArgs {
    const bf16_t* X; int ldx; int M;        // tokens [M][K]
    const bf16_t* W; int N;                 // weight [N][K]; SK_SWIGLU: N = 2 I
    const bf16_t* res; int ldr;             // SK_RESIDUAL
    bf16_t* Y; int ldy;                     // [M][N] (SK_SWIGLU: [M][I])
    int nrb;                                // weight-row groups: N / (16 RB)
    int mt;                                 // workgroups per row group; workgroup (rb, j) takes token tiles j, j + mt, ...
    int no_stagger;                         // measurement switch: 1 = every workgroup walks the token tiles from tile 0 (see `rot` below)
    // FRAGMENT-MAJOR copy of W (round 6), or null: P[row block][K / 32 steps][64 lanes][8] -- lane (k group fq, row fr) of step t holds
    // W[row(block, fr)][32 t + 8 fq .. + 8], i.e. the 1 KB an A-operand load of one wave reads is CONTIGUOUS.  From the row-major
    // matrix the same load is 16 rows x 64 B -- half cache lines of 16 different rows: measured 40 GB/s per CU against 125-135 for
    // contiguous kilobytes out of the L2, and 4.3-5.5 against 5.2-6.5 TB/s out of HBM (tools/microbench/l2_rate_bench.hip,
    // profiles/r06_l2_rate_by_pattern.txt).  Rows of a block: 16 b + fr, or for SK_SWIGLU the 8 gate + 8 up rows the epilogue pairs.
    // Same values into the same registers: bit-identical.  skinny_pack() builds it.
    const bf16_t* Wp;
#ifdef FQ3_SK_TRACE
    unsigned long long* trace;              // measurement builds only (tools/microbench/skinny_trace.hip): [workgroup][wave][8] time stamps
#endif
    // ---- RMSNorm folded into the GEMM pair (round 5; the lock-step batch above 32 lanes) ----
    // producer side (SK_RESIDUAL): per token and 16-column block of Y the sum of squares of the STORED (rounded) values:
    // ssq_out[token * ssq_ld + column / 16]; null = not wanted
    float* ssq_out; int ssq_ld;
    // consumer side (the NORM instantiations of SK_STORE / SK_SWIGLU; X is the raw residual stream): rs = rsqrt(sum / K + eps) from the
    // K / 16 partials of a row summed in one fixed order, x_n = rnd(rnd(x * rs) * gain) applied on the way from the staging registers
    // to LDS -- the two roundings of the reference's norm, so only the ORDER of the sum of squares differs from rmsnorm_batch_kernel
    const float* ssq; const bf16_t* gain; float eps;
};

// The 27 kernel instantiations live in ONE translation unit of the library (fq3_prefill.hip defines FQ3_SKINNY_DEFINE); the others
// (FQ3_SKINNY_EXTERN: codec, batch, prompt, reference-audio TUs) see declarations only and call through skinny_launch_epi.  A
// standalone tool that includes this header with neither macro gets everything inline.
// K values served (the talker's hidden / q / intermediate widths at 0.6B and 1.7B)
inline bool skinny_k_ok(int K) { return K == 1024 || K == 2048 || K == 3072 || K == 6144; }
// the normalising form (SkinnyArgs::ssq): K = hidden of the 0.6B / 1.7B stacks, every row's 1 / rms in an LDS table
inline bool skinny_norm_ok(int K, int M) { return (K == 1024 || K == 2048) && M >= 1 && M <= 256; }

// shapes the fragment-major copy serves (the kernels' own: K a multiple of 1024 the kernels are built for, whole 16-row blocks)
constexpr int kSkMaxDevices = 16;           // devices a process may drive (the LDS limit of a kernel is a per-device setting)
inline bool skinny_pack_ok(int N, int K) { return skinny_k_ok(K) && N % 16 == 0 && N > 0; }

#ifdef FQ3_SKINNY_EXTERN
void skinny_pack_ext(const bf16_t* W, bf16_t* P, int N, int K, int swiglu_I, hipStream_t s);
inline void skinny_pack(const bf16_t* W, bf16_t* P, int N, int K, int swiglu_I, hipStream_t s) { skinny_pack_ext(W, P, N, K, swiglu_I, s); }
void skinny_launch_epi(int epi, const SkinnyArgs& a, int K, hipStream_t s, int rb_force);
bool skinny_prepare_epi(int epi);
template <int EPI> inline void skinny_launch(const SkinnyArgs& a, int K, hipStream_t s, int rb_force = 0) { skinny_launch_epi(EPI, a, K, s, rb_force); }
template <int EPI> inline bool skinny_prepare() { return skinny_prepare_epi(EPI); }
#else
typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
constexpr int kSkKC = 1024;                 // columns of one unit (all 8 waves); a wave's share: 128 columns = 256 B per token row
constexpr int kSkNW = 8;
constexpr int kSkSlotB = 16 * 256;          // one wave's slot: 16 token rows x 256 B
constexpr int kSkT = 2;                     // token tiles per reduction group
constexpr size_t skinny_lds_bytes(int RB) { return (size_t)kSkNW * 2 * kSkSlotB + (size_t)2 * kSkT * kSkNW * RB * 1024; }

constexpr int kSkNormMaxRows = 256;         // NORM instantiations: rows whose 1 / rms fit the LDS table (two passes of 128 rows, 4 threads per row)
template <int K, int RB, int EPI, bool NORM = false>
__global__ __launch_bounds__(512) void skinny_gemm_kernel(SkinnyArgs a) {
    typedef bf16_t T_;
    static_assert(!NORM || EPI != SK_RESIDUAL, "the normalising form feeds qkv / gate | up / heads");
    constexpr int NW = kSkNW, KC = kSkKC, NCH = K / KC, KS = KC / NW / 32, T = kSkT;   // KS = 4 MFMA steps per wave and unit
    static_assert(K % KC == 0 && (T * NCH) % 2 == 0, "shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char sk_smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fq = lane >> 4;
#ifdef FQ3_SK_TRACE
    unsigned long long sk_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long sk_w0 = wall_clock64();
#define FQ3_SK_STAMP(i) sk_t[i] = clock64()
#else
#define FQ3_SK_STAMP(i) do {} while (0)
#endif
    FQ3_SK_STAMP(0);
    unsigned char* ring = sk_smem + (size_t)wave * (2 * kSkSlotB);              // this wave's [2][16 rows][256 B]
    float* red = reinterpret_cast<float*>(sk_smem + (size_t)NW * 2 * kSkSlotB); // [2][T][NW][RB][64 lanes][4]
    constexpr int kRedTile = NW * RB * 256, kRedBuf = T * kRedTile;             // floats
    // workgroups that share a row group (mt > 1) sit on the same XCD (consecutive workgroup ids go round the 8 XCDs), so the group's
    // weight rows cross the fabric once per XCD L2
    int rb, mj;
    {
        const int b = blockIdx.x;
        if (a.nrb % 8 == 0) { const int q = b >> 3; mj = q % a.mt; rb = (q / a.mt) * 8 + (b & 7); }
        else { rb = b / a.mt; mj = b % a.mt; }
    }
    const int ntiles = (a.M + 15) >> 4;
    const int nmine = mj < ntiles ? (ntiles - mj + a.mt - 1) / a.mt : 0;
    if (nmine == 0) return;
    const int ngroups = (nmine + T - 1) / T;
    const int nunits = nmine * NCH;
    // every workgroup reads the same token matrix: workgroup q of an XCD starts its walk q tiles further on (no measurable effect,
    // kept: it costs nothing)
    const int rot = a.no_stagger ? 0 : (int)((blockIdx.x >> 3) % (unsigned)nmine);
    auto tile_t0 = [&](int i) { int j = i + rot; j = j >= nmine ? j - nmine : j; return (mj + j * a.mt) << 4; };

    // Unit v = (this workgroup's tile v / NCH, column chunk v % NCH): this wave's share is 16 token rows x 256 B.  Load p of a unit
    // moves rows 4p .. 4p+3 (lane: row 4p + lane / 16, 16-byte chunk lane % 16: every quarter-wave reads 256 contiguous bytes) into
    // staging registers; a unit later they go to the wave's LDS slot v % 2 with the chunk index XORed by the row (row stride 256 B =
    // all 64 banks; the swizzle spreads the 16 rows of a fragment read over the banks), and come back as B-operand fragments.
    // Units past the end repeat the last one (uniform control flow; nobody uses them).
    const int ld_r = lane >> 4, ld_c = lane & 15;
    // byte offsets in 32 bits off the uniform base (no 64-bit multiplies in the loop): rows past M clamp to row M - 1 by a min on the
    // offset (offsets grow with the row)
    const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.X);
    const uint32_t rowb = (uint32_t)a.ldx * 2u;
    const uint32_t lane_off = (uint32_t)ld_r * rowb + (uint32_t)wave * (KC / NW * 2) + (uint32_t)ld_c * 16u;
    const uint32_t max_off = (uint32_t)(a.M - 1) * rowb + (uint32_t)wave * (KC / NW * 2) + (uint32_t)ld_c * 16u;
    auto load_unit = [&](u32x4 (&st)[4], int v) {
        v = v < nunits - 1 ? v : nunits - 1;
        const int i = v / NCH, kc = v - i * NCH;
        const uint32_t tile_off = (uint32_t)tile_t0(i) * rowb, kc_off = (uint32_t)kc * (KC * 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            uint32_t off = tile_off + (uint32_t)(4 * p) * rowb + lane_off;
            off = off < max_off ? off : max_off;
            st[p] = *reinterpret_cast<const u32x4*>(xb + (size_t)(off + kc_off));
        }
    };
    const int wr_off = ld_r * 256 + ((ld_c ^ ld_r) & 15) * 16;                  // + p * 1024 + ((4p) swizzle): row 4p + ld_r, chunk c ^ row
    auto write_unit = [&](const u32x4 (&st)[4], int v, int kcs, const float (*gwp)[8], const float (&rt)[4]) {
        unsigned char* slot = ring + (v & 1) * kSkSlotB;
        if constexpr (NORM) {
            // the unit's rows and column chunk (as load_unit resolved them); rows past M took row M - 1's data and take its 1 / rms.
            // A tile's 16 rows lie in one 64-row block of the table: one register of `rt` (uniform choice), one cross-lane read per row group.
            const int vv = v < nunits - 1 ? v : nunits - 1;
            const int i = vv / NCH;
            const int t0 = __builtin_amdgcn_readfirstlane(tile_t0(i));
            const int blk = t0 >> 6, last = a.M - 1;
            // (mask arithmetic, not a select over an array: the latter is turned into an indexed private-memory access = scratch)
            const uint32_t tvb = (__float_as_uint(rt[0]) & (blk == 0 ? ~0u : 0u)) | (__float_as_uint(rt[1]) & (blk == 1 ? ~0u : 0u)) |
                                 (__float_as_uint(rt[2]) & (blk == 2 ? ~0u : 0u)) | (__float_as_uint(rt[3]) & (blk == 3 ? ~0u : 0u));
            float rs4[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {           // (all four cross-lane reads go out before the first is needed)
                int row = t0 + 4 * p + ld_r;
                row = row < last ? row : last;
                // a cross-lane read through the LDS crossbar (no LDS memory): the index depends on the lane, so not a DPP pattern
                rs4[p] = __int_as_float(__builtin_amdgcn_ds_bpermute((row & 63) << 2, (int)tvb));
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                Raw8<T_> xr8; xr8.v = st[p];
                float x[8];
                unpack(xr8, x);
                // kcs = the unit's column chunk v % NCH, a constant at every call site after unrolling (a unit past the end is clamped to
                // the last one, whose chunk may differ: its values are never used)
                const u32x4 y = norm8_pack(x, rs4[p], gwp[kcs]);
                *reinterpret_cast<u32x4*>(slot + p * 1024 + (wr_off ^ ((4 * p) << 4))) = y;
            }
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)             // row = 4p + ld_r: chunk ^ row = (ld_c ^ ld_r) ^ 4p
                *reinterpret_cast<u32x4*>(slot + p * 1024 + (wr_off ^ ((4 * p) << 4))) = st[p];
        }
    };
    const int fro = fr * 256;
    auto read_frags = [&](sk_bf16x8 (&f)[KS], int v) {
        const unsigned char* slot = ring + (v & 1) * kSkSlotB + fro;
#pragma unroll
        for (int s = 0; s < KS; ++s) f[s] = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const u32x4*>(slot + (((s * 4 + fq) ^ fr) & 15) * 16));
    };

    // NORM: the rows' sum-of-squares partials (K / 16 per row, written by the producer's epilogue) go out FIRST -- they retire first, so
    // 1 / rms is formed while the token units and the weights are still in flight.  4 threads per row (a quarter of the partials each,
    // summed in index order), two passes of 128 rows; rows past M re-read row M - 1.
    constexpr int NPQ = K / 16 / 4 / 4;                                         // float4 loads per thread and row
    f32x4 sq[NORM ? 2 : 1][NORM ? NPQ : 1];
    u32x4 graw[NORM ? NCH : 1];
    if constexpr (NORM) {
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            int m = ps * 128 + (tid >> 2);
            m = m < a.M ? m : a.M - 1;
            const f32x4* sp = reinterpret_cast<const f32x4*>(a.ssq + (size_t)m * (K / 16) + (tid & 3) * (K / 64));
#pragma unroll
            for (int j = 0; j < NPQ; ++j) sq[ps][j] = sp[j];
        }
    }
    u32x4 stg[2][4];
    load_unit(stg[0], 0);
    load_unit(stg[1], 1);
    if constexpr (NORM) {           // this lane's 8 gain columns of every 1024-column chunk (the same columns in every unit)
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc)
            graw[kc] = *reinterpret_cast<const u32x4*>(a.gain + kc * KC + wave * (KC / NW) + ld_c * 8);
    }
    // issue order pinned (round 5): the token-side loads above, THEN the weight rows, then the first arithmetic -- left alone the
    // scheduler hoisted the first two units' LDS writes (and their full vmcnt wait: one L2 round trip) above the weight loads, i.e. the
    // HBM stream -- the long pole of the launch -- started a round trip late
    __builtin_amdgcn_sched_barrier(0);
    // this wave's K share (KS steps in every 1024-column chunk) of the 16 RB weight rows, in A-operand layout
    Raw8<T_> wreg[RB][NCH * KS];
#pragma unroll
    for (int b = 0; b < RB; ++b) {
        int wrow;
        if constexpr (EPI == SK_SWIGLU) { const int j0 = (rb * RB + b) * 8; wrow = fr < 8 ? j0 + fr : (a.N >> 1) + j0 + (fr - 8); }
        else wrow = (rb * RB + b) * 16 + fr;
        // row-major: row wrow, this wave's 128 columns of every chunk, 8 columns per k group; fragment-major: the wave's KS steps of a
        // chunk are 4 consecutive kilobytes of row block rb RB + b (uniform choice: one base and two strides)
        const T_* wp = a.Wp ? a.Wp + ((size_t)(rb * RB + b) * (K / 32) + wave * KS) * 512 + lane * 8
                            : a.W + (size_t)wrow * K + wave * (KC / NW) + fq * 8;
        const int st_kc = a.Wp ? (KC / 32) * 512 : KC, st_s = a.Wp ? 512 : 32;
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc)
#pragma unroll
            for (int s = 0; s < KS; ++s) ldraw<false>(wreg[b][kc * KS + s], wp + kc * st_kc + s * st_s);
    }
    __builtin_amdgcn_sched_barrier(0);

    // NORM: 1 / rms of every row.  The table is built cooperatively in the (still unused) partial-sum area, then every wave takes a
    // copy into registers -- lane L: rows L, 64 + L, 128 + L, 192 + L -- from which a unit's 16 values come by v_readlane (the RB = 3
    // instantiations use all 160 KB of LDS already).
    float gw[NORM ? NCH : 1][8], rtab[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (NORM) {
        float* tab = red;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
            float s4 = 0.f;
#pragma unroll
            for (int j = 0; j < NPQ; ++j) { s4 += sq[ps][j].x; s4 += sq[ps][j].y; s4 += sq[ps][j].z; s4 += sq[ps][j].w; }
            // the four quarters of a row: (q0 + q1) + (q2 + q3) in every lane of the quad (commutative pairings)
            s4 += dpp_move<kDppXor1, 0xF>(0.f, s4);
            s4 += dpp_move<kDppXor2, 0xF>(0.f, s4);
            if ((tid & 3) == 0) tab[ps * 128 + (tid >> 2)] = 1.0f / sqrtf(s4 / (float)K + a.eps);
        }
#pragma unroll
        for (int kc = 0; kc < NCH; ++kc) {
            Raw8<T_> g; g.v = graw[kc];
            unpack(g, gw[kc]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): the table is written
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int j = 0; j < 4; ++j) rtab[j] = tab[j * 64 + lane];
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();                                // nobody writes partial sums over the table before everybody has read it
    }
    // epilogue of row block b of one token tile by the calling wave: the 8 partial products at `rd`, summed in wave order
    auto epilogue = [&](const float* rd, int t0, int b) {
        const int nb = a.M - t0 < 16 ? a.M - t0 : 16;
        const int yc = (EPI == SK_SWIGLU ? (rb * RB + b) * 8 : (rb * RB + b) * 16) + fq * 4;
        uint2 rv{0u, 0u};
        if constexpr (EPI == SK_RESIDUAL) {
            const int row = t0 + (fr < nb ? fr : nb - 1);
            rv = *reinterpret_cast<const uint2*>(a.res + (size_t)row * a.ldr + yc);
        }
        const float* rp = rd + b * 256 + lane * 4;
        f32x4 t = *reinterpret_cast<const f32x4*>(rp);
#pragma unroll
        for (int w = 1; w < NW; ++w) t += *reinterpret_cast<const f32x4*>(rp + w * RB * 256);
        float v[4] = {t.x, t.y, t.z, t.w};
        bool live = fr < nb;
        if constexpr (EPI == SK_SWIGLU) {
            const float* up = rd + b * 256 + ((lane & 31) + 32) * 4;            // the up rows of gate rows fq * 4 + e sit two k groups further
            f32x4 u = *reinterpret_cast<const f32x4*>(up);
#pragma unroll
            for (int w = 1; w < NW; ++w) u += *reinterpret_cast<const f32x4*>(up + w * RB * 256);
            const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g = DT<T_>::rnd(v[e]);
                v[e] = DT<T_>::rnd(g / (1.0f + expf(-g))) * DT<T_>::rnd(uu[e]);
            }
            live = live && fq < 2;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = DT<T_>::rnd(v[e]);
            if constexpr (EPI == SK_RESIDUAL) {
                v[0] += __uint_as_float(rv.x << 16); v[1] += __uint_as_float(rv.x & 0xffff0000u);
                v[2] += __uint_as_float(rv.y << 16); v[3] += __uint_as_float(rv.y & 0xffff0000u);
            }
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]); o.y = pack_bf16x2(v[2], v[3]);
        if (live) *reinterpret_cast<uint2*>(a.Y + (size_t)(t0 + fr) * a.ldy + yc) = o;
        if constexpr (EPI == SK_RESIDUAL) {
            // the next RMSNorm's sum of squares, over the values as STORED: this lane's four columns in column order, then the four
            // column groups of the token's 16-column block (lanes fr, fr + 16, fr + 32, fr + 48) -- one partial per token and block,
            // summed in block order by the consumer (skinny_gemm_kernel<.., NORM>): deterministic, no atomics
            if (a.ssq_out) {
                const float y0 = __uint_as_float(o.x << 16), y1 = __uint_as_float(o.x & 0xffff0000u);
                const float y2 = __uint_as_float(o.y << 16), y3 = __uint_as_float(o.y & 0xffff0000u);
                float sq = y0 * y0;
                sq = fmaf(y1, y1, sq); sq = fmaf(y2, y2, sq); sq = fmaf(y3, y3, sq);
                sq = xrow_sum(sq);
                if (live && fq == 0) a.ssq_out[(size_t)(t0 + fr) * a.ssq_ld + (rb * RB + b)] = sq;
            }
        }
    };

    // Groups of T tiles: every wave walks the token units on its own (no barrier between a load and its MFMAs), leaves its partial
    // 16 x 16 products of the group's tiles in LDS, ONE barrier per group, then T waves (a different set each group) run one
    // tile's epilogue each while the others start the next group into the other partial-sum buffer.  Tiles past the end of the last
    // group recompute the last tile and are not stored.
    // The wave's pipeline, unit u multiplying: fragments of u in registers (read a unit ago), fragments of u + 1 being read from LDS,
    // unit u + 2 going from staging registers to LDS, units u + 3 and u + 4 in flight from L2; the K steps of a unit alternate
    // between two accumulators per row block (two independent MFMA chains).
    sk_bf16x8 fcur[KS], fnext[KS];
    write_unit(stg[0], 0, 0, gw, rtab);
    FQ3_SK_STAMP(1);                                                 // token unit 0 has landed
    load_unit(stg[0], 2);
    write_unit(stg[1], 1, 1 % NCH, gw, rtab);
    load_unit(stg[1], 3);
    read_frags(fcur, 0);
    for (int g = 0; g < ngroups; ++g) {
        float* rbuf = red + (g & 1) * kRedBuf;
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
            const int i = g * T + tt;
            f32x4 acc[RB][2];
#pragma unroll
            for (int b = 0; b < RB; ++b) acc[b][0] = acc[b][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < NCH; ++kc) {
                const int par = (tt * NCH + kc) & 1;                 // = u & 1 (a group has an even number of units): static after unrolling
                const int u = i * NCH + kc;
                read_frags(fnext, u + 1);                            // written a unit ago
                __builtin_amdgcn_sched_barrier(0);                   // the reads go out BEFORE this unit's MFMAs
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int b = 0; b < RB; ++b)
                        acc[b][s & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(sk_bf16x8, wreg[b][kc * KS + s].v), fcur[s], acc[b][s & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (g == 0 && tt == 0 && kc == 0) FQ3_SK_STAMP(2);   // the first unit's MFMAs are issued: its weight fragments have landed
                // unit u + 2 into the slot of unit u (its fragments are in registers, the MFMAs above have read them), then its
                // staging registers take unit u + 4
                if (par == 0) { write_unit(stg[0], u + 2, (kc + 2) % NCH, gw, rtab); load_unit(stg[0], u + 4); }
                else { write_unit(stg[1], u + 2, (kc + 2) % NCH, gw, rtab); load_unit(stg[1], u + 4); }
#pragma unroll
                for (int s = 0; s < KS; ++s) fcur[s] = fnext[s];
            }
            float* rd = rbuf + tt * kRedTile + wave * RB * 256 + lane * 4;
#pragma unroll
            for (int b = 0; b < RB; ++b) *reinterpret_cast<f32x4*>(rd + b * 256) = acc[b][0] + acc[b][1];
        }
        if (g == 0) FQ3_SK_STAMP(3);                                 // the first group's units are multiplied
        __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): my partial sums are written
        __builtin_amdgcn_s_barrier();
        if (g == 0) FQ3_SK_STAMP(4);
        // the group's T x RB (tile, row block) epilogues go to T x RB different waves (rotating with the group)
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int i = g * T + tt;
                if (i < nmine && wave == ((g * (T * RB) + tt * RB + b) % NW)) epilogue(rbuf + tt * kRedTile, tile_t0(i), b);
            }
        if (g == 0) FQ3_SK_STAMP(5);
    }
#ifdef FQ3_SK_TRACE
    FQ3_SK_STAMP(6);
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((size_t)blockIdx.x * NW + wave) * 8;
        __builtin_amdgcn_s_waitcnt(0);                               // every store of this wave has been issued AND acknowledged
        const unsigned long long w1 = wall_clock64();
        for (int i2 = 1; i2 < 7; ++i2) tr[i2] = sk_t[i2] - sk_t[0];   // core cycles since entry
        tr[0] = sk_w0; tr[7] = w1;                                   // 100 MHz wall clock at entry / exit (shared by every CU)
    }
#endif
#undef FQ3_SK_STAMP
}

// Row-major [N][K] -> fragment-major (SkinnyArgs::Wp).  swiglu_I > 0: W is [gate | up] = [2 I][K] and row block r holds gate rows
// 8 r .. 8 r + 7 and up rows I + 8 r .. (the pairing of the SK_SWIGLU epilogue); else row block r = rows 16 r .. 16 r + 15.
// One thread per 16-byte piece; one-time cost at weight-binding time.
__global__ __launch_bounds__(256) void skinny_pack_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ P, int N, int K, int swiglu_I) {
    const size_t piece = (size_t)blockIdx.x * 256 + threadIdx.x;            // = (row block * (K / 32) + step) * 64 + lane
    const size_t total = (size_t)N * K / 8;
    if (piece >= total) return;
    if (swiglu_I < 0) {
        // ROW-MAJOR copy of a [gate | up] matrix with the halves interleaved in blocks of 16 rows (gate rows 16 b .., then up rows I + 16 b ..):
        // the layout whose 256-wide ring tiles hold both factors of an output in one wave's accumulators (codec_kernels.cuh, TR epilogue)
        const int I = -swiglu_I, ppr = K / 8;
        const int orow = (int)(piece / (size_t)ppr), cp = (int)(piece % (size_t)ppr);
        const int blk = orow >> 5, w = orow & 31;
        const int srow = w < 16 ? blk * 16 + w : I + blk * 16 + (w - 16);
        *reinterpret_cast<u32x4*>(P + piece * 8) = *reinterpret_cast<const u32x4*>(W + (size_t)srow * K + cp * 8);
        return;
    }
    const int ln = (int)(piece & 63), fr = ln & 15, fq = ln >> 4;
    const size_t bs = piece >> 6;
    const int t = (int)(bs % (size_t)(K / 32)), r = (int)(bs / (size_t)(K / 32));
    const int row = swiglu_I > 0 ? (fr < 8 ? r * 8 + fr : swiglu_I + r * 8 + (fr - 8)) : r * 16 + fr;
    *reinterpret_cast<u32x4*>(P + piece * 8) = *reinterpret_cast<const u32x4*>(W + (size_t)row * K + t * 32 + fq * 8);
}
inline void skinny_pack(const bf16_t* W, bf16_t* P, int N, int K, int swiglu_I, hipStream_t s) {
    const size_t total = (size_t)N * K / 8;
    hipLaunchKernelGGL(skinny_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, W, P, N, K, swiglu_I);
}


// the kernels need more than the default 64 KB of dynamic LDS: raised once per process and instantiation
template <int K, int RB, int EPI, bool NORM = false>
inline bool skinny_attr() {
    constexpr size_t shm = skinny_lds_bytes(RB);
    static_assert(shm <= 160 * 1024, "LDS");
    static int state[kSkMaxDevices] = {};                 // per device: 0 = not asked yet, 1 = raised, -1 = failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kSkMaxDevices) return false;
    if (!state[dev])
        state[dev] = hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_gemm_kernel<K, RB, EPI, NORM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) == hipSuccess ? 1 : -1;
    return state[dev] > 0;
}
template <int K, int RB, int EPI, bool NORM = false>
inline void skinny_go(const SkinnyArgs& a, hipStream_t s) {
    (void)skinny_attr<K, RB, EPI, NORM>();
    hipLaunchKernelGGL((skinny_gemm_kernel<K, RB, EPI, NORM>), dim3(a.nrb * a.mt), dim3(512), skinny_lds_bytes(RB), s, a);
}
// Raise the LDS limit of every instantiation of one epilogue NOW: for callers whose first launch could otherwise happen inside a
// hipGraph stream capture (the batch decode graph).
template <int EPI>
inline bool skinny_prepare() {
    // (every attribute call is made, whatever the earlier ones returned)
    const bool r[] = {skinny_attr<1024, 1, EPI>(), skinny_attr<1024, 2, EPI>(), skinny_attr<1024, 3, EPI>(),
                      skinny_attr<2048, 1, EPI>(), skinny_attr<2048, 2, EPI>(), skinny_attr<2048, 3, EPI>(),
                      skinny_attr<3072, 1, EPI>(), skinny_attr<3072, 2, EPI>(), skinny_attr<6144, 1, EPI>()};
    bool ok = true;
    for (bool b : r) ok = ok && b;
    if constexpr (EPI != SK_RESIDUAL) {         // the normalising instantiations (K = hidden: 1024 or 2048)
        const bool rn[] = {skinny_attr<1024, 1, EPI, true>(), skinny_attr<1024, 2, EPI, true>(), skinny_attr<1024, 3, EPI, true>(),
                           skinny_attr<2048, 1, EPI, true>(), skinny_attr<2048, 2, EPI, true>(), skinny_attr<2048, 3, EPI, true>()};
        for (bool b : rn) ok = ok && b;
    }
    return ok;
}

// rb_force / a0.mt: 0 = pick (measurement hooks of tools/microbench/gemm_bench.hip)
template <int EPI>
inline void skinny_launch(const SkinnyArgs& a0, int K, hipStream_t s, int rb_force = 0) {
    SkinnyArgs a = a0;
    const int ntiles = (a.M + 15) / 16;
    const int groups1 = a.N / 16;                    // SK_SWIGLU: I / 8 pairs of 8 gate + 8 up rows = the same count
    // row blocks per wave: more of them = fewer re-reads of the token matrix; as many as still leave one full round of workgroups
    int RB = 1;
    if (rb_force) RB = rb_force;
    else if (groups1 % 3 == 0 && groups1 / 3 >= 128 && K <= 2048) RB = 3;
    else if (groups1 % 2 == 0 && groups1 / 2 >= 128 && K <= 3072) RB = 2;
    if (K > 3072 || groups1 % RB || (RB == 3 && K > 2048)) RB = 1;           // register budget: 4 K / 1024 weight fragments per row block
    a.nrb = groups1 / RB;
    if (a.mt <= 0) {
        a.mt = 1;
        while (a.nrb * a.mt * 2 <= 256 && a.mt * 2 <= ntiles) a.mt *= 2;      // fill the CUs before deepening the per-workgroup tile walk
    }
    if (a.mt > ntiles) a.mt = ntiles;
    // the normalising form: a.ssq names the rows' sum-of-squares partials (K / 16 per row); K = hidden = 1024 or 2048, M <= kSkNormMaxRows
    if constexpr (EPI != SK_RESIDUAL) {
        if (a.ssq) {
#define FQ3_SKN(KK) do { if (RB == 3) skinny_go<KK, 3, EPI, true>(a, s); else if (RB == 2) skinny_go<KK, 2, EPI, true>(a, s); \
                         else skinny_go<KK, 1, EPI, true>(a, s); } while (0)
            if (K == 1024) FQ3_SKN(1024); else FQ3_SKN(2048);
#undef FQ3_SKN
            return;
        }
    }
#define FQ3_SK(KK) do { if constexpr (KK <= 2048) { if (RB == 3) { skinny_go<KK, 3, EPI>(a, s); break; } } \
                        if constexpr (KK <= 3072) { if (RB == 2) { skinny_go<KK, 2, EPI>(a, s); break; } } \
                        skinny_go<KK, 1, EPI>(a, s); } while (0)
    switch (K) {
        case 1024: FQ3_SK(1024); break;
        case 2048: FQ3_SK(2048); break;
        case 3072: FQ3_SK(3072); break;
        default:   FQ3_SK(6144); break;
    }
#undef FQ3_SK
}


#ifdef FQ3_SKINNY_DEFINE
void skinny_launch_epi(int epi, const SkinnyArgs& a, int K, hipStream_t s, int rb_force) {
    if (epi == SK_SWIGLU) skinny_launch<SK_SWIGLU>(a, K, s, rb_force);
    else if (epi == SK_RESIDUAL) skinny_launch<SK_RESIDUAL>(a, K, s, rb_force);
    else skinny_launch<SK_STORE>(a, K, s, rb_force);
}
void skinny_pack_ext(const bf16_t* W, bf16_t* P, int N, int K, int swiglu_I, hipStream_t s) { skinny_pack(W, P, N, K, swiglu_I, s); }
bool skinny_prepare_epi(int epi) {
    return epi == SK_SWIGLU ? skinny_prepare<SK_SWIGLU>() : (epi == SK_RESIDUAL ? skinny_prepare<SK_RESIDUAL>() : skinny_prepare<SK_STORE>());
}
#endif
#endif      // FQ3_SKINNY_EXTERN

}  // namespace fq3
