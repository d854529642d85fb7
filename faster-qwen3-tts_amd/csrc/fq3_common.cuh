// Shared device helpers for libfq3hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace fq3 {

typedef uint16_t bf16_t;   // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, same as torch's float->bfloat16 cast: gfx950 has the conversion in hardware
// (v_cvt_pk_bf16_f32), one instruction instead of the 5-op integer emulation
__device__ __forceinline__ bf16_t f_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
typedef float fq3_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 fq3_bf16x2 __attribute__((ext_vector_type(2)));
// two values per v_cvt_pk_bf16_f32: returns lo = bf16(a), hi = bf16(b)
__device__ __forceinline__ uint32_t pack_bf16x2(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(fq3_f32x2{a, b}, fq3_bf16x2));
}

template <typename T> struct DT;
template <> struct DT<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f_to_bf16(v); }
    static __device__ __forceinline__ float rnd(float v) { return bf16_to_f(f_to_bf16(v)); }
    // round two values with one conversion instruction
    static __device__ __forceinline__ void rnd2(float& a, float& b) {
        const uint32_t u = pack_bf16x2(a, b);
        a = __uint_as_float(u << 16); b = __uint_as_float(u & 0xFFFF0000u);
    }
    // store 8 consecutive values (16-byte aligned destination)
    static __device__ __forceinline__ void st8(bf16_t* p, const float (&f)[8]);
};
template <> struct DT<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
    static __device__ __forceinline__ void rnd2(float&, float&) {}
    static __device__ __forceinline__ void st8(float* p, const float (&f)[8]);
};

// "bf16 x 2": a value kept as a bf16 high part + a bf16 residual in one 32-bit word (low half = high part, so that in memory
// the pair reads as two consecutive bf16: [hi, lo]).  16 mantissa bits: one rounding is 2^-17 relative, against 2^-9 for bf16.
// The codec decoder's high-precision mode stores every activation like this (fq3_codec.hip, FQ3_BF16X2): a [rows][C] tensor of
// bfs_t IS a [rows][2C] bf16 matrix, so a GEMM against bf16 weights whose K columns are duplicated ([.., w_c, w_c, ..]) runs on
// v_mfma_f32_16x16x32_bf16 unchanged -- hi * w and lo * w are exact fp32 products -- at twice the bf16 MFMA count instead of the
// 16x of v_mfma_f32_16x16x4_f32.
struct bfs_t { uint32_t u; };
__device__ __forceinline__ float bfs_to_f(uint32_t u) { return __uint_as_float(u << 16) + __uint_as_float(u & 0xFFFF0000u); }
__device__ __forceinline__ uint32_t f_to_bfs(float v) {
    const uint32_t hi = f_to_bf16(v);
    const float r = v - __uint_as_float(hi << 16);           // exact in fp32
    return hi | ((uint32_t)f_to_bf16(r) << 16);
}
template <> struct DT<bfs_t> {
    static __device__ __forceinline__ float ld(const bfs_t* p) { return bfs_to_f(p->u); }
    static __device__ __forceinline__ void st(bfs_t* p, float v) { p->u = f_to_bfs(v); }
    static __device__ __forceinline__ float rnd(float v) { return bfs_to_f(f_to_bfs(v)); }
    static __device__ __forceinline__ void rnd2(float& a, float& b) { a = rnd(a); b = rnd(b); }
    static __device__ __forceinline__ void st8(bfs_t* p, const float (&f)[8]);
};

// A pointer that was LOADED from memory (a per-lane pointer table, a field of the on-device loop state) is a generic ("flat") pointer
// to the compiler: its accesses become flat_load / flat_store, which count against BOTH vmcnt and lgkmcnt and may return out of order
// between the two paths -- every wait on them is `s_waitcnt vmcnt(0) lgkmcnt(0)`, i.e. the counted, pipelined waits the kernels rely on
// collapse.  Everything this library points at lives in device global memory: say so (generic -> integer -> address space 1 ->
// generic; the address-space inference pass then rewrites the accesses to global_load / global_store).
template <typename U>
__device__ __forceinline__ U* gptr(U* p) {
    typedef U __attribute__((address_space(1))) GU;
    return (U*)reinterpret_cast<GU*>(reinterpret_cast<uintptr_t>(p));
}

// ---- 8-element (one lane's chunk) raw loads: issue now, convert later ------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { u32x4 v; };
template <> struct Raw8<float> { f32x4 a, b; };
template <> struct Raw8<bfs_t> { u32x4 a, b; };

__device__ __forceinline__ void DT<bf16_t>::st8(bf16_t* p, const float (&f)[8]) {
    *reinterpret_cast<u32x4*>(p) = u32x4{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
}
__device__ __forceinline__ void DT<float>::st8(float* p, const float (&f)[8]) {
    reinterpret_cast<f32x4*>(p)[0] = f32x4{f[0], f[1], f[2], f[3]};
    reinterpret_cast<f32x4*>(p)[1] = f32x4{f[4], f[5], f[6], f[7]};
}

__device__ __forceinline__ void DT<bfs_t>::st8(bfs_t* p, const float (&f)[8]) {
    reinterpret_cast<u32x4*>(p)[0] = u32x4{f_to_bfs(f[0]), f_to_bfs(f[1]), f_to_bfs(f[2]), f_to_bfs(f[3])};
    reinterpret_cast<u32x4*>(p)[1] = u32x4{f_to_bfs(f[4]), f_to_bfs(f[5]), f_to_bfs(f[6]), f_to_bfs(f[7])};
}
template <bool NT> __device__ __forceinline__ void ldraw(Raw8<bfs_t>& r, const bfs_t* p) {
    r.a = reinterpret_cast<const u32x4*>(p)[0];
    r.b = reinterpret_cast<const u32x4*>(p)[1];
}
__device__ __forceinline__ void zero(Raw8<bfs_t>& r) { r.a = u32x4{0u, 0u, 0u, 0u}; r.b = r.a; }
__device__ __forceinline__ void unpack(const Raw8<bfs_t>& r, float (&f)[8]) {
    f[0] = bfs_to_f(r.a.x); f[1] = bfs_to_f(r.a.y); f[2] = bfs_to_f(r.a.z); f[3] = bfs_to_f(r.a.w);
    f[4] = bfs_to_f(r.b.x); f[5] = bfs_to_f(r.b.y); f[6] = bfs_to_f(r.b.z); f[7] = bfs_to_f(r.b.w);
}

template <bool NT> __device__ __forceinline__ void ldraw(Raw8<bf16_t>& r, const bf16_t* p) {
    if (NT) r.v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p));
    else r.v = *reinterpret_cast<const u32x4*>(p);
}
template <bool NT> __device__ __forceinline__ void ldraw(Raw8<float>& r, const float* p) {
    if (NT) {
        r.a = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
        r.b = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p) + 1);
    } else {
        r.a = reinterpret_cast<const f32x4*>(p)[0];
        r.b = reinterpret_cast<const f32x4*>(p)[1];
    }
}
__device__ __forceinline__ void zero(Raw8<bf16_t>& r) { r.v = u32x4{0u, 0u, 0u, 0u}; }
__device__ __forceinline__ void zero(Raw8<float>& r) { r.a = f32x4{0.f, 0.f, 0.f, 0.f}; r.b = r.a; }

__device__ __forceinline__ void unpack(const Raw8<bf16_t>& r, float (&f)[8]) {
    f[0] = __uint_as_float(r.v.x << 16); f[1] = __uint_as_float(r.v.x & 0xFFFF0000u);
    f[2] = __uint_as_float(r.v.y << 16); f[3] = __uint_as_float(r.v.y & 0xFFFF0000u);
    f[4] = __uint_as_float(r.v.z << 16); f[5] = __uint_as_float(r.v.z & 0xFFFF0000u);
    f[6] = __uint_as_float(r.v.w << 16); f[7] = __uint_as_float(r.v.w & 0xFFFF0000u);
}
__device__ __forceinline__ void unpack(const Raw8<float>& r, float (&f)[8]) {
    f[0] = r.a.x; f[1] = r.a.y; f[2] = r.a.z; f[3] = r.a.w;
    f[4] = r.b.x; f[5] = r.b.y; f[6] = r.b.z; f[7] = r.b.w;
}
// sequential fp32 fma chain over the 8 elements (fixed order => deterministic)
template <typename T> __device__ __forceinline__ float dot8(const Raw8<T>& r, const float (&x)[8], float acc) {
    float w[8];
    unpack(r, w);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(w[i], x[i], acc);
    return acc;
}

// RMSNorm of 8 values straight to packed bf16: bf16(bf16(x * rs) * w), the two roundings of the reference's norm (and of the loop it
// replaces: same products, same roundings), in 5 VALU operations per value instead of 8 -- packed fp32 multiplies (v_pk_mul_f32:
// IEEE products, two per instruction) and no unpack / re-pack round trip after the second rounding.  The normalisation prologue is
// VALU time every workgroup spends on every token (~1700 instructions per wave at 32 lanes x K = 1024).
__device__ __forceinline__ u32x4 norm8_pack(const float (&x)[8], float rs, const float (&w)[8]) {
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const fq3_f32x2 p = fq3_f32x2{x[2 * i], x[2 * i + 1]} * fq3_f32x2{rs, rs};
        const uint32_t u = pack_bf16x2(p.x, p.y);
        const fq3_f32x2 q = fq3_f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)} * fq3_f32x2{w[2 * i], w[2 * i + 1]};
        o[i] = pack_bf16x2(q.x, q.y);
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}

// ---- wave / block reductions ------------------------------------------------------------------
// DPP (data-parallel primitive) lane exchanges run in the VALU pipe; __shfl_xor compiles to ds_bpermute_b32, an LDS
// round trip of ~100 cycles.  A 64-lane butterfly of six dependent bpermutes costs ~0.25 us -- measurable against a
// 3-5 us decode kernel -- so every reduction below is DPP: 4 row ops inside each 16-lane row, then row_bcast:15/31.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_move(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppMirror = 0x140;
constexpr int kDppBcast15 = 0x142, kDppBcast31 = 0x143, kDppRor8 = 0x128;
// every lane gets the sum / max over its 16-lane row (same pairing as an xor-1,2,4,8 butterfly)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<kDppXor1, 0xF>(0.f, v); v += dpp_move<kDppXor2, 0xF>(0.f, v);
    v += dpp_move<kDppHalfMirror, 0xF>(0.f, v); v += dpp_move<kDppMirror, 0xF>(0.f, v);
    return v;
}
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_move<kDppXor1, 0xF>(v, v)); v = fmaxf(v, dpp_move<kDppXor2, 0xF>(v, v));
    v = fmaxf(v, dpp_move<kDppHalfMirror, 0xF>(v, v)); v = fmaxf(v, dpp_move<kDppMirror, 0xF>(v, v));
    return v;
}
// value of lane (lane ^ 8) inside the 16-lane row
__device__ __forceinline__ float row16_xor8(float v) { return dpp_move<kDppRor8, 0xF>(v, v); }
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    v += dpp_move<kDppBcast15, 0xA>(0.f, v);        // rows 1, 3 += row 0, 2
    v += dpp_move<kDppBcast31, 0xC>(0.f, v);        // rows 2, 3 += rows 0 + 1
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    v = fmaxf(v, dpp_move<kDppBcast15, 0xA>(v, v));
    v = fmaxf(v, dpp_move<kDppBcast31, 0xC>(v, v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_i32(int v) {
    auto mv = [](auto ctrl, auto mask, int x) {
        return __builtin_amdgcn_update_dpp(x, x, decltype(ctrl)::value, decltype(mask)::value, 0xF, false);
    };
    using std::integral_constant;
    v = min(v, mv(integral_constant<int, kDppXor1>{}, integral_constant<int, 0xF>{}, v));
    v = min(v, mv(integral_constant<int, kDppXor2>{}, integral_constant<int, 0xF>{}, v));
    v = min(v, mv(integral_constant<int, kDppHalfMirror>{}, integral_constant<int, 0xF>{}, v));
    v = min(v, mv(integral_constant<int, kDppMirror>{}, integral_constant<int, 0xF>{}, v));
    v = min(v, mv(integral_constant<int, kDppBcast15>{}, integral_constant<int, 0xA>{}, v));
    v = min(v, mv(integral_constant<int, kDppBcast31>{}, integral_constant<int, 0xC>{}, v));
    return __builtin_amdgcn_readlane(v, 63);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false); }
__device__ __forceinline__ int wave_sum_i32(int v) {
    v += dpp_move_i<kDppXor1, 0xF>(0, v); v += dpp_move_i<kDppXor2, 0xF>(0, v);
    v += dpp_move_i<kDppHalfMirror, 0xF>(0, v); v += dpp_move_i<kDppMirror, 0xF>(0, v);
    v += dpp_move_i<kDppBcast15, 0xA>(0, v);
    v += dpp_move_i<kDppBcast31, 0xC>(0, v);
    return __builtin_amdgcn_readlane(v, 63);
}
// lane-wise reductions ACROSS the four 16-lane rows (lanes c, c+16, c+32, c+48): gfx950's v_permlane32_swap /
// v_permlane16_swap exchange half-waves / odd-even rows in the VALU pipe; swap(v, v) leaves {v[lane], v[partner]} in
// the two results, so a commutative combine needs no knowledge of which is which.  Every lane gets the result.
typedef unsigned fq3_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float xrow_sum(float v) {
    fq3_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(r.x) + __uint_as_float(r.y);
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r.x) + __uint_as_float(r.y);
}
__device__ __forceinline__ float xrow_max(float v) {
    fq3_u32x2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
    r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r.x), __uint_as_float(r.y));
}
// block-wide sum for blockDim.x == 64*NW; red must hold NW floats; all threads get the result
template <int NW> __device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) t += red[i];
    return t;
}
template <int NW> __device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) t = fmaxf(t, red[i]);
    return t;
}

}  // namespace fq3
