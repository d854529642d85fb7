// Shared device helpers for libfq3hip (gfx950 / CDNA4 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fq3 {

typedef uint16_t bf16_t;   // raw bfloat16 bits

__device__ __forceinline__ float bf16_to_f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, same as torch's float->bfloat16 cast (NaN not special-cased: inputs are finite or +-inf)
__device__ __forceinline__ bf16_t f_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename T> struct DT;
template <> struct DT<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f_to_bf16(v); }
    static __device__ __forceinline__ float rnd(float v) { return bf16_to_f(f_to_bf16(v)); }
};
template <> struct DT<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
    static __device__ __forceinline__ float rnd(float v) { return v; }
};

// ---- 8-element (one lane's chunk) raw loads: issue now, convert later ------------------------
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T> struct Raw8;
template <> struct Raw8<bf16_t> { u32x4 v; };
template <> struct Raw8<float> { f32x4 a, b; };

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

// ---- wave / block reductions ------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
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
