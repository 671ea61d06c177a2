// Shared helpers for the gfx950 kernels of libc2mi.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/c2mi.h"

namespace c2 {

typedef unsigned short bf16_t;  // raw bf16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

void set_error(const std::string& msg);

#define C2_CHECK(cond, msg)                                      \
    do {                                                         \
        if (!(cond)) {                                           \
            c2::set_error(std::string(__func__) + ": " + (msg)); \
            return 1;                                            \
        }                                                        \
    } while (0)

#define C2_LAUNCH_CHECK()                                                             \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            c2::set_error(std::string(__func__) + ": launch failed: " + hipGetErrorString(e__)); \
            return 2;                                                                 \
        }                                                                             \
    } while (0)

// fp32 -> bf16 round-to-nearest-even; the plain cast keeps NaN a NaN on gfx950.
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    unsigned int u = ((unsigned int)b) << 16;
    return __builtin_bit_cast(float, u);
}

typedef _Float16 f16_t;  // IEEE half (C2_F16)
// two fp32 values rounded to T (bf16_t or f16_t), packed low | high << 16
template <typename T>
__device__ __forceinline__ unsigned pack2(float a, float b);
template <>
__device__ __forceinline__ unsigned pack2<bf16_t>(float a, float b) {
    return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16);
}
template <>
__device__ __forceinline__ unsigned pack2<f16_t>(float a, float b) {
    return (unsigned)__builtin_bit_cast(unsigned short, (f16_t)a) | ((unsigned)__builtin_bit_cast(unsigned short, (f16_t)b) << 16);
}

template <typename T>
__device__ __forceinline__ float ld_as_f32(const T* p);
template <>
__device__ __forceinline__ float ld_as_f32<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p) { return bf2f(*p); }

template <typename T>
__device__ __forceinline__ void st_from_f32(T* p, float v);
template <>
__device__ __forceinline__ void st_from_f32<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void st_from_f32<bf16_t>(bf16_t* p, float v) { *p = f2bf(v); }
template <>
__device__ __forceinline__ void st_from_f32<f16_t>(f16_t* p, float v) { *p = (f16_t)v; }

// log(1 + x) of a count (x >= 0) on v_log_f32 (1 ulp): 1 + x is a normal number >= 1, so none of libm's range and
// denormal handling is needed -- the library logf costs ~30 VALU instructions per element and made the batch gather
// VALU-bound.
__device__ __forceinline__ float log1p_count(float x) { return __builtin_amdgcn_logf(1.0f + x) * 0.6931471805599453f; }

// 64-lane butterfly reductions
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

// Block-wide sum, result valid in every thread.  red: LDS scratch of >= 32 floats.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    constexpr int NW = NT / 64;
    v = wave_sum(v);
    if (NW == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}
template <int NT>
__device__ __forceinline__ float block_max(float v, float* red) {
    constexpr int NW = NT / 64;
    v = wave_max(v);
    if (NW == 1) return v;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) s = fmaxf(s, red[i]);
    return s;
}

// Deterministic second stage of a loss reduction: one block sums `n` partials and ADDS to *out.
int launch_finalize_sum(const float* partials, int n, float* out, hipStream_t stream);
bool set_defer_sums(bool on);  // (epilogue.hip) deferral of those sums for this thread, on / off; returns what it was

constexpr int kReduceWsFloats = 8192;  // max partials of a two-stage loss reduction

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace c2
