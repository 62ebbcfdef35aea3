// Shared helpers for the gfx950 kernels: status/error plumbing, 16-bit element traits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/panfusion_hip.h"

namespace pf {

void set_error(const char* fmt, ...);

#define PF_REQUIRE(cond, ...)                         \
    do {                                              \
        if (!(cond)) {                                \
            pf::set_error(__VA_ARGS__);               \
            return PF_ERR_ARG;                        \
        }                                             \
    } while (0)

#define PF_CHECK_LAUNCH(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            pf::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));    \
            return PF_ERR_LAUNCH;                                                    \
        }                                                                            \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline long cdiv(long a, long b) { return (a + b - 1) / b; }

// ---- 16-bit element types ------------------------------------------------------------------
// Storage is always a raw 16-bit word; Bf16 / F16 tag types select the conversion and the MFMA.
struct Bf16 { static constexpr int id = PF_BF16; };
struct F16  { static constexpr int id = PF_F16; };

typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <typename T> __device__ __forceinline__ float to_f32(unsigned short v);
template <> __device__ __forceinline__ float to_f32<Bf16>(unsigned short v) {
    return __uint_as_float(static_cast<unsigned>(v) << 16);
}
template <> __device__ __forceinline__ float to_f32<F16>(unsigned short v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return static_cast<float>(h);
}

template <typename T> __device__ __forceinline__ unsigned short from_f32(float f);
template <> __device__ __forceinline__ unsigned short from_f32<Bf16>(float f) {
    __bf16 b = static_cast<__bf16>(f);   // v_cvt_pk_bf16_f32 on gfx950, round to nearest even
    unsigned short v;
    __builtin_memcpy(&v, &b, 2);
    return v;
}
template <> __device__ __forceinline__ unsigned short from_f32<F16>(float f) {
    _Float16 h = static_cast<_Float16>(f);   // v_cvt_f16_f32, RNE
    unsigned short v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}

template <typename T> __device__ __forceinline__ void unpack8(const u16x8& v, float (&f)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = to_f32<T>(v[i]);
}
template <typename T> __device__ __forceinline__ u16x8 pack8(const float (&f)[8]) {
    u16x8 v;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = from_f32<T>(f[i]);
    return v;
}

// Element offset of channel c inside a split-precision pair row: per block of 32 channels [hi(32) | lo(32)] (lo = + 32).  A 64-element
// K block of the GEMM kernels then holds hi and lo of the SAME 32 channels (pf_conv_desc.split3).
__host__ __device__ __forceinline__ int pair_off(int c) { return ((c >> 5) << 6) | (c & 31); }

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }

// Element access by storage tag (fp32 / bf16 / fp16) for the kernels that take any of the three.
template <typename S> __device__ __forceinline__ float ld_any(const void* p, long i);
struct AnyF32 {}; struct AnyBf16 {}; struct AnyF16 {};
template <> __device__ __forceinline__ float ld_any<AnyF32>(const void* p, long i) { return static_cast<const float*>(p)[i]; }
template <> __device__ __forceinline__ float ld_any<AnyBf16>(const void* p, long i) { return to_f32<Bf16>(static_cast<const unsigned short*>(p)[i]); }
template <> __device__ __forceinline__ float ld_any<AnyF16>(const void* p, long i) { return to_f32<F16>(static_cast<const unsigned short*>(p)[i]); }
template <typename S> __device__ __forceinline__ void st_any(void* p, long i, float v);
template <> __device__ __forceinline__ void st_any<AnyF32>(void* p, long i, float v) { static_cast<float*>(p)[i] = v; }
template <> __device__ __forceinline__ void st_any<AnyBf16>(void* p, long i, float v) { static_cast<unsigned short*>(p)[i] = from_f32<Bf16>(v); }
template <> __device__ __forceinline__ void st_any<AnyF16>(void* p, long i, float v) { static_cast<unsigned short*>(p)[i] = from_f32<F16>(v); }

// v_mfma_f32_32x32x16 on 16-bit operands (attention forward / backward).  A[row][k]: lane l supplies row l & 31,
// k = 8 (l >> 5) .. +7; B[k][col]: col l & 31, same k; C[row][col]: lane l holds col l & 31, rows (r & 3) + 8 (r >> 2) + 4 (l >> 5).
template <typename T> struct Mfma32;
template <> struct Mfma32<Bf16> {
    typedef __attribute__((ext_vector_type(8))) __bf16 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct Mfma32<F16> {
    typedef __attribute__((ext_vector_type(8))) _Float16 frag;
    static __device__ __forceinline__ f32x16 run(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// value * gelu(gate), erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7), arranged as  g Phi(g) = relu(g) - |g| h,  h = 0.5 P(t) exp(-g^2 / 2),
// t = 1 / (1 + p |g| / sqrt 2): 12 plain VALU operations + rcp + exp2 per output (the straightforward 0.5 g (1 + erf(g / sqrt 2))
// with copysign costs 20): the GEGLU epilogues are VALU-bound
// (profiles/r4e_lws_pmc_kernel.txt: VALU busy 1.5 x MFMA busy).  Used by both GEMM kernels (pf_gemm.hip, pf_linear_ws.hip): transformer.py:8-21.
__device__ __forceinline__ float geglu_value(float v, float g) {
    const float ax = fabsf(g);
    const float t = __builtin_amdgcn_rcpf(fmaf(ax, 0.23164189f, 1.0f));
    float y = fmaf(t, 0.5307027145f, -0.7265760135f);
    y = fmaf(y, t, 0.7107068705f);
    y = fmaf(y, t, -0.142248368f);
    y = fmaf(y, t, 0.127414796f);
    y *= t;
    const float h = y * __builtin_amdgcn_exp2f(-0.72134752044448170368f * (g * g));
    return v * fmaf(-ax, h, fmaxf(g, 0.0f));
}

// Dispatch a 16-bit dtype id to a tag type.
#define PF_DISPATCH_16(dtype, name, ...)                                         \
    do {                                                                         \
        if ((dtype) == PF_BF16) { using T = pf::Bf16; __VA_ARGS__; }             \
        else if ((dtype) == PF_F16) { using T = pf::F16; __VA_ARGS__; }          \
        else { pf::set_error("%s: dtype must be PF_BF16 or PF_F16", name); return PF_ERR_ARG; } \
    } while (0)

}  // namespace pf
