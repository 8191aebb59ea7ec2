// Shared host/device helpers for libsdnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/sdnet_hip.h"

namespace sd {

void set_error(const char* fmt, ...);

#define SD_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            sd::set_error(__VA_ARGS__);        \
            return (code);                     \
        }                                      \
    } while (0)

#define SD_HIP(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            sd::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                             \
        }                                                                               \
    } while (0)

#define SD_LAUNCH_CHECK()                                                               \
    do {                                                                                \
        hipError_t e_ = hipGetLastError();                                              \
        if (e_ != hipSuccess) {                                                         \
            sd::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                             \
        }                                                                               \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

constexpr float kClampLo = 1e-6f;                    // fp32(1e-6)
constexpr float kClampHi = (float)(1.0 - 1e-6);      // fp32(1-1e-6) = 0.99999898672

// clamp(sigmoid(x), 1e-6, 1-1e-6): src/sdnet/utils/utils.py:355-361
__device__ __forceinline__ float sigmoid_raw(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float clamped_sigmoid(float x) {
    return fminf(fmaxf(sigmoid_raw(x), kClampLo), kClampHi);
}

// bf16 <-> fp32 (round to nearest even: v_cvt_pk_bf16_f32) and four-element activation accessors used by the kernels that exist
// for both activation types (fp32, and bf16 for the mixed-precision training path: i indexes groups of four elements)
__device__ __forceinline__ float bf16_to_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }
// (activation streams of the HBM-bound passes are read ONCE: non-temporal loads -- `global_load_dwordx4 ... nt` -- keep them from displacing
//  what the next kernel re-reads; same-box A/B: fp32 step -0.7 %, mixed-precision step -0.6 %; non-temporal STORES changed nothing)
typedef float sd_f32x4_nt __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p, int64_t i) {
    const sd_f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const sd_f32x4_nt*>(p) + i);
    return make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ float4 ld4(const uint16_t* p, int64_t i) {
    const uint2 r = reinterpret_cast<const uint2*>(p)[i];
    return make_float4(bf16_to_f32((uint16_t)(r.x & 0xffff)), bf16_to_f32((uint16_t)(r.x >> 16)), bf16_to_f32((uint16_t)(r.y & 0xffff)),
                       bf16_to_f32((uint16_t)(r.y >> 16)));
}
__device__ __forceinline__ void st4(float* p, int64_t i, const float4 v) { reinterpret_cast<float4*>(p)[i] = v; }
__device__ __forceinline__ void st4(uint16_t* p, int64_t i, const float4 v) {
    uint2 pk;
    pk.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    pk.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    reinterpret_cast<uint2*>(p)[i] = pk;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

}  // namespace sd
