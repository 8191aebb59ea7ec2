// MFMA / LDS-DMA helpers shared by the conv kernels of libsdnet_hip.so (gfx950 only).
#pragma once
#include "sd_common.h"

namespace sd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }   // round-to-nearest-even (v_cvt_pk_bf16_f32)

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // blocks b and b+8 share an XCD (round-robin dispatch): give each XCD a contiguous range of
    // tiles so that neighbouring tiles (shared input rows / weight panels) hit the same L2.
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// 16-byte LDS-DMA: each lane's global address is its own, the LDS destination is (wave-uniform base + 16 * lane).
// (The address-space cast only exists in the device pass; the host pass of hipcc just needs a stub body.)
__device__ __forceinline__ void lds_dma16(const void* gsrc, float* lds_wave_base) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#else
    (void)gsrc; (void)lds_wave_base;
#endif
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));     // gfx9 encoding: expcnt / lgkmcnt untouched
#endif
}
// all but the N youngest vector-memory operations done AND every LDS read of this wave returned (before a bare s_barrier:
// the stage this wave was reading may be overwritten by the other waves' DMA right after it)
template <int N>
__device__ __forceinline__ void wait_vmcnt_and_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (0 << 8) | ((N >> 4) << 14));
#endif
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// LDS byte address of a __shared__ object (device pass only; the host pass needs a body)
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
#else
    (void)p; return 0;
#endif
}
// ds_read_b128 the compiler does not know about: its wait insertion would otherwise put `s_waitcnt vmcnt(0)` in front of
// the fragment reads at the loop header (in-flight LDS-DMA of OTHER stages counted as possibly aliasing) and drain the
// prefetch once per trip.  The result is valid only after lds_wait_*() below.
template <int OFF>
__device__ __forceinline__ f32x4 lds_read128_async(uint32_t addr) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// s_waitcnt lgkmcnt(N) that the fragments are threaded through, so that no consumer can be scheduled above it
#define SD_LDS_WAIT6(N, a, b, c, d, e, f) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f) :: "memory")
#define SD_LDS_WAIT8(N, a, b, c, d, e, f, g, h) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) :: "memory")
#define SD_LDS_WAIT4(N, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")
#define SD_LDS_WAIT2(N, a, b) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a), "+v"(b) :: "memory")

}  // namespace sd
