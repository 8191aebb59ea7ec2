// Micro-benchmark: v_mfma_f32_32x32x16_bf16 issue rate with the B operand in a VGPR vs an AGPR, accumulators in AGPRs, one wave per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_agpr.hip -o tools/micro/mfma_agpr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>   // 0: B in VGPR, 2 accumulators alternating; 1: B in AGPR; 2: A and B in AGPR; 3: B in VGPR, ONE accumulator; 4: B AGPR one accumulator
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
    f32x16 c0, c1;
    f32x4 a = {1.f, 2.f, 3.f, 4.f}, b0 = {1.f, 1.f, 1.f, 1.f}, b1 = {2.f, 2.f, 2.f, 2.f};
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    asm volatile("" : "+a"(c0), "+a"(c1));
    if (MODE == 1 || MODE == 4) asm volatile("" : "+a"(b0), "+a"(b1));
    if (MODE == 2) asm volatile("" : "+a"(b0), "+a"(b1), "+a"(a));
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 0) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b0)); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "v"(b1)); }
            if (MODE == 1) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "a"(b0)); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(a), "a"(b1)); }
            if (MODE == 2) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "a"(a), "a"(b0)); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "a"(a), "a"(b1)); }
            if (MODE == 3) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b0)); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "v"(b1)); }
            if (MODE == 4) { asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "a"(b0)); asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(a), "a"(b1)); }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3];
}
template <int MODE> void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    k<MODE><<<256, 256>>>(out, cyc, 10);
    k<MODE><<<256, 256>>>(out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[256]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
    printf("%-60s %6.1f cycles per MFMA\n", name, avg / (iters * 16.0));
}
int main() {
    float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&cyc, 4096));
    run<0>("A, B in VGPRs, two accumulators (AGPR) alternating", out, cyc);
    run<1>("B in AGPRs", out, cyc);
    run<2>("A and B in AGPRs", out, cyc);
    run<3>("A, B in VGPRs, ONE accumulator chain", out, cyc);
    run<4>("B in AGPRs, ONE accumulator chain", out, cyc);
    return 0;
}
