// Micro-benchmark: v_mfma_f32_32x32x2_f32 (64 cycles) issued by ONE wave per SIMD with other instructions between the MFMAs:
// what a 4-MFMA step of k_conv3x3_c64_rows_f32 costs with its address add, ds_read_b128 and s_waitcnt, and with real operand data.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_mix.hip -o tools/micro/mfma_f32_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 4: MFMAs only, ONE accumulator chain; 5: as 2 but one s_waitcnt per EIGHT MFMAs (two reads issued together); MODE 0: MFMAs only; 1: + v_add per 4; 2: + ds_read_b128 (3 ahead) + s_waitcnt lgkmcnt(3) per 4; 3: as 2 with random data in LDS and B
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters, const float* rnd) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = MODE == 3 ? rnd[i] : 0.f;
    __syncthreads();
    f32x16 c0, c1;
    float b[8];
    for (int i = 0; i < 8; ++i) b[i] = MODE == 3 ? rnd[threadIdx.x * 8 + i] : 0.f;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    asm volatile("" : "+a"(c0), "+a"(c1));
    asm volatile("" : "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]), "+a"(b[4]), "+a"(b[5]), "+a"(b[6]), "+a"(b[7]));
    uint32_t base = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 16, off = 1024;
    f32x4 A[4];
    for (int u = 0; u < 4; ++u) A[u] = f32x4{1.f, 2.f, 3.f, 4.f};
    if (MODE == 5) for (int u = 0; u < 2; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(A[u]) : "v"(base + u * 1024) : "memory");
    else if (MODE >= 2 && MODE != 4) for (int u = 0; u < 3; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(A[u]) : "v"(base + u * 1024) : "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 5) {
                if ((u & 1) == 0) {      // two reads with immediate offsets off one address register, one wait per eight MFMAs
                    asm volatile("ds_read_b128 %0, %1" : "=v"(A[(u + 2) & 3]) : "v"(base) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(A[(u + 3) & 3]) : "v"(base) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[u & 3]), "+v"(A[(u + 1) & 3]) :: "memory");
                }
            } else
            if (MODE >= 1 && MODE != 4) { uint32_t ad; asm volatile("v_add_u32 %0, %1, %2" : "=v"(ad) : "v"(base), "v"(off));
                             if (MODE >= 2) { asm volatile("ds_read_b128 %0, %1" : "=v"(A[(u + 3) & 3]) : "v"(ad) : "memory");
                                              asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[u & 3]) :: "memory"); }
                             else asm volatile("" :: "v"(ad)); }
            const f32x4 a = A[u & 3];
            if (MODE == 4) {
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a[0]), "a"(b[0]));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a[1]), "a"(b[1]));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a[2]), "a"(b[2]));
                asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a[3]), "a"(b[3]));
            } else {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a[0]), "a"(b[0]));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c1) : "v"(a[1]), "a"(b[1]));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c0) : "v"(a[2]), "a"(b[2]));
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c1) : "v"(a[3]), "a"(b[3]));
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3] + A[0][0] + A[1][1] + A[2][2] + A[3][3];
}
template <int MODE> void run(const char* name, float* out, unsigned long long* cyc, const float* rnd) {
    const int iters = 1000;
    k<MODE><<<256, 256>>>(out, cyc, 10, rnd);
    k<MODE><<<256, 256>>>(out, cyc, iters, rnd);
    CHECK(hipDeviceSynchronize());
    unsigned long long h[256]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
    printf("%-70s %6.1f cycles per MFMA\n", name, avg / (iters * 32.0));
}
int main() {
    float* out; unsigned long long* cyc; float* rnd;
    CHECK(hipMalloc(&out, 1 << 20)); CHECK(hipMalloc(&cyc, 4096)); CHECK(hipMalloc(&rnd, 1 << 16));
    float* h = (float*)malloc(1 << 16);
    for (int i = 0; i < (1 << 14); ++i) h[i] = (float)rand() / RAND_MAX - 0.5f;
    CHECK(hipMemcpy(rnd, h, 1 << 16, hipMemcpyHostToDevice));
    run<0>("32x32x2 f32 MFMAs only (B in AGPRs, two accumulators)", out, cyc, rnd);
    run<1>("+ one v_add_u32 per four MFMAs", out, cyc, rnd);
    run<2>("+ ds_read_b128 three steps ahead + s_waitcnt lgkmcnt(3) per four MFMAs", out, cyc, rnd);
    run<3>("the same on random operands", out, cyc, rnd);
    run<4>("MFMAs only, ONE accumulator chain", out, cyc, rnd);
    run<5>("two ds_read_b128 (immediate offsets, no v_add) + one s_waitcnt per EIGHT MFMAs", out, cyc, rnd);
    return 0;
}
