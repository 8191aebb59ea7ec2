// Micro-benchmark: v_mfma_f32_32x32x2_f32 (64 cycles) issued by one or two waves per SIMD with other instructions between the MFMAs
// (what a 4-MFMA step of k_conv3x3_c64_rows_f32 costs with an address add, ds_read_b128 and s_waitcnt), and the fp32 MFMA rate the
// chip sustains on zero and on random operands (wall clock: the power limit sets the clock).  MI355X, round 2: pure MFMA stream
// 147-151 TFLOP/s on zeros, 139 TFLOP/s on random operands; one v_add per four MFMAs: -5.7 % with one wave per SIMD, -1.7 % with two.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_mix.hip -o tools/micro/mfma_f32_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 4: MFMAs only, ONE accumulator chain; 5: as 2 but one s_waitcnt per EIGHT MFMAs (two reads issued together); MODE 0: MFMAs only; 1: + v_add per 4; 2: + ds_read_b128 (3 ahead) + s_waitcnt lgkmcnt(3) per 4; 3: as 2 with random data in LDS and B
template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int iters, const float* rnd) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (MODE == 3 || MODE == 6 || MODE == 7) ? rnd[i] : 0.f;
    __syncthreads();
    f32x16 c0, c1;
    float b[8];
    for (int i = 0; i < 8; ++i) b[i] = (MODE == 3 || MODE == 6 || MODE == 7) ? rnd[(threadIdx.x & 255) * 8 + i] : 0.f;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    asm volatile("" : "+a"(c0), "+a"(c1));
    asm volatile("" : "+a"(b[0]), "+a"(b[1]), "+a"(b[2]), "+a"(b[3]), "+a"(b[4]), "+a"(b[5]), "+a"(b[6]), "+a"(b[7]));
    uint32_t base = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 16, off = 1024;
    f32x4 A[4];
    for (int u = 0; u < 4; ++u) A[u] = (MODE == 6 || MODE == 7) ? f32x4{rnd[threadIdx.x + 64 * u], rnd[threadIdx.x + 64 * u + 1000], rnd[threadIdx.x + 64 * u + 2000], rnd[threadIdx.x + 64 * u + 3000]} : f32x4{1.f, 2.f, 3.f, 4.f};
    if (MODE == 5 || MODE == 7) for (int u = 0; u < 2; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(A[u]) : "v"(base + u * 1024) : "memory");
    else if (MODE >= 2 && MODE != 4 && MODE != 6) for (int u = 0; u < 3; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(A[u]) : "v"(base + u * 1024) : "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MODE == 5 || MODE == 7) {
                if ((u & 1) == 0) {      // two reads with immediate offsets off one address register, one wait per eight MFMAs
                    asm volatile("ds_read_b128 %0, %1" : "=v"(A[(u + 2) & 3]) : "v"(base) : "memory");
                    asm volatile("ds_read_b128 %0, %1 offset:128" : "=v"(A[(u + 3) & 3]) : "v"(base) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[u & 3]), "+v"(A[(u + 1) & 3]) :: "memory");
                }
            } else
            if (MODE >= 1 && MODE != 4 && MODE != 6) { uint32_t ad; asm volatile("v_add_u32 %0, %1, %2" : "=v"(ad) : "v"(base), "v"(off));
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
template <int MODE, int THREADS = 256> void run(const char* name, float* out, unsigned long long* cyc, const float* rnd) {
    const int iters = 1000;
    k<MODE><<<256, THREADS>>>(out, cyc, 10, rnd);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    k<MODE><<<256, THREADS>>>(out, cyc, iters, rnd);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256]; CHECK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
    const double flops = 256.0 * (THREADS / 64) * iters * 32.0 * 4096.0;
    printf("%-70s %6.1f counter ticks per MFMA and wave; wall clock %.3f ms = %.1f TFLOP/s (%d waves per SIMD)\n", name, avg / (iters * 32.0), ms,
           flops / (ms * 1e-3) / 1e12, THREADS / 256);
}

// bf16 32x32x16 (32 cycles): MODE 0 MFMAs only; 1: + one v_add per two MFMAs; 2: + one ds_read_b128 + s_waitcnt per two MFMAs (no VALU)
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ __launch_bounds__(256) void kb(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.f;
    __syncthreads();
    f32x16 c0, c1;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    asm volatile("" : "+a"(c0), "+a"(c1), "+a"(b0), "+a"(b1));
    uint32_t base = (uint32_t)(uintptr_t)lds + (threadIdx.x & 63) * 16, off = 1024;
    f32x4 A[4];
    for (int u = 0; u < 4; ++u) A[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (MODE == 2) for (int u = 0; u < 3; ++u) asm volatile("ds_read_b128 %0, %1" : "=v"(A[u]) : "v"(base + u * 1024) : "memory");
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 1) { uint32_t ad; asm volatile("v_add_u32 %0, %1, %2" : "=v"(ad) : "v"(base), "v"(off)); asm volatile("" :: "v"(ad)); }
            if (MODE == 2) { asm volatile("ds_read_b128 %0, %1 offset:256" : "=v"(A[(u + 3) & 3]) : "v"(base) : "memory");
                             asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[u & 3]) :: "memory"); }
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c0) : "v"(A[u & 3]), "a"(b0));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c1) : "v"(A[u & 3]), "a"(b1));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[3] + A[0][0] + A[1][1] + A[2][2] + A[3][3];
}
template <int MODE> void runb(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 1000;
    kb<MODE><<<256, 256>>>(out, cyc, 10);
    kb<MODE><<<256, 256>>>(out, cyc, iters);
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
    run<6>("MFMAs only on RANDOM operands", out, cyc, rnd);
    run<7>("two ds_read_b128 + one s_waitcnt per eight MFMAs on RANDOM operands (the loop of k_conv3x3_c64_rows_f32)", out, cyc, rnd);
    run<0, 512>("f32 MFMAs only, TWO waves per SIMD", out, cyc, rnd);
    run<1, 512>("+ one v_add_u32 per four MFMAs, TWO waves per SIMD", out, cyc, rnd);
    run<2, 512>("+ ds_read_b128 + s_waitcnt per four MFMAs, TWO waves per SIMD", out, cyc, rnd);
    runb<0>("32x32x16 bf16 MFMAs only (B in AGPRs, two accumulators)", out, cyc);
    runb<1>("+ one v_add_u32 per two MFMAs", out, cyc);
    runb<2>("+ one ds_read_b128 + s_waitcnt lgkmcnt(3) per two MFMAs (no VALU)", out, cyc);
    return 0;
}
