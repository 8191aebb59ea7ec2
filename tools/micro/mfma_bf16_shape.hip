// Micro-benchmark: which bf16 MFMA shape does the chip run faster BY WALL CLOCK on random data at the same wave tile (128 x 64 outputs
// per wave, all operands re-read from LDS by ds_read_b128 every K = 32 step -- the inner loop of k_conv3x3_bf16_pp without its DMA)?
//   shape 0: v_mfma_f32_32x32x16_bf16, 16 per step (32 cycles each)     shape 1: v_mfma_f32_16x16x32_bf16, 32 per step (16 cycles each)
// Both read 12 x ds_read_b128 per step and keep 128 accumulator registers.  The guide (MI355X_MICROARCH.md, DVFS give-back item 7) reports
// 1.12-1.15 x the FLOP/s for the 16x16x32 loop at equal cycles; this is the check on the boxes this repository is measured on.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_bf16_shape.hip -o tools/micro/mfma_bf16_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int LDS_BYTES = 64 * 1024;

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, const uint32_t* rnd) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += blockDim.x) lds[i] = rnd[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* const base = reinterpret_cast<const char*>(lds) + (wave & 3) * 4096;
    float res = 0.f;
    unsigned long long t0, t1;
    if constexpr (SHAPE == 0) {
        const int fr = lane & 31, fh = lane >> 5;
        const uint32_t a_off = fr * 64 + ((fh ^ ((fr >> 2) & 3)) << 4);
        f32x16 acc[4][2];
        for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 2; ++ni) for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            const char* s = base + (it & 7) * 2048;
            bf16x8 a[2][4], b[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int mi = 0; mi < 4; ++mi) a[h][mi] = *reinterpret_cast<const bf16x8*>(s + ((mi * 2048 + a_off) ^ (h * 32)));
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) b[h][ni] = *reinterpret_cast<const bf16x8*>(s + 8192 + ((ni * 2048 + a_off) ^ (h * 32)));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][mi], b[h][ni], acc[mi][ni], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 2; ++ni) res += acc[mi][ni][0] + acc[mi][ni][7];
    } else {
        const int r16 = lane & 15, sl = lane >> 4;
        // slot swizzle by bit 3 of the row: conflict-free for the ds_read_b128 lane groups of this operand layout (rows 0-15 x 4 slots)
        const uint32_t a_off = r16 * 64 + ((sl ^ (((r16 >> 2) & 1) << 1)) << 4);
        f32x4 acc[8][4];
        for (int mi = 0; mi < 8; ++mi) for (int ni = 0; ni < 4; ++ni) for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
            const char* s = base + (it & 7) * 2048;
            bf16x8 a[8], b[4];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) a[mi] = *reinterpret_cast<const bf16x8*>(s + mi * 1024 + a_off);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const bf16x8*>(s + 8192 + ni * 1024 + a_off);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        t1 = __builtin_readcyclecounter();
        for (int mi = 0; mi < 8; ++mi) for (int ni = 0; ni < 4; ++ni) res += acc[mi][ni][0] + acc[mi][ni][3];
    }
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int SHAPE> double run(const char* name, int threads, float* out, unsigned long long* cyc, const uint32_t* rnd, double seconds) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    const int iters = 40000;       // 40000 steps x 512 MFMA cycles = 20.5 M cycles ~ 10 ms per launch
    k<SHAPE><<<256, threads, LDS_BYTES>>>(out, cyc, 100, rnd);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // warm the power state: back-to-back launches for `seconds`, the LAST launch is the one timed
    const int reps = (int)(seconds / 0.010) + 1;
    for (int r = 0; r < reps; ++r) k<SHAPE><<<256, threads, LDS_BYTES>>>(out, cyc, iters, rnd);
    CHECK(hipEventRecord(e0));
    k<SHAPE><<<256, threads, LDS_BYTES>>>(out, cyc, iters, rnd);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(256); CHECK(hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= 256;
    const double flops = 256.0 * (threads / 64) * (double)iters * 2.0 * 128 * 64 * 32;
    const double tf = flops / (ms * 1e-3) / 1e12;
    printf("%-44s %d waves/SIMD  %7.1f counter ticks per K=32 step (512 MFMA cycles)  wall %.3f ms = %7.1f TFLOP/s\n", name, threads / 256, avg / iters, ms, tf);
    return tf;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    float* out; unsigned long long* cyc; uint32_t* rnd;
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&cyc, 256 * 8)); CHECK(hipMalloc(&rnd, LDS_BYTES));
    std::vector<uint32_t> h(LDS_BYTES / 4);
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (auto& v : h) {
            if (pass == 0) { v = 0; continue; }
            // two bf16 uniform in [-1, 1): sign, exponent 119..126, random mantissa
            uint32_t w = 0;
            for (int q = 0; q < 2; ++q) { const uint32_t sgn = rand() & 1, ex = 119 + (rand() & 7), man = rand() & 127; w |= ((sgn << 15) | (ex << 7) | man) << (16 * q); }
            v = w;
        }
        CHECK(hipMemcpy(rnd, h.data(), LDS_BYTES, hipMemcpyHostToDevice));
        printf("---- %s operands\n", pass == 0 ? "ZERO" : "RANDOM");
        for (int threads : {256, 512}) {
            const double t0 = run<0>("v_mfma_f32_32x32x16_bf16, wave tile 128x64", threads, out, cyc, rnd, seconds);
            const double t1 = run<1>("v_mfma_f32_16x16x32_bf16, wave tile 128x64", threads, out, cyc, rnd, seconds);
            const double t0b = run<0>("v_mfma_f32_32x32x16_bf16 (again)", threads, out, cyc, rnd, seconds);
            printf("     16x16x32 / 32x32x16 by wall clock: %.3f (%.3f against the second run)\n", t1 / t0, t1 / t0b);
        }
    }
    return 0;
}
