// Micro-benchmark: does the chip hold a higher clock on v_mfma_f32_16x16x4_f32 than on v_mfma_f32_32x32x2_f32 at the same flops (the bf16
// shapes do: tools/micro/mfma_bf16_shape.hip)?  Same wave tile (128 x 64 outputs, 128 accumulator registers), operands re-read from LDS by
// ds_read_b128 (one 16-byte slot = this lane's four k values), random and zero data, one and two waves per SIMD, wall clock after warm-up.
//   shape 0: 32x32x2, per K = 4 step 4 A + 2 B reads and 32 MFMAs x 64 cycles ... no: 8 tiles x 4 k = 32 MFMAs (2048 cycles)
//   shape 1: 16x16x4, per K = 4 step 8 A + 4 B reads (4 bytes per lane each) and 32 MFMAs x 32 cycles (1024 cycles) -- half the K per read
// Both loops are normalised to FLOP/s.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/mfma_f32_shape.hip -o tools/micro/mfma_f32_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int LDS_BYTES = 64 * 1024;

// SHAPE 0: wave tile 128 x 64 = 4 x 2 tiles of 32 x 32; a lane's ds_read_b128 = 4 consecutive k of its row: K = 8 per step (two lane halves)
// SHAPE 1: wave tile 128 x 64 = 8 x 4 tiles of 16 x 16; a lane's ds_read_b128 = 4 consecutive k of its row: K = 16 per step (four lane groups)
template <int SHAPE>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters, const float* rnd) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x; i < LDS_BYTES / 4; i += blockDim.x) lds[i] = rnd[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* const base = reinterpret_cast<const char*>(lds) + (wave & 3) * 4096;
    float res = 0.f;
    if constexpr (SHAPE == 0) {
        const int fr = lane & 31, fh = lane >> 5;
        const uint32_t a_off = fr * 32 + ((fh ^ ((fr >> 3) & 1)) << 4);          // rows of 32 bytes (8 k), slot = k half
        f32x16 acc[4][2];
        for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 2; ++ni) for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const char* s = base + (it & 7) * 2048;
            f32x4 a[4], b[2];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(s + mi * 1024 + a_off);
#pragma unroll
            for (int ni = 0; ni < 2; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(s + 8192 + ni * 1024 + a_off);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 4; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
        }
        for (int mi = 0; mi < 4; ++mi) for (int ni = 0; ni < 2; ++ni) res += acc[mi][ni][0] + acc[mi][ni][7];
    } else {
        const int r16 = lane & 15, sl = lane >> 4;
        const uint32_t a_off = r16 * 64 + ((sl ^ (((r16 >> 2) & 1) << 1)) << 4);   // rows of 64 bytes (16 k), slot = k quarter
        f32x4 acc[8][4];
        for (int mi = 0; mi < 8; ++mi) for (int ni = 0; ni < 4; ++ni) for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const char* s = base + (it & 7) * 2048;
            f32x4 a[8], b[4];
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) a[mi] = *reinterpret_cast<const f32x4*>(s + mi * 1024 + a_off);
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) b[ni] = *reinterpret_cast<const f32x4*>(s + 8192 + ni * 1024 + a_off);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
#pragma unroll
                    for (int mi = 0; mi < 8; ++mi) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][kk], b[ni][kk], acc[mi][ni], 0, 0, 0);
        }
        for (int mi = 0; mi < 8; ++mi) for (int ni = 0; ni < 4; ++ni) res += acc[mi][ni][0] + acc[mi][ni][3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = res;
}

template <int SHAPE> double run(const char* name, int threads, float* out, const float* rnd, double seconds) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    // per step and wave: shape 0 = 2 * 128 * 64 * 8 flop (32 MFMAs x 64 cycles), shape 1 = 2 * 128 * 64 * 16 flop (128 MFMAs x 32 cycles)
    const double flop_step = 2.0 * 128 * 64 * (SHAPE == 0 ? 8 : 16);
    const int iters = SHAPE == 0 ? 10000 : 5000;          // ~20 M MFMA cycles ~ 10 ms per launch
    k<SHAPE><<<256, threads, LDS_BYTES>>>(out, 100, rnd);
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = (int)(seconds / 0.010) + 1;
    for (int r = 0; r < reps; ++r) k<SHAPE><<<256, threads, LDS_BYTES>>>(out, iters, rnd);
    CHECK(hipEventRecord(e0));
    k<SHAPE><<<256, threads, LDS_BYTES>>>(out, iters, rnd);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = 256.0 * (threads / 64) * (double)iters * flop_step / (ms * 1e-3) / 1e12;
    printf("%-44s %d waves/SIMD  wall %.3f ms = %7.1f TFLOP/s\n", name, threads / 256, ms, tf);
    return tf;
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
    float* out; float* rnd;
    CHECK(hipMalloc(&out, 256 * 512 * 4)); CHECK(hipMalloc(&rnd, LDS_BYTES));
    std::vector<float> h(LDS_BYTES / 4);
    for (int pass = 0; pass < 2; ++pass) {
        srand(1);
        for (auto& v : h) v = pass == 0 ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f;
        CHECK(hipMemcpy(rnd, h.data(), LDS_BYTES, hipMemcpyHostToDevice));
        printf("---- %s operands\n", pass == 0 ? "ZERO" : "RANDOM");
        for (int threads : {256, 512}) {
            const double t0 = run<0>("v_mfma_f32_32x32x2_f32, wave tile 128x64", threads, out, rnd, seconds);
            const double t1 = run<1>("v_mfma_f32_16x16x4_f32, wave tile 128x64", threads, out, rnd, seconds);
            const double t0b = run<0>("v_mfma_f32_32x32x2_f32 (again)", threads, out, rnd, seconds);
            printf("     16x16x4 / 32x32x2 by wall clock: %.3f (%.3f against the second run)\n", t1 / t0, t1 / t0b);
        }
    }
    return 0;
}
