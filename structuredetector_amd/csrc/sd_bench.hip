// The box's own bf16 MFMA ceiling, measured in the process that reports a fraction of it (bench.py: `frac_of_measured_bf16_stream`).
// (No reference counterpart: measurement infrastructure)  The nominal dense bf16 peak (2.5 PFLOP/s) is not reachable on
// random operands: under its power limit the chip holds ~1.6 PFLOP/s on a bare loop that does nothing but read its operands from LDS and
// multiply (tools/micro/mfma_bf16_shape.hip, `profiles/r04_mfma_bf16_shape_microbench.txt`).  `k_mfma_bf16_stream` is that loop with the
// shape the bf16 conv kernels use (v_mfma_f32_16x16x32_bf16, wave tile 128 x 64, 12 ds_read_b128 per K = 32 step, two waves per SIMD, one
// 512-thread block per CU): every operand is re-read from LDS every step, nothing is loaded from or stored to global memory in the loop.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sdnet_hip.h"
#include "sd_common.h"
#include "sd_mfma.h"

namespace sd {

constexpr int MS_LDS_BYTES = 64 * 1024, MS_BLOCKS = 256, MS_THREADS = 512;

__global__ __launch_bounds__(MS_THREADS, 1) void k_mfma_bf16_stream(const uint32_t* __restrict__ rnd, float* __restrict__ out, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint32_t ms_lds[];
    for (int i = threadIdx.x; i < MS_LDS_BYTES / 4; i += MS_THREADS) ms_lds[i] = rnd[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const char* const base = reinterpret_cast<const char*>(ms_lds) + (wave & 3) * 4096;
    const int r16 = lane & 15, sl = lane >> 4;
    const uint32_t a_off = r16 * 64 + ((sl ^ (((r16 >> 2) & 1) << 1)) << 4);       // conflict-free slot swizzle for this operand layout
    f32x4 acc[8][4];
    for (int mi = 0; mi < 8; ++mi) for (int ni = 0; ni < 4; ++ni) for (int e = 0; e < 4; ++e) acc[mi][ni][e] = 0.f;
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
    float res = 0.f;
    for (int mi = 0; mi < 8; ++mi) for (int ni = 0; ni < 4; ++ni) res += acc[mi][ni][0] + acc[mi][ni][3];
    out[blockIdx.x * MS_THREADS + threadIdx.x] = res;
}

}  // namespace sd

extern "C" {

double sd_mfma_bf16_stream_flops(int iters) { return (double)sd::MS_BLOCKS * (sd::MS_THREADS / 64) * (double)iters * 2.0 * 128 * 64 * 32; }

int sd_mfma_bf16_stream(const void* operands64k, float* out, int iters, sd_stream_t stream) {
    SD_REQUIRE(operands64k && out && iters > 0, SD_ERR_INVALID, "sd_mfma_bf16_stream: null pointer or iters <= 0");
    static thread_local bool raised = false;
    if (!raised) {
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(sd::k_mfma_bf16_stream), hipFuncAttributeMaxDynamicSharedMemorySize, sd::MS_LDS_BYTES));
        raised = true;
    }
    hipLaunchKernelGGL(sd::k_mfma_bf16_stream, dim3(sd::MS_BLOCKS), dim3(sd::MS_THREADS), sd::MS_LDS_BYTES, (hipStream_t)stream,
                       (const uint32_t*)operands64k, out, iters);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
