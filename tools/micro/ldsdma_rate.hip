// Micro-benchmark: sustained rate of 16-byte LDS-DMA (global_load_lds_dwordx4) and of plain 16-byte global loads per CU, sources L2-hot.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/ldsdma_rate.hip -o tools/micro/ldsdma_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }

// MODE 0: LDS-DMA 16 B/lane; 1: plain global_load_dwordx4 into VGPRs (summed); 2: LDS-DMA 4 B/lane
// MODE 3: as 0, each DMA followed by 16 dependent v_fma (64 cycles of VALU); 4: the VALU work alone; 5: as 3 with 4 MFMA 32x32x16 (128 cycles) instead; 6: MFMA alone
template <int MODE>
__global__ __launch_bounds__(512) void k_rate(const float* __restrict__ src, float* __restrict__ out, unsigned long long* cyc, int iters, int span_kb) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = blockDim.x >> 6;
    // each block walks its own `span_kb` window (L2-resident after the first pass); a wave-instruction reads 1 KB contiguous
    const char* base = reinterpret_cast<const char*>(src) + (size_t)(blockIdx.x % 64) * span_kb * 1024;
    const int pieces = span_kb;                       // 1 KB pieces in the window
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    f32x16 macc = {0}; bf16x8 ma = {0}, mb = {0};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    int pc = wave;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const char* g = base + (size_t)pc * 1024 + lane * 16;
            if (MODE == 0 || MODE == 3 || MODE == 5) __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(lds + (wave * 4 + u) * 256), 16, 0, 0);
            else if (MODE == 2) __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(lds + (wave * 4 + u) * 256), 4, 0, 0);
            else if (MODE == 1) { const float4 v = *reinterpret_cast<const float4*>(g); acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
            if (MODE == 3 || MODE == 4) {
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(acc.x));
            }
            if (MODE == 5 || MODE == 6) {
#pragma unroll
                for (int k = 0; k < 4; ++k) macc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ma, mb, macc, 0, 0, 0);
            }
            pc += nw; if (pc >= pieces) pc -= pieces;
        }
        if (MODE != 1) wait_vmcnt<8>();
    }
    if (MODE != 1) wait_vmcnt<0>();
    if (MODE >= 5) acc.x += macc[0] + macc[5];
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (MODE == 1 || MODE >= 3) out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    else if (lds[threadIdx.x] == 123.456f) out[0] = 1.f;
}

template <int MODE>
static void run(const char* name, const float* src, float* out, unsigned long long* cyc, int threads, int span_kb) {
    const int blocks = 256, iters = 2000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    k_rate<MODE><<<blocks, threads, 65536>>>(src, out, cyc, 50, span_kb);
    CHECK(hipEventRecord(e0));
    k_rate<MODE><<<blocks, threads, 65536>>>(src, out, cyc, iters, span_kb);
    CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h(blocks);
    CHECK(hipMemcpy(h.data(), cyc, blocks * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double avg = 0; for (auto v : h) avg += (double)v; avg /= blocks;
    const double instr = (double)iters * 4 * (threads / 64);                   // wave-instructions per block
    const double bytes = instr * (MODE == 2 ? 256.0 : 1024.0);
    printf("%-28s waves/CU %d window %3d KB: %7.1f cycles per wave-instruction per CU, %6.1f B/clk/CU, %6.2f TB/s chip (%.3f ms, clock %.2f GHz)\n", name,
           threads / 64, span_kb, avg / instr, bytes / avg, bytes * blocks / (ms * 1e-3) / 1e12, ms, avg / (ms * 1e-3) / 1e9);
}

int main() {
    float *src, *out; unsigned long long* cyc;
    CHECK(hipMalloc(&src, 64u << 20)); CHECK(hipMemset(src, 0, 64u << 20));
    CHECK(hipMalloc(&out, 4u << 20)); CHECK(hipMalloc(&cyc, 4096));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<6>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int threads : {64, 256}) {
        run<3>("DMA + 16 v_fma each", src, out, cyc, threads, 64);
        run<4>("16 v_fma alone", src, out, cyc, threads, 64);
        run<5>("DMA + 4 MFMA each", src, out, cyc, threads, 64);
        run<6>("4 MFMA alone", src, out, cyc, threads, 64);
    }
    for (int span : {64}) {
        for (int threads : {64, 128, 256, 512}) run<0>("LDS-DMA 16 B/lane", src, out, cyc, threads, span);
        for (int threads : {256, 512}) run<1>("global_load_dwordx4 -> VGPR", src, out, cyc, threads, span);
        run<2>("LDS-DMA 4 B/lane", src, out, cyc, 512, span);
    }
    return 0;
}
