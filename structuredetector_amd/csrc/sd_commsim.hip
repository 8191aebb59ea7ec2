// Stand-in for an RCCL all-reduce on ONE GPU: the footprint a collective has on the chip while the backward is running beside it.
// (No reference counterpart: the reference is single-device, src/sdnet/model/trainer.py:113-124; SURVEY.md 8e / section 0: data-parallel is new work.)
//
// An RCCL ring / tree kernel is a PERSISTENT launch of a few workgroups (one or two per channel: 16-32 on an 8-GPU xGMI node) of 256
// threads with a few KB of LDS each, which occupy their CUs for the whole transfer -- milliseconds, bound by the links (7 x ~50 GB/s per
// direction), not by HBM -- while streaming 2 (N - 1) / N x the buffer through local memory.  `k_comm_sim` reproduces exactly that shape on
// one device: `workgroups` blocks of 256 threads and 8 KB of LDS copy `bytes` from src to dst in 64 KB chunks (16 bytes per lane per
// access, through LDS like RCCL's reduce-copy), and after every chunk spin on the 100 MHz wall clock until the block's share of `gbps`
// allows the next one: the launch lasts bytes / gbps whatever the HBM could do.  With it bench.py can say, BEFORE eight ranks exist, how
// much a training step stretches when five such launches run on a side stream beside the backward kernels -- whose 512-thread blocks
// own whole CUs (152 KB of LDS, `__launch_bounds__(512, 1)`): a CU held by a communication block is a CU a tile round has to wait for.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/sdnet_hip.h"
#include "sd_common.h"

namespace sd {

constexpr int CS_THREADS = 256, CS_CHUNK = 64 * 1024;

__global__ __launch_bounds__(CS_THREADS) void k_comm_sim(const uint4* __restrict__ src, uint4* __restrict__ dst, uint64_t src_elems, uint64_t move_elems,
                                                         uint64_t ticks_per_chunk) {
    __shared__ uint4 stage[CS_THREADS * 2];                               // 8 KB, touched like a reduce-copy's staging buffer
    const uint64_t chunk_elems = CS_CHUNK / 16;
    const uint64_t nchunks = (move_elems + chunk_elems - 1) / chunk_elems;
    const uint64_t t0 = wall_clock64();
    uint64_t done = 0;
    for (uint64_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const uint64_t base = c * chunk_elems;
        for (uint64_t i = threadIdx.x; i < chunk_elems && base + i < move_elems; i += 2 * CS_THREADS) {
            const uint64_t e0 = (base + i) % src_elems, e1 = (base + i + CS_THREADS) % src_elems;      // (the source wraps: 2 (N - 1) / N x the bucket is moved)
            const bool second = i + CS_THREADS < chunk_elems && base + i + CS_THREADS < move_elems;
            stage[threadIdx.x] = src[e0];
            if (second) stage[CS_THREADS + threadIdx.x] = src[e1];
            dst[e0] = stage[threadIdx.x];
            if (second) dst[e1] = stage[CS_THREADS + threadIdx.x];
        }
        ++done;
        // link-bound: this block may not be ahead of its share of the modelled bus bandwidth (wave-uniform spin on a constant-rate clock;
        // bounded: the loop ends when the clock has advanced, which it always does)
        const uint64_t due = t0 + done * ticks_per_chunk;
        while (wall_clock64() < due) __builtin_amdgcn_s_sleep(8);
    }
}

}  // namespace sd

extern "C" {

int sd_comm_sim_copy(const void* src, void* dst, size_t src_bytes, size_t move_bytes, int workgroups, float gbps, sd_stream_t stream) {
    SD_REQUIRE(src && dst && src_bytes >= 16 && src_bytes % 16 == 0 && move_bytes % 16 == 0, SD_ERR_INVALID,
               "sd_comm_sim_copy: null pointer or sizes that are not multiples of 16 bytes");
    SD_REQUIRE(workgroups >= 1 && workgroups <= 256 && gbps > 0.f, SD_ERR_INVALID, "sd_comm_sim_copy: workgroups in 1 .. 256, gbps > 0");
    SD_REQUIRE(sd::aligned16(src) && sd::aligned16(dst), SD_ERR_ALIGN, "sd_comm_sim_copy: pointers must be 16-byte aligned");
    if (move_bytes == 0) return 0;
    // a block handles every `workgroups`-th chunk: its chunk may start no earlier than (chunks done) x (time the whole launch is allowed per chunk)
    // x workgroups; wall_clock64 ticks at 100 MHz
    const double chunk_seconds = (double)sd::CS_CHUNK / ((double)gbps * 1e9) * workgroups;
    const uint64_t ticks = (uint64_t)(chunk_seconds * 1e8 + 0.5);
    hipLaunchKernelGGL(sd::k_comm_sim, dim3(workgroups), dim3(sd::CS_THREADS), 0, (hipStream_t)stream, (const uint4*)src, (uint4*)dst,
                       (uint64_t)(src_bytes / 16), (uint64_t)(move_bytes / 16), ticks);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
