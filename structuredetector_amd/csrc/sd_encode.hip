// Ground-truth heatmap rendering for a whole batch on gfx950.
// Follows Encode.__call__, src/sdnet/data/transforms.py:143,160-161,173-174 and gaussian_2d,
// src/sdnet/utils/utils.py:418-419: every keypoint is a FULL-FRAME Gaussian merged with an
// elementwise max.  exp is monotone and the squared distance is an exact integer, so
// max_k exp(-d_k^2 / 2s^2) == exp(-(min_k d_k^2) / 2s^2): one expf per pixel instead of one per
// (pixel, keypoint).  HBM-bound: (M+N)*h*w*4 bytes written per image, nothing re-read.
#include "sd_common.h"

namespace sd {

__global__ __launch_bounds__(256) void k_render_targets(const int32_t* __restrict__ cx, const int32_t* __restrict__ cy,
                                                         const int32_t* __restrict__ chan_ptr, int C, int h, int w,
                                                         float two_sigma2, float* __restrict__ out) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int q = blockIdx.x * 256 + threadIdx.x;          // one float4 (4 consecutive x) per thread
    const int qpr = w >> 2;                                // float4 per row (w % 4 == 0)
    if (q >= qpr * h) return;
    const int y = q / qpr, x0 = (q - y * qpr) << 2;
    const int beg = chan_ptr[b * C + c], end = chan_ptr[b * C + c + 1];   // wave-uniform -> scalar loads
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (end > beg) {
        int d0 = INT_MAX, d1 = INT_MAX, d2 = INT_MAX, d3 = INT_MAX;
        for (int k = beg; k < end; ++k) {
            const int dy = y - cy[k], dx = x0 - cx[k];
            const int dy2 = dy * dy;
            d0 = min(d0, dx * dx + dy2);
            d1 = min(d1, (dx + 1) * (dx + 1) + dy2);
            d2 = min(d2, (dx + 2) * (dx + 2) + dy2);
            d3 = min(d3, (dx + 3) * (dx + 3) + dy2);
        }
        // fp32(-(d^2)) / fp32(2 sigma^2), true division, then fp32 exp
        v.x = expf((float)(-d0) / two_sigma2);
        v.y = expf((float)(-d1) / two_sigma2);
        v.z = expf((float)(-d2) / two_sigma2);
        v.w = expf((float)(-d3) / two_sigma2);
    }
    reinterpret_cast<float4*>(out + ((int64_t)b * C + c) * h * w)[q] = v;
}

}  // namespace sd

extern "C" int sd_render_targets(const int32_t* cx, const int32_t* cy, const int32_t* chan_ptr, int B, int C, int h, int w,
                                 float two_sigma2, float* out, sd_stream_t stream) {
    using namespace sd;
    SD_REQUIRE(chan_ptr && out && B > 0 && C > 0 && h > 0 && w > 0, SD_ERR_INVALID, "sd_render_targets: bad arguments");
    SD_REQUIRE(w % 4 == 0, SD_ERR_INVALID, "sd_render_targets: w must be a multiple of 4 (got %d)", w);
    SD_REQUIRE(h <= 16384 && w <= 16384, SD_ERR_INVALID, "sd_render_targets: map too large for int32 squared distances");
    SD_REQUIRE(two_sigma2 > 0.f, SD_ERR_INVALID, "sd_render_targets: two_sigma2 must be > 0");
    SD_REQUIRE(aligned16(out), SD_ERR_ALIGN, "sd_render_targets: out must be 16-byte aligned");
    hipLaunchKernelGGL(k_render_targets, dim3(cdiv((int64_t)h * w / 4, 256), C, B), dim3(256), 0, (hipStream_t)stream, cx, cy,
                       chan_ptr, C, h, w, two_sigma2, out);
    SD_LAUNCH_CHECK();
    return 0;
}
