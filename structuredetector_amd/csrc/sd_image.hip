// Input pipeline on gfx950: resize + horizontal / vertical flip + ImageNet normalisation of a batch of
// decoded RGB images, replacing the PIL / torchvision chain the reference runs in its DataLoader workers
// (src/sdnet/data/transforms.py:9-35 flips, :47-60 Resize, :108-118 Normalize, composed at :217-234 and :255-261).
//
// Resize: torchvision's F.resize on a PIL image is PIL's Image.resize(BILINEAR) = a separable resampling with a triangle filter
// whose support grows with the scale factor (antialiased when shrinking), computed in 8-bit fixed point: horizontal pass into an
// 8-bit image, then the vertical pass (Pillow, src/libImaging/Resample.c: precompute_coeffs + normalize_coeffs_8bpc with
// PRECISION_BITS = 22, ImagingResampleHorizontal_8bpc / Vertical_8bpc; Pillow 12.2 is the dependency present in the image).
// The coefficient tables are computed on the host exactly like Pillow does (double -> 22-bit fixed point); the two kernels
// below reproduce the integer accumulation and the clip, so the resized bytes are IDENTICAL to PIL's -- which makes the
// normalised float image bit-identical to the reference's (to_tensor = u8 / 255 in fp32, then (x - mean) / std in fp32: two
// correctly rounded fp32 operations each; contraction is off in this file).
// HBM-bound byte work: reads B*Hin*Win*3, writes B*Hin*Wout*3 (8-bit intermediate) + B*3*Hout*Wout*4.
#pragma clang fp contract(off)
#include "sd_common.h"

namespace sd {

constexpr int PRECISION_BITS = 32 - 8 - 2;

__device__ __forceinline__ uint8_t clip8(int v) {
    if (v >= (1 << PRECISION_BITS << 8)) return 255;
    if (v <= 0) return 0;
    return (uint8_t)(v >> PRECISION_BITS);
}

// horizontal pass: in (B, Hin, Win, 3) u8 -> tmp (B, Hin, Wout, 3) u8.  bounds[x] = {xmin, count}, kk[x * ksize + i] fixed-point weights
__global__ __launch_bounds__(256) void k_resample_h(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int Hin, int Win, int Wout,
                                                     const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int64_t rows) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one output pixel (3 bytes) per thread
    if (i >= rows * Wout) return;
    const int64_t row = i / Wout;
    const int x = (int)(i - row * Wout);
    const int xmin = bounds[2 * x], n = bounds[2 * x + 1];
    const int* k = kk + (int64_t)x * ksize;
    const uint8_t* src = in + (row * Win + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < n; ++t) {
        const int w = k[t];
        s0 += src[3 * t + 0] * w; s1 += src[3 * t + 1] * w; s2 += src[3 * t + 2] * w;
    }
    uint8_t* dst = out + i * 3;
    dst[0] = clip8(s0); dst[1] = clip8(s1); dst[2] = clip8(s2);
}

// vertical pass + flips + to_tensor + Normalize: tmp (B, Hin, Wout, 3) u8 -> out (B, 3, Hout, Wout) fp32 NCHW
__global__ __launch_bounds__(256) void k_resample_v_norm(const uint8_t* __restrict__ in, float* __restrict__ out, int Hin, int Hout, int Wout,
                                                          const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                          const uint8_t* __restrict__ flips, float m0, float m1, float m2, float d0, float d1,
                                                          float d2, int B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one output pixel (three planes) per thread
    if (i >= (int64_t)B * Hout * Wout) return;
    const int x = (int)(i % Wout);
    const int64_t t = i / Wout;
    const int y = (int)(t % Hout), b = (int)(t / Hout);
    const int f = flips ? flips[b] : 0;                                  // bit 0: horizontal flip, bit 1: vertical flip (applied AFTER the resize)
    const int sx = (f & 1) ? Wout - 1 - x : x, sy = (f & 2) ? Hout - 1 - y : y;
    const int ymin = bounds[2 * sy], n = bounds[2 * sy + 1];
    const int* k = kk + (int64_t)sy * ksize;
    const uint8_t* src = in + (((int64_t)b * Hin + ymin) * Wout + sx) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int r = 0; r < n; ++r) {
        const int w = k[r];
        const uint8_t* px = src + (int64_t)r * Wout * 3;
        s0 += px[0] * w; s1 += px[1] * w; s2 += px[2] * w;
    }
    const int64_t plane = (int64_t)Hout * Wout;
    float* o = out + (int64_t)b * 3 * plane + (int64_t)y * Wout + x;
    o[0] = ((float)clip8(s0) / 255.0f - m0) / d0;                        // to_tensor (u8 / 255), then Normalize: (x - mean) / std
    o[plane] = ((float)clip8(s1) / 255.0f - m1) / d1;
    o[2 * plane] = ((float)clip8(s2) / 255.0f - m2) / d2;
}

}  // namespace sd

using namespace sd;

extern "C" {

size_t sd_preprocess_workspace_bytes(int B, int Hin, int Win, int Wout) { return align_up((size_t)B * Hin * Wout * 3, 256); }

int sd_preprocess_images(const uint8_t* images, int B, int Hin, int Win, int Hout, int Wout, const int* h_bounds, const int* h_kk,
                         int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize, const uint8_t* flips, const float* mean3,
                         const float* std3, float* out, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    SD_REQUIRE(images && out && h_bounds && h_kk && v_bounds && v_kk && mean3 && std3 && workspace, SD_ERR_INVALID,
               "sd_preprocess_images: null pointer");
    SD_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && h_ksize > 0 && v_ksize > 0, SD_ERR_INVALID,
               "sd_preprocess_images: bad shape");
    SD_REQUIRE((int64_t)B * std::max(Hin, Hout) * std::max(Win, Wout) * 3 < (1ll << 40), SD_ERR_INVALID, "sd_preprocess_images: batch too large");
    SD_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, SD_ERR_INVALID, "sd_preprocess_images: zero std");
    SD_REQUIRE(workspace_bytes >= sd_preprocess_workspace_bytes(B, Hin, Win, Wout), SD_ERR_WORKSPACE, "sd_preprocess_images: workspace %zu < %zu",
               workspace_bytes, sd_preprocess_workspace_bytes(B, Hin, Win, Wout));
    hipStream_t st = (hipStream_t)stream;
    uint8_t* tmp = reinterpret_cast<uint8_t*>(workspace);
    const int64_t rows = (int64_t)B * Hin;
    hipLaunchKernelGGL(k_resample_h, dim3(cdiv(rows * Wout, 256)), dim3(256), 0, st, images, tmp, Hin, Win, Wout, h_bounds, h_kk, h_ksize, rows);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_resample_v_norm, dim3(cdiv((int64_t)B * Hout * Wout, 256)), dim3(256), 0, st, tmp, out, Hin, Hout, Wout, v_bounds, v_kk,
                       v_ksize, flips, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], B);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
