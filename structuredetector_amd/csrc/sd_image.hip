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

// ---------------------------------------------------------------------------------------------------------------------------------
// RandomColorJitter (src/sdnet/data/transforms.py:37-47, in the default training chain :217-226) on the RESIZED 8-bit image, before the
// flips and Normalize: torchvision's ColorJitter on a PIL image = Pillow's ImageEnhance.Brightness / Contrast / Color and an HSV round
// trip for the hue, applied in a random order per image.  The byte arithmetic of Pillow (libImaging/Blend.c: float blend, truncated
// inside [0, 1], clipped outside; Convert.c: rgb2l, rgb2hsv_row, hsv2rgb with their float / double mix) is reproduced exactly --
// oracle/pil_photometric.py is the restatement pinned against Pillow over all 2^24 colours, this is the same arithmetic on the device
// (contraction is off in this file; float and double divisions are correctly rounded).
//   order word: bits 0-7 = the four op ids (2 bits each, first op in bits 0-1: 0 brightness, 1 contrast, 2 saturation, 3 hue),
//               bits 8-15 = the hue shift byte uint8(hue_factor * 255); factors = {brightness, contrast, saturation} as floats.
// Contrast blends with the image's mean grey level AT THAT POINT of the op order, a reduction over the whole image: k_jitter_lsum sums
// the grey level of every pixel after the ops that precede the contrast op (integer atomics: exact, order-independent).
// ---------------------------------------------------------------------------------------------------------------------------------
struct Rgb8 { int r, g, b; };

__device__ __forceinline__ int pil_l(const Rgb8& c) { return (c.r * 19595 + c.g * 38470 + c.b * 7471 + 0x8000) >> 16; }
__device__ __forceinline__ int pil_blend1(int d, int x, float a, bool inside) {
    const float t = (float)d + a * (float)(x - d);
    if (inside) return (int)t;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}
__device__ __forceinline__ Rgb8 pil_blend(const Rgb8& d, const Rgb8& x, float a) {
    const bool inside = a >= 0.0f && a <= 1.0f;
    return Rgb8{pil_blend1(d.r, x.r, a, inside), pil_blend1(d.g, x.g, a, inside), pil_blend1(d.b, x.b, a, inside)};
}
__device__ __forceinline__ int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
__device__ __forceinline__ Rgb8 pil_hue(const Rgb8& c, int shift) {
    const int maxc = max(c.r, max(c.g, c.b)), minc = min(c.r, min(c.g, c.b));
    int uh = 0, us = 0;
    if (minc != maxc) {                                        // rgb2hsv_row
        const float cr = (float)(maxc - minc);
        const float sat = cr / (float)maxc;
        const float rc = (float)(maxc - c.r) / cr, gc = (float)(maxc - c.g) / cr, bc = (float)(maxc - c.b) / cr;
        float h;
        if (c.r == maxc) h = bc - gc;
        else if (c.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
        else h = (float)(4.0 + (double)gc - (double)rc);
        const double hd = (double)h / 6.0 + 1.0;
        h = (float)(hd - floor(hd));                           // fmod(x, 1.0) of a non-negative double
        uh = clip255((int)((double)h * 255.0));
        us = clip255((int)((double)sat * 255.0));
    }
    uh = (uh + shift) & 255;                                   // uint8 wrap-around (torchvision adjust_hue)
    if (us == 0) return Rgb8{maxc, maxc, maxc};                // hsv2rgb
    const double hf = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(hf);
    const float f = (float)(hf - (double)(float)i);
    const float fs = (float)((double)(float)us / 255.0);
    const double vf = (double)(float)maxc;
    const int p = clip255((int)floor(vf * (1.0 - (double)fs) + 0.5));
    const int q = clip255((int)floor(vf * (1.0 - (double)fs * (double)f) + 0.5));
    const int t = clip255((int)floor(vf * (1.0 - (double)fs * (1.0 - (double)f)) + 0.5));
    switch (i % 6) {
        case 0: return Rgb8{maxc, t, p};
        case 1: return Rgb8{q, maxc, p};
        case 2: return Rgb8{p, maxc, t};
        case 3: return Rgb8{p, q, maxc};
        case 4: return Rgb8{t, p, maxc};
        default: return Rgb8{maxc, p, q};
    }
}
// the ops of one image in order; stops BEFORE the contrast op when mean < 0 (the reduction pass)
__device__ __forceinline__ Rgb8 jitter_pixel(Rgb8 c, int order, float fb, float fc, float fs, int mean) {
    const int shift = (order >> 8) & 255;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int op = (order >> (2 * k)) & 3;
        if (op == 0) c = pil_blend(Rgb8{0, 0, 0}, c, fb);
        else if (op == 1) { if (mean < 0) return c; c = pil_blend(Rgb8{mean, mean, mean}, c, fc); }
        else if (op == 2) { const int l = pil_l(c); c = pil_blend(Rgb8{l, l, l}, c, fs); }
        else c = pil_hue(c, shift);
    }
    return c;
}

// vertical pass only: tmp (B, Hin, Wout, 3) u8 -> img (B, Hout, Wout, 3) u8 (the resized image the jitter works on)
__global__ __launch_bounds__(256) void k_resample_v_u8(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, int Hin, int Hout, int Wout,
                                                        const int* __restrict__ bounds, const int* __restrict__ kk, int ksize, int B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * Hout * Wout) return;
    const int x = (int)(i % Wout);
    const int64_t t = i / Wout;
    const int y = (int)(t % Hout), b = (int)(t / Hout);
    const int ymin = bounds[2 * y], n = bounds[2 * y + 1];
    const int* k = kk + (int64_t)y * ksize;
    const uint8_t* src = in + (((int64_t)b * Hin + ymin) * Wout + x) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int r = 0; r < n; ++r) {
        const int w = k[r];
        const uint8_t* px = src + (int64_t)r * Wout * 3;
        s0 += px[0] * w; s1 += px[1] * w; s2 += px[2] * w;
    }
    uint8_t* dst = out + i * 3;
    dst[0] = clip8(s0); dst[1] = clip8(s1); dst[2] = clip8(s2);
}

// grey-level sum of every image after the ops that precede its contrast op: blockIdx.y = image, grid-stride over its pixels
__global__ __launch_bounds__(256) void k_jitter_lsum(const uint8_t* __restrict__ img, int64_t npix, const int* __restrict__ order,
                                                      const float* __restrict__ factors, unsigned long long* __restrict__ lsum) {
    const int b = blockIdx.y;
    const int ord = order[b];
    const float fb = factors[3 * b], fc = factors[3 * b + 1], fs = factors[3 * b + 2];
    const uint8_t* src = img + (int64_t)b * npix * 3;
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (int64_t)gridDim.x * 256) {
        const Rgb8 c = jitter_pixel(Rgb8{src[3 * i], src[3 * i + 1], src[3 * i + 2]}, ord, fb, fc, fs, -1);
        acc += (unsigned)pil_l(c);
    }
    acc = (unsigned long long)wave_sum((double)acc);           // < 2^53: exact in double
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(lsum + b, acc);
}

// jitter + flips + to_tensor + Normalize: img (B, Hout, Wout, 3) u8 -> out (B, 3, Hout, Wout) fp32 NCHW
__global__ __launch_bounds__(256) void k_jitter_norm(const uint8_t* __restrict__ img, float* __restrict__ out, int Hout, int Wout,
                                                      const int* __restrict__ order, const float* __restrict__ factors,
                                                      const unsigned long long* __restrict__ lsum, const uint8_t* __restrict__ flips, float m0,
                                                      float m1, float m2, float d0, float d1, float d2, int B) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t)B * Hout * Wout) return;
    const int x = (int)(i % Wout);
    const int64_t t = i / Wout;
    const int y = (int)(t % Hout), b = (int)(t / Hout);
    const int f = flips ? flips[b] : 0;
    const int sx = (f & 1) ? Wout - 1 - x : x, sy = (f & 2) ? Hout - 1 - y : y;
    const uint8_t* px = img + (((int64_t)b * Hout + sy) * Wout + sx) * 3;
    const int64_t plane = (int64_t)Hout * Wout;
    // ImageStat mean: exact sum / count in double, then int(mean + 0.5)
    const int mean = (int)((double)lsum[b] / (double)plane + 0.5);
    const Rgb8 c = jitter_pixel(Rgb8{px[0], px[1], px[2]}, order[b], factors[3 * b], factors[3 * b + 1], factors[3 * b + 2], mean);
    float* o = out + (int64_t)b * 3 * plane + (int64_t)y * Wout + x;
    o[0] = ((float)c.r / 255.0f - m0) / d0;
    o[plane] = ((float)c.g / 255.0f - m1) / d1;
    o[2 * plane] = ((float)c.b / 255.0f - m2) / d2;
}

}  // namespace sd

using namespace sd;

extern "C" {

size_t sd_preprocess_workspace_bytes(int B, int Hin, int Win, int Wout) { return align_up((size_t)B * Hin * Wout * 3, 256); }

size_t sd_preprocess_jitter_workspace_bytes(int B, int Hin, int Win, int Hout, int Wout) {
    return sd_preprocess_workspace_bytes(B, Hin, Win, Wout) + align_up((size_t)B * Hout * Wout * 3, 256) + align_up((size_t)B * 8, 256);
}

static int preprocess_check(const char* what, const uint8_t* images, int B, int Hin, int Win, int Hout, int Wout, const int* h_bounds, const int* h_kk,
                            int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize, const float* mean3, const float* std3, float* out,
                            void* workspace) {
    SD_REQUIRE(images && out && h_bounds && h_kk && v_bounds && v_kk && mean3 && std3 && workspace, SD_ERR_INVALID, "%s: null pointer", what);
    SD_REQUIRE(B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && h_ksize > 0 && v_ksize > 0, SD_ERR_INVALID, "%s: bad shape", what);
    SD_REQUIRE((int64_t)B * std::max(Hin, Hout) * std::max(Win, Wout) * 3 < (1ll << 40), SD_ERR_INVALID, "%s: batch too large", what);
    SD_REQUIRE(std3[0] != 0.f && std3[1] != 0.f && std3[2] != 0.f, SD_ERR_INVALID, "%s: zero std", what);
    return 0;
}

int sd_preprocess_images(const uint8_t* images, int B, int Hin, int Win, int Hout, int Wout, const int* h_bounds, const int* h_kk,
                         int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize, const uint8_t* flips, const float* mean3,
                         const float* std3, float* out, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = preprocess_check("sd_preprocess_images", images, B, Hin, Win, Hout, Wout, h_bounds, h_kk, h_ksize, v_bounds, v_kk, v_ksize, mean3, std3,
                                 out, workspace)) return e;
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

int sd_preprocess_images_jitter(const uint8_t* images, int B, int Hin, int Win, int Hout, int Wout, const int* h_bounds, const int* h_kk,
                                int h_ksize, const int* v_bounds, const int* v_kk, int v_ksize, const uint8_t* flips, const int* jitter_order,
                                const float* jitter_factors, const float* mean3, const float* std3, float* out, void* workspace,
                                size_t workspace_bytes, sd_stream_t stream) {
    if (int e = preprocess_check("sd_preprocess_images_jitter", images, B, Hin, Win, Hout, Wout, h_bounds, h_kk, h_ksize, v_bounds, v_kk, v_ksize, mean3,
                                 std3, out, workspace)) return e;
    SD_REQUIRE(jitter_order && jitter_factors, SD_ERR_INVALID, "sd_preprocess_images_jitter: null jitter parameters");
    SD_REQUIRE(B <= 65535, SD_ERR_INVALID, "sd_preprocess_images_jitter: batch %d > 65535", B);
    SD_REQUIRE(workspace_bytes >= sd_preprocess_jitter_workspace_bytes(B, Hin, Win, Hout, Wout), SD_ERR_WORKSPACE,
               "sd_preprocess_images_jitter: workspace %zu < %zu", workspace_bytes, sd_preprocess_jitter_workspace_bytes(B, Hin, Win, Hout, Wout));
    hipStream_t st = (hipStream_t)stream;
    uint8_t* tmp = reinterpret_cast<uint8_t*>(workspace);
    uint8_t* img = tmp + sd_preprocess_workspace_bytes(B, Hin, Win, Wout);
    unsigned long long* lsum = reinterpret_cast<unsigned long long*>(img + align_up((size_t)B * Hout * Wout * 3, 256));
    SD_HIP(hipMemsetAsync(lsum, 0, (size_t)B * 8, st));
    const int64_t rows = (int64_t)B * Hin, npix = (int64_t)Hout * Wout;
    hipLaunchKernelGGL(k_resample_h, dim3(cdiv(rows * Wout, 256)), dim3(256), 0, st, images, tmp, Hin, Win, Wout, h_bounds, h_kk, h_ksize, rows);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_resample_v_u8, dim3(cdiv((int64_t)B * npix, 256)), dim3(256), 0, st, tmp, img, Hin, Hout, Wout, v_bounds, v_kk, v_ksize, B);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_jitter_lsum, dim3(std::min<int64_t>(cdiv(npix, 256), 64), B), dim3(256), 0, st, img, npix, jitter_order, jitter_factors, lsum);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_jitter_norm, dim3(cdiv((int64_t)B * npix, 256)), dim3(256), 0, st, img, out, Hout, Wout, jitter_order, jitter_factors, lsum,
                       flips, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], B);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
