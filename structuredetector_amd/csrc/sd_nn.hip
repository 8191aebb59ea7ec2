// Non-GEMM layers of the SDNet backbone / FPN on gfx950 (all HBM-bound, NHWC fp32):
// BatchNorm2d (training statistics, apply + residual + ReLU, backward), MaxPool2d(3,2,1),
// nearest-x2 upsample backward, the 1x1 head (NHWC -> NCHW) forward/backward, column sums for
// bias gradients and the fused Adam step.  They replace the torch.nn modules used by
// src/sdnet/model/network.py:6-57 and torchvision's resnet34 (BasicBlock), and
// torch.optim.Adam in src/sdnet/model/trainer.py:53,124.
#include "sd_common.h"
#include "sd_mfma.h"
#include <type_traits>

namespace sd {

// ------------------------------------------------------------------------------------------
// column reductions over a [M][C] matrix (C % 4 == 0, C <= 1024): per-block partial sums of up
// to two quantities, then a finalize kernel in double.  Thread layout: C/4 float4 columns x
// (256 / (C/4)) row lanes.
// ------------------------------------------------------------------------------------------
constexpr int RED_ROWS_PER_BLOCK = 256;

// MODE 0: sum x, sum x^2                      (BN statistics)
// MODE 1: sum g, sum g*xhat  with g = dy * [y > 0 if relu]   (BN backward)
// MODE 2: sum x                               (bias gradient)
template <int MODE, typename T = float>
__global__ __launch_bounds__(256) void k_col_reduce(const T* __restrict__ a, const T* __restrict__ b, const T* __restrict__ y,
                                                     const float* __restrict__ mean, const float* __restrict__ invstd,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                     int64_t M, int C, float* __restrict__ partial, int rpb) {
    __shared__ float4 red[2][256];
    const int cols = C >> 2;                       // float4 columns
    const int lanes = 256 / cols;                  // row lanes (cols <= 256)
    const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
    const int64_t r0 = (int64_t)blockIdx.x * rpb;          // rpb rows per block (red_rows(M): >= 1024 blocks where M allows)
    const int64_t r1 = min(r0 + rpb, M);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    float4 mu = s0, is = s0, ga = s0, be = s0;
    if (MODE == 1 && rl < lanes) {
        mu = reinterpret_cast<const float4*>(mean)[col];
        is = reinterpret_cast<const float4*>(invstd)[col];
        if (relu == 2) { ga = reinterpret_cast<const float4*>(gamma)[col]; be = reinterpret_cast<const float4*>(beta)[col]; }
    }
    if (rl < lanes) {
#pragma unroll 4
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const float4 v = ld4(a, r * cols + col);
            if (MODE == 0) {
                s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
                s1.x += v.x * v.x; s1.y += v.y * v.y; s1.z += v.z * v.z; s1.w += v.w * v.w;
            } else if (MODE == 2) {
                s0.x += v.x; s0.y += v.y; s0.z += v.z; s0.w += v.w;
            } else {
                float4 g = v;
                const float4 xv = ld4(b, r * cols + col);
                if (relu == 1) {            // mask from the saved post-activation (layers with a residual input)
                    const float4 yy = ld4(y, r * cols + col);
                    g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
                    g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
                } else if (relu == 2) {     // mask recomputed from x (same expression as k_bn_apply): one tensor read less
                    g.x = ((xv.x - mu.x) * is.x * ga.x + be.x) > 0.f ? g.x : 0.f; g.y = ((xv.y - mu.y) * is.y * ga.y + be.y) > 0.f ? g.y : 0.f;
                    g.z = ((xv.z - mu.z) * is.z * ga.z + be.z) > 0.f ? g.z : 0.f; g.w = ((xv.w - mu.w) * is.w * ga.w + be.w) > 0.f ? g.w : 0.f;
                } else if (relu == 3) {     // mask bytes written by k_bn_apply (residual layers): `y` points at them
                    const uint8_t mb = reinterpret_cast<const uint8_t*>(y)[r * cols + col];
                    g.x = (mb & 1) ? g.x : 0.f; g.y = (mb & 2) ? g.y : 0.f; g.z = (mb & 4) ? g.z : 0.f; g.w = (mb & 8) ? g.w : 0.f;
                }
                s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
                s1.x += g.x * ((xv.x - mu.x) * is.x); s1.y += g.y * ((xv.y - mu.y) * is.y);
                s1.z += g.z * ((xv.z - mu.z) * is.z); s1.w += g.w * ((xv.w - mu.w) * is.w);
            }
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < lanes; ++l) {
            const float4 u = red[0][l * cols + col], w = red[1][l * cols + col];
            s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
            s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
        }
        float* dst = partial + (int64_t)blockIdx.x * 2 * C;
        reinterpret_cast<float4*>(dst)[col] = s0;
        reinterpret_cast<float4*>(dst + C)[col] = s1;
    }
}

// out[j][w] = sum over the rows of slab j of in[r][w]  (W <= 1024 floats per row, W % 4 == 0): coalesced row reads
__global__ __launch_bounds__(256) void k_rows_fold(const float* __restrict__ in, int rows, int W, int slab, float* __restrict__ out) {
    __shared__ float4 red[256];
    const int cols = W >> 2, lanes = 256 / cols;
    const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
    const int r0 = blockIdx.x * slab, r1 = min(r0 + slab, rows);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rl < lanes) {
#pragma unroll 4
        for (int r = r0 + rl; r < r1; r += lanes) {
            const float4 v = reinterpret_cast<const float4*>(in + (int64_t)r * W)[col];
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < lanes; ++l) { const float4 u = red[l * cols + col]; s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w; }
        reinterpret_cast<float4*>(out + (int64_t)blockIdx.x * W)[col] = s;
    }
}

// FIN 0: BN statistics -> mean, invstd, running stats (momentum, unbiased running var)
// FIN 1: BN backward   -> dgamma (+=), dbeta (+=), and the two means needed by the apply pass
// FIN 2: bias gradient -> out0 (+=)
template <int FIN>
__global__ __launch_bounds__(256) void k_col_finalize(const float* __restrict__ partial, int nblocks, int C, double M, float eps,
                                                       float momentum, float* __restrict__ out0, float* __restrict__ out1,
                                                       float* __restrict__ run_mean, float* __restrict__ run_var,
                                                       float* __restrict__ aux0, float* __restrict__ aux1, int accumulate) {
    // 4 channels x 64 lanes over the partial rows per workgroup, 8 loads in flight per lane (the pass is latency-bound: up to
    // 8192 partial rows of a few KB each); fixed order -> deterministic
    __shared__ double r0[4][4], r1[4][4];
    const int lc = threadIdx.x & 3, lr = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + lc;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        double t0[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // eight rows in flight per lane and trip, the last trip's missing rows read row 0 and add nothing (no serial tail: with 128-512 partial
        // rows -- most layers -- the pass used to be two to seven dependent round trips)
        for (int b = lr; b < nblocks; b += 64 * 8) {
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int row = b + 64 * u;
                const int64_t o = (int64_t)(row < nblocks ? row : 0) * 2 * C + c;
                v0[u] = partial[o];
                v1[u] = partial[o + C];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = b + 64 * u < nblocks;
                t0[u] += ok ? (double)v0[u] : 0.0; t1[u] += ok ? (double)v1[u] : 0.0;
            }
        }
        s0 = ((t0[0] + t0[1]) + (t0[2] + t0[3])) + ((t0[4] + t0[5]) + (t0[6] + t0[7]));
        s1 = ((t1[0] + t1[1]) + (t1[2] + t1[3])) + ((t1[4] + t1[5]) + (t1[6] + t1[7]));
    }
    // the 16 lanes of a wave that share a channel (thread index stride 4), then the 4 waves through LDS
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
    if ((threadIdx.x & 63) < 4) { r0[threadIdx.x >> 6][lc] = s0; r1[threadIdx.x >> 6][lc] = s1; }
    __syncthreads();
    if (lr != 0 || c >= C) return;
    s0 = (r0[0][lc] + r0[1][lc]) + (r0[2][lc] + r0[3][lc]);
    s1 = (r1[0][lc] + r1[1][lc]) + (r1[2][lc] + r1[3][lc]);
    if (FIN == 0) {
        const double mean = s0 / M;
        const double var = fmax(s1 / M - mean * mean, 0.0);             // biased variance (normalisation)
        out0[c] = (float)mean;
        out1[c] = (float)(1.0 / sqrt(var + (double)eps));
        if (run_mean) {
            const double unb = M > 1.0 ? var * M / (M - 1.0) : var;     // torch: running_var uses the unbiased estimate
            run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * mean);
            run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
        }
    } else if (FIN == 1) {
        out0[c] = (accumulate ? out0[c] : 0.f) + (float)s1;             // dgamma = sum g * xhat
        out1[c] = (accumulate ? out1[c] : 0.f) + (float)s0;             // dbeta  = sum g
        aux0[c] = (float)(s0 / M);
        aux1[c] = (float)(s1 / M);
    } else {
        out0[c] = (accumulate ? out0[c] : 0.f) + (float)s0;
    }
}

// y = [relu]( (x - mean) * invstd * gamma + beta [+ res] )
template <typename T>
__global__ __launch_bounds__(256) void k_bn_apply(const T* __restrict__ x, T* __restrict__ y, int64_t n4, int C,
                                                   const float* __restrict__ mean, const float* __restrict__ invstd,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   const T* __restrict__ res, int relu, uint8_t* __restrict__ mask) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int cols = C >> 2;
    // The grid stride is a multiple of the column count for every power-of-two channel count (256 % cols == 0): a thread then stays on
    // ONE column and its four parameter vectors are loaded once -- inside the loop they were four 16-byte L1 requests next to one or two
    // requests of data per element (`fixed`); other channel counts take the per-element lookup.
    const bool fixed = stride % cols == 0;
    const int col0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cols);
    float4 mu = reinterpret_cast<const float4*>(mean)[col0], is = reinterpret_cast<const float4*>(invstd)[col0];
    float4 g = reinterpret_cast<const float4*>(gamma)[col0], b = reinterpret_cast<const float4*>(beta)[col0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = ld4(x, i);
        if (!fixed) {
            const int col = (int)(i % cols);
            mu = reinterpret_cast<const float4*>(mean)[col]; is = reinterpret_cast<const float4*>(invstd)[col];
            g = reinterpret_cast<const float4*>(gamma)[col]; b = reinterpret_cast<const float4*>(beta)[col];
        }
        float4 o;
        o.x = (v.x - mu.x) * is.x * g.x + b.x; o.y = (v.y - mu.y) * is.y * g.y + b.y;
        o.z = (v.z - mu.z) * is.z * g.z + b.z; o.w = (v.w - mu.w) * is.w * g.w + b.w;
        if (res) {
            const float4 r = ld4(res, i);
            o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        // one mask byte per float4 (bit j = element j is positive): the backward of a residual layer reads 1/16 of the bytes of y
        if (mask) mask[i] = (uint8_t)((o.x > 0.f ? 1 : 0) | (o.y > 0.f ? 2 : 0) | (o.z > 0.f ? 4 : 0) | (o.w > 0.f ? 8 : 0));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st4(y, i, o);
    }
}

// eval-mode BN folded into a per-channel affine (consumed by the conv epilogue)
__global__ __launch_bounds__(256) void k_bn_fold(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                                  int C, float* scale, float* shift) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
}

// dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * [y > 0];  optionally g_out = g
template <typename T>
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                       int relu, int64_t n4, int C, const float* __restrict__ mean,
                                                       const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, const float* __restrict__ mg,
                                                       const float* __restrict__ mgx, T* __restrict__ dx, T* __restrict__ g_out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int cols = C >> 2;
    const bool fixed = stride % cols == 0;                      // (see k_bn_apply: one column per thread, parameters loaded once)
    const int col0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cols);
    float4 mu = reinterpret_cast<const float4*>(mean)[col0], is = reinterpret_cast<const float4*>(invstd)[col0];
    float4 ga = reinterpret_cast<const float4*>(gamma)[col0];
    float4 be = relu == 2 ? reinterpret_cast<const float4*>(beta)[col0] : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 a = reinterpret_cast<const float4*>(mg)[col0], b = reinterpret_cast<const float4*>(mgx)[col0];
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 g = ld4(dy, i);
        const float4 xv = ld4(x, i);
        if (!fixed) {
            const int col = (int)(i % cols);
            mu = reinterpret_cast<const float4*>(mean)[col]; is = reinterpret_cast<const float4*>(invstd)[col];
            ga = reinterpret_cast<const float4*>(gamma)[col];
            if (relu == 2) be = reinterpret_cast<const float4*>(beta)[col];
            a = reinterpret_cast<const float4*>(mg)[col]; b = reinterpret_cast<const float4*>(mgx)[col];
        }
        if (relu == 1) {
            const float4 yy = ld4(y, i);
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f; g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        } else if (relu == 3) {
            const uint8_t mb = reinterpret_cast<const uint8_t*>(y)[i];
            g.x = (mb & 1) ? g.x : 0.f; g.y = (mb & 2) ? g.y : 0.f; g.z = (mb & 4) ? g.z : 0.f; g.w = (mb & 8) ? g.w : 0.f;
        } else if (relu == 2) {
            g.x = ((xv.x - mu.x) * is.x * ga.x + be.x) > 0.f ? g.x : 0.f; g.y = ((xv.y - mu.y) * is.y * ga.y + be.y) > 0.f ? g.y : 0.f;
            g.z = ((xv.z - mu.z) * is.z * ga.z + be.z) > 0.f ? g.z : 0.f; g.w = ((xv.w - mu.w) * is.w * ga.w + be.w) > 0.f ? g.w : 0.f;
        }
        float4 o;
        o.x = ga.x * is.x * (g.x - a.x - (xv.x - mu.x) * is.x * b.x);
        o.y = ga.y * is.y * (g.y - a.y - (xv.y - mu.y) * is.y * b.y);
        o.z = ga.z * is.z * (g.z - a.z - (xv.z - mu.z) * is.z * b.z);
        o.w = ga.w * is.w * (g.w - a.w - (xv.w - mu.w) * is.w * b.w);
        st4(dx, i, o);
        if (g_out) st4(g_out, i, g);
    }
}

// MaxPool2d(3, stride 2, pad 1), NHWC.  idx = winning tap (first maximum in row-major window
// order, as ATen) for the backward.
__global__ __launch_bounds__(256) void k_maxpool_fwd(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx, int B,
                                                      int Hi, int Wi, int Ho, int Wo, int C) {
    const int cols = C >> 2;
    const int64_t n4 = (int64_t)B * Ho * Wo * cols;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int col = (int)(i % cols);
    int64_t t = i / cols;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 w = make_uchar4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iy = oy * 2 - 1 + r, ix = ox * 2 - 1 + s;
            if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
            const float4 v = reinterpret_cast<const float4*>(x + (((int64_t)b * Hi + iy) * Wi + ix) * C)[col];
            const uint8_t tap = (uint8_t)(r * 3 + s);
            if (v.x > m.x) { m.x = v.x; w.x = tap; }
            if (v.y > m.y) { m.y = v.y; w.y = tap; }
            if (v.z > m.z) { m.z = v.z; w.z = tap; }
            if (v.w > m.w) { m.w = v.w; w.w = tap; }
        }
    }
    reinterpret_cast<float4*>(y)[i] = m;
    reinterpret_cast<uchar4*>(idx)[i] = w;
}

// gather form of the max-pool backward (no atomics): every input pixel looks at the <= 4 windows
// that contain it.
__global__ __launch_bounds__(256) void k_maxpool_bwd(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                      int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int cols = C >> 2;
    const int64_t n4 = (int64_t)B * Hi * Wi * cols;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int col = (int)(i % cols);
    int64_t t = i / cols;
    const int ix = (int)(t % Wi); t /= Wi;
    const int iy = (int)(t % Hi);
    const int b = (int)(t / Hi);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = (iy + 1) / 2 - ((iy + 1) % 2 == 0 ? 1 : 0); oy <= (iy + 1) / 2; ++oy) {
        if (oy < 0 || oy >= Ho) continue;
        const int r = iy - (oy * 2 - 1);
        if (r < 0 || r > 2) continue;
        for (int ox = (ix + 1) / 2 - ((ix + 1) % 2 == 0 ? 1 : 0); ox <= (ix + 1) / 2; ++ox) {
            if (ox < 0 || ox >= Wo) continue;
            const int s = ix - (ox * 2 - 1);
            if (s < 0 || s > 2) continue;
            const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * cols + col;
            const uchar4 w = reinterpret_cast<const uchar4*>(idx)[o];
            const float4 g = reinterpret_cast<const float4*>(dy)[o];
            const uint8_t tap = (uint8_t)(r * 3 + s);
            if (w.x == tap) acc.x += g.x;
            if (w.y == tap) acc.y += g.y;
            if (w.z == tap) acc.z += g.z;
            if (w.w == tap) acc.w += g.w;
        }
    }
    reinterpret_cast<float4*>(dx)[i] = acc;
}

// ---- stem tail fused: BatchNorm(train) + ReLU + MaxPool2d(3, 2, 1) without materialising the full-resolution activation.
// forward: the pooled map and the winning taps straight from the conv output x (same affine expression as k_bn_apply, same
// first-maximum rule as k_maxpool_fwd).
// XT / PT: element type of the conv output x and of the pooled map (float, or uint16_t = bf16 in the mixed-precision step: fp32 arithmetic,
// one rounding at the store; the winning tap is decided on the fp32 values)
template <typename XT, typename PT>
__global__ __launch_bounds__(256) void k_bn_relu_maxpool_fwd(const XT* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, PT* __restrict__ y,
                                                              uint8_t* __restrict__ idx, int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int cols = C >> 2;
    const int64_t n4 = (int64_t)B * Ho * Wo * cols;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int col = (int)(i % cols);
    int64_t t = i / cols;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    const float4 mu = reinterpret_cast<const float4*>(mean)[col], is = reinterpret_cast<const float4*>(invstd)[col];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[col], be = reinterpret_cast<const float4*>(beta)[col];
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    uchar4 w = make_uchar4(0, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iy = oy * 2 - 1 + r, ix = ox * 2 - 1 + s;
            if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
            const float4 xv = ld4(x, (((int64_t)b * Hi + iy) * Wi + ix) * cols + col);
            float4 v;
            v.x = fmaxf((xv.x - mu.x) * is.x * ga.x + be.x, 0.f); v.y = fmaxf((xv.y - mu.y) * is.y * ga.y + be.y, 0.f);
            v.z = fmaxf((xv.z - mu.z) * is.z * ga.z + be.z, 0.f); v.w = fmaxf((xv.w - mu.w) * is.w * ga.w + be.w, 0.f);
            const uint8_t tap = (uint8_t)(r * 3 + s);
            if (v.x > m.x) { m.x = v.x; w.x = tap; }
            if (v.y > m.y) { m.y = v.y; w.y = tap; }
            if (v.z > m.z) { m.z = v.z; w.z = tap; }
            if (v.w > m.w) { m.w = v.w; w.w = tap; }
        }
    }
    st4(y, i, m);
    reinterpret_cast<uchar4*>(idx)[i] = w;
}

// The same pass, two horizontally adjacent windows per thread (Wo even) and 16 bytes per load for both element types (fp32: 4 channels,
// bf16: 8): the windows (oy, 2 j) and (oy, 2 j + 1) share the input column 4 j + 1, so 15 loads serve two outputs (7.5 per output; the
// one-window form needs 9, of 8 bytes for bf16: 258 us at bs = 64 for a 140 us floor).  Same affine expression, same scan order
// (row-major, strict >): the same winning taps.  Plain loads: every input element is read by two or four threads.
template <int VEC> struct PoolVec { float v[VEC]; };
__device__ __forceinline__ PoolVec<4> pool_ld(const float* p, int64_t e) {
    const float4 r = *reinterpret_cast<const float4*>(p + e);
    return PoolVec<4>{{r.x, r.y, r.z, r.w}};
}
__device__ __forceinline__ PoolVec<8> pool_ld(const uint16_t* p, int64_t e) {
    const uint4 r = *reinterpret_cast<const uint4*>(p + e);
    return PoolVec<8>{{bf16_to_f32((uint16_t)(r.x & 0xffff)), bf16_to_f32((uint16_t)(r.x >> 16)), bf16_to_f32((uint16_t)(r.y & 0xffff)), bf16_to_f32((uint16_t)(r.y >> 16)),
                       bf16_to_f32((uint16_t)(r.z & 0xffff)), bf16_to_f32((uint16_t)(r.z >> 16)), bf16_to_f32((uint16_t)(r.w & 0xffff)), bf16_to_f32((uint16_t)(r.w >> 16))}};
}
__device__ __forceinline__ void pool_st(float* p, int64_t e, const PoolVec<4>& m) { *reinterpret_cast<float4*>(p + e) = make_float4(m.v[0], m.v[1], m.v[2], m.v[3]); }
__device__ __forceinline__ void pool_st(uint16_t* p, int64_t e, const PoolVec<8>& m) {
    uint4 r;
    r.x = (uint32_t)f32_to_bf16(m.v[0]) | ((uint32_t)f32_to_bf16(m.v[1]) << 16); r.y = (uint32_t)f32_to_bf16(m.v[2]) | ((uint32_t)f32_to_bf16(m.v[3]) << 16);
    r.z = (uint32_t)f32_to_bf16(m.v[4]) | ((uint32_t)f32_to_bf16(m.v[5]) << 16); r.w = (uint32_t)f32_to_bf16(m.v[6]) | ((uint32_t)f32_to_bf16(m.v[7]) << 16);
    *reinterpret_cast<uint4*>(p + e) = r;
}
template <typename XT, typename PT, int VEC>
__global__ __launch_bounds__(256) void k_bn_relu_maxpool_fwd_pair(const XT* __restrict__ x, const float* __restrict__ mean,
                                                                   const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, PT* __restrict__ y,
                                                                   uint8_t* __restrict__ idx, int B, int Hi, int Wi, int Ho, int Wo, int C) {
    const int cols = C / VEC, Wp = Wo >> 1;
    const int64_t n = (int64_t)B * Ho * Wp * cols;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int col = (int)(i % cols);
    int64_t t = i / cols;
    const int oxp = (int)(t % Wp); t /= Wp;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float mu[VEC], is[VEC], ga[VEC], be[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { mu[k] = mean[col * VEC + k]; is[k] = invstd[col * VEC + k]; ga[k] = gamma[col * VEC + k]; be[k] = beta[col * VEC + k]; }
    PoolVec<VEC> m0, m1;
    uint8_t w0[VEC], w1[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { m0.v[k] = m1.v[k] = -INFINITY; w0[k] = w1[k] = 0; }
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = oy * 2 - 1 + r;
        if (iy < 0 || iy >= Hi) continue;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int ix = oxp * 4 - 1 + c;
            if (ix < 0 || ix >= Wi) continue;
            const PoolVec<VEC> xv = pool_ld(x, (((int64_t)b * Hi + iy) * Wi + ix) * C + col * VEC);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float v = fmaxf((xv.v[k] - mu[k]) * is[k] * ga[k] + be[k], 0.f);
                if (c <= 2 && v > m0.v[k]) { m0.v[k] = v; w0[k] = (uint8_t)(r * 3 + c); }
                if (c >= 2 && v > m1.v[k]) { m1.v[k] = v; w1[k] = (uint8_t)(r * 3 + c - 2); }
            }
        }
    }
    const int64_t o = (((int64_t)b * Ho + oy) * Wo + 2 * oxp) * C + col * VEC;
    pool_st(y, o, m0); pool_st(y, o + C, m1);
    uint32_t p0[VEC / 4], p1[VEC / 4];
#pragma unroll
    for (int k = 0; k < VEC / 4; ++k) {
        p0[k] = w0[4 * k] | ((uint32_t)w0[4 * k + 1] << 8) | ((uint32_t)w0[4 * k + 2] << 16) | ((uint32_t)w0[4 * k + 3] << 24);
        p1[k] = w1[4 * k] | ((uint32_t)w1[4 * k + 1] << 8) | ((uint32_t)w1[4 * k + 2] << 16) | ((uint32_t)w1[4 * k + 3] << 24);
    }
#pragma unroll
    for (int k = 0; k < VEC / 4; ++k) { reinterpret_cast<uint32_t*>(idx + o)[k] = p0[k]; reinterpret_cast<uint32_t*>(idx + o + C)[k] = p1[k]; }
}

// gradient w.r.t. the (never stored) BatchNorm+ReLU output at input pixel (b, iy, ix): gather form of the max-pool backward
// (k_maxpool_bwd) followed by the ReLU mask recomputed from x
__device__ __forceinline__ float4 pool_relu_grad(const float* __restrict__ dpool, const uint8_t* __restrict__ idx, int b, int iy, int ix,
                                                 int col, int cols, int Ho, int Wo, const float4 xv, const float4 mu, const float4 is,
                                                 const float4 ga, const float4 be) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int oy = (iy + 1) / 2 - ((iy + 1) % 2 == 0 ? 1 : 0); oy <= (iy + 1) / 2; ++oy) {
        if (oy < 0 || oy >= Ho) continue;
        const int r = iy - (oy * 2 - 1);
        if (r < 0 || r > 2) continue;
        for (int ox = (ix + 1) / 2 - ((ix + 1) % 2 == 0 ? 1 : 0); ox <= (ix + 1) / 2; ++ox) {
            if (ox < 0 || ox >= Wo) continue;
            const int s = ix - (ox * 2 - 1);
            if (s < 0 || s > 2) continue;
            const int64_t o = (((int64_t)b * Ho + oy) * Wo + ox) * cols + col;
            const uchar4 w = reinterpret_cast<const uchar4*>(idx)[o];
            const float4 g = reinterpret_cast<const float4*>(dpool)[o];
            const uint8_t tap = (uint8_t)(r * 3 + s);
            if (w.x == tap) acc.x += g.x;
            if (w.y == tap) acc.y += g.y;
            if (w.z == tap) acc.z += g.z;
            if (w.w == tap) acc.w += g.w;
        }
    }
    acc.x = ((xv.x - mu.x) * is.x * ga.x + be.x) > 0.f ? acc.x : 0.f; acc.y = ((xv.y - mu.y) * is.y * ga.y + be.y) > 0.f ? acc.y : 0.f;
    acc.z = ((xv.z - mu.z) * is.z * ga.z + be.z) > 0.f ? acc.z : 0.f; acc.w = ((xv.w - mu.w) * is.w * ga.w + be.w) > 0.f ? acc.w : 0.f;
    return acc;
}

// The same gradient for the four pixels (2k + dy, 2j + dx) of an even-aligned 2x2 quad at once (Hi, Wi even): they only see the
// windows (k, j), (k, j+1), (k+1, j), (k+1, j+1), so four window loads serve four pixels (the per-pixel gather needs nine) and
// the window set is fixed -- no data-dependent loops.  g[2 * dy + dx].
template <typename PT>
__device__ __forceinline__ void pool_relu_grad_quad(const PT* __restrict__ dpool, const uint8_t* __restrict__ idx, int b, int k, int j,
                                                    int col, int cols, int Ho, int Wo, const float4* xv, const float4 mu, const float4 is,
                                                    const float4 ga, const float4 be, float4* g) {
    const bool hy = k + 1 < Ho, hx = j + 1 < Wo;
    const int64_t o00 = (((int64_t)b * Ho + k) * Wo + j) * cols + col;
    const int64_t o01 = hx ? o00 + cols : o00, o10 = hy ? o00 + (int64_t)Wo * cols : o00, o11 = (hx && hy) ? o00 + (int64_t)(Wo + 1) * cols : o00;
    const uchar4 w00 = reinterpret_cast<const uchar4*>(idx)[o00], w01 = reinterpret_cast<const uchar4*>(idx)[o01];
    const uchar4 w10 = reinterpret_cast<const uchar4*>(idx)[o10], w11 = reinterpret_cast<const uchar4*>(idx)[o11];
    const float4 d00 = ld4(dpool, o00);
    float4 d01 = ld4(dpool, o01), d10 = ld4(dpool, o10);
    float4 d11 = ld4(dpool, o11);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!hx) d01 = z;
    if (!hy) d10 = z;
    if (!(hx && hy)) d11 = z;
#define SD_PICK(W, D, TAP) make_float4((W).x == (TAP) ? (D).x : 0.f, (W).y == (TAP) ? (D).y : 0.f, (W).z == (TAP) ? (D).z : 0.f, (W).w == (TAP) ? (D).w : 0.f)
#define SD_ADD4(A, B) { (A).x += (B).x; (A).y += (B).y; (A).z += (B).z; (A).w += (B).w; }
    // pixel (2k, 2j): window (k, j) tap (1, 1)
    g[0] = SD_PICK(w00, d00, 4);
    // pixel (2k, 2j+1): (k, j) tap (1, 2); (k, j+1) tap (1, 0)
    g[1] = SD_PICK(w00, d00, 5); { const float4 t = SD_PICK(w01, d01, 3); SD_ADD4(g[1], t) }
    // pixel (2k+1, 2j): (k, j) tap (2, 1); (k+1, j) tap (0, 1)
    g[2] = SD_PICK(w00, d00, 7); { const float4 t = SD_PICK(w10, d10, 1); SD_ADD4(g[2], t) }
    // pixel (2k+1, 2j+1): (k, j) tap (2, 2); (k, j+1) tap (2, 0); (k+1, j) tap (0, 2); (k+1, j+1) tap (0, 0)
    g[3] = SD_PICK(w00, d00, 8);
    { const float4 t = SD_PICK(w01, d01, 6); SD_ADD4(g[3], t) }
    { const float4 t = SD_PICK(w10, d10, 2); SD_ADD4(g[3], t) }
    { const float4 t = SD_PICK(w11, d11, 0); SD_ADD4(g[3], t) }
#undef SD_PICK
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        g[q].x = ((xv[q].x - mu.x) * is.x * ga.x + be.x) > 0.f ? g[q].x : 0.f; g[q].y = ((xv[q].y - mu.y) * is.y * ga.y + be.y) > 0.f ? g[q].y : 0.f;
        g[q].z = ((xv[q].z - mu.z) * is.z * ga.z + be.z) > 0.f ? g[q].z : 0.f; g[q].w = ((xv[q].w - mu.w) * is.w * ga.w + be.w) > 0.f ? g[q].w : 0.f;
    }
}

// quad form of the reduction pass: a block takes RED_ROWS_PER_BLOCK / 4 quads (same partial-row count as the pixel form)
template <typename XT, typename PT>
__global__ __launch_bounds__(256) void k_pool_bn_bwd_reduce_quad(const PT* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                  const XT* __restrict__ x, const float* __restrict__ mean,
                                                                  const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, int Hi, int Wi, int Ho, int Wo, int64_t MQ,
                                                                  int C, float* __restrict__ partial) {
    __shared__ float4 red[2][256];
    const int cols = C >> 2, lanes = 256 / cols;
    const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
    constexpr int QPB = RED_ROWS_PER_BLOCK / 4;
    const int64_t q0 = (int64_t)blockIdx.x * QPB, q1 = min(q0 + QPB, MQ);
    const int Hq = Hi >> 1, Wq = Wi >> 1;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (rl < lanes) {
        const float4 mu = reinterpret_cast<const float4*>(mean)[col], is = reinterpret_cast<const float4*>(invstd)[col];
        const float4 ga = reinterpret_cast<const float4*>(gamma)[col], be = reinterpret_cast<const float4*>(beta)[col];
        for (int64_t q = q0 + rl; q < q1; q += lanes) {
            const int j = (int)(q % Wq);
            const int64_t t = q / Wq;
            const int k = (int)(t % Hq), b = (int)(t / Hq);
            const int64_t p00 = (((int64_t)b * Hi + 2 * k) * Wi + 2 * j) * cols + col;
            float4 xv[4], g[4];
            xv[0] = ld4(x, p00); xv[1] = ld4(x, p00 + cols);
            xv[2] = ld4(x, p00 + (int64_t)Wi * cols); xv[3] = ld4(x, p00 + (int64_t)(Wi + 1) * cols);
            pool_relu_grad_quad(dpool, idx, b, k, j, col, cols, Ho, Wo, xv, mu, is, ga, be, g);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s0.x += g[u].x; s0.y += g[u].y; s0.z += g[u].z; s0.w += g[u].w;
                s1.x += g[u].x * ((xv[u].x - mu.x) * is.x); s1.y += g[u].y * ((xv[u].y - mu.y) * is.y);
                s1.z += g[u].z * ((xv[u].z - mu.z) * is.z); s1.w += g[u].w * ((xv[u].w - mu.w) * is.w);
            }
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < lanes; ++l) {
            const float4 u = red[0][l * cols + col], w = red[1][l * cols + col];
            s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
            s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
        }
        float* dst = partial + (int64_t)blockIdx.x * 2 * C;
        reinterpret_cast<float4*>(dst)[col] = s0;
        reinterpret_cast<float4*>(dst + C)[col] = s1;
    }
}

template <typename XT, typename PT, typename DT = float>
__global__ __launch_bounds__(256) void k_pool_bn_bwd_apply_quad(const PT* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                                 const XT* __restrict__ x, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ mg,
                                                                 const float* __restrict__ mgx, int Hi, int Wi, int Ho, int Wo, int64_t nq4,
                                                                 int C, DT* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq4) return;
    const int cols = C >> 2;
    const int col = (int)(i % cols);
    const int64_t q = i / cols;
    const int Hq = Hi >> 1, Wq = Wi >> 1;
    const int j = (int)(q % Wq);
    const int64_t t = q / Wq;
    const int k = (int)(t % Hq), b = (int)(t / Hq);
    const float4 mu = reinterpret_cast<const float4*>(mean)[col], is = reinterpret_cast<const float4*>(invstd)[col];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[col], be = reinterpret_cast<const float4*>(beta)[col];
    const float4 a = reinterpret_cast<const float4*>(mg)[col], bb = reinterpret_cast<const float4*>(mgx)[col];
    const int64_t p00 = (((int64_t)b * Hi + 2 * k) * Wi + 2 * j) * cols + col;
    const int64_t off[4] = {p00, p00 + cols, p00 + (int64_t)Wi * cols, p00 + (int64_t)(Wi + 1) * cols};
    float4 xv[4], g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xv[u] = ld4(x, off[u]);
    pool_relu_grad_quad(dpool, idx, b, k, j, col, cols, Ho, Wo, xv, mu, is, ga, be, g);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        float4 o;
        o.x = ga.x * is.x * (g[u].x - a.x - (xv[u].x - mu.x) * is.x * bb.x);
        o.y = ga.y * is.y * (g[u].y - a.y - (xv[u].y - mu.y) * is.y * bb.y);
        o.z = ga.z * is.z * (g[u].z - a.z - (xv[u].z - mu.z) * is.z * bb.z);
        o.w = ga.w * is.w * (g[u].w - a.w - (xv[u].w - mu.w) * is.w * bb.w);
        st4(dx, off[u], o);
    }
}
#undef SD_ADD4

// reduction pass of the fused backward: per-channel partial sums of g and g * xhat over RED_ROWS_PER_BLOCK pixels (layout as k_col_reduce)
__global__ __launch_bounds__(256) void k_pool_bn_bwd_reduce(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                             const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, int Hi, int Wi, int Ho, int Wo, int64_t M,
                                                             int C, float* __restrict__ partial) {
    __shared__ float4 red[2][256];
    const int cols = C >> 2, lanes = 256 / cols;
    const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
    const int64_t r0 = (int64_t)blockIdx.x * RED_ROWS_PER_BLOCK, r1 = min(r0 + RED_ROWS_PER_BLOCK, M);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (rl < lanes) {
        const float4 mu = reinterpret_cast<const float4*>(mean)[col], is = reinterpret_cast<const float4*>(invstd)[col];
        const float4 ga = reinterpret_cast<const float4*>(gamma)[col], be = reinterpret_cast<const float4*>(beta)[col];
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const int ix = (int)(r % Wi);
            const int64_t t = r / Wi;
            const int iy = (int)(t % Hi), b = (int)(t / Hi);
            const float4 xv = reinterpret_cast<const float4*>(x + r * C)[col];
            const float4 g = pool_relu_grad(dpool, idx, b, iy, ix, col, cols, Ho, Wo, xv, mu, is, ga, be);
            s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
            s1.x += g.x * ((xv.x - mu.x) * is.x); s1.y += g.y * ((xv.y - mu.y) * is.y);
            s1.z += g.z * ((xv.z - mu.z) * is.z); s1.w += g.w * ((xv.w - mu.w) * is.w);
        }
    }
    red[0][threadIdx.x] = s0;
    red[1][threadIdx.x] = s1;
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < lanes; ++l) {
            const float4 u = red[0][l * cols + col], w = red[1][l * cols + col];
            s0.x += u.x; s0.y += u.y; s0.z += u.z; s0.w += u.w;
            s1.x += w.x; s1.y += w.y; s1.z += w.z; s1.w += w.w;
        }
        float* dst = partial + (int64_t)blockIdx.x * 2 * C;
        reinterpret_cast<float4*>(dst)[col] = s0;
        reinterpret_cast<float4*>(dst + C)[col] = s1;
    }
}

// apply pass of the fused backward: dx = gamma * invstd * (g - mean(g) - xhat * mean(g * xhat))
__global__ __launch_bounds__(256) void k_pool_bn_bwd_apply(const float* __restrict__ dpool, const uint8_t* __restrict__ idx,
                                                            const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ mg,
                                                            const float* __restrict__ mgx, int Hi, int Wi, int Ho, int Wo, int64_t n4, int C,
                                                            float* __restrict__ dx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int cols = C >> 2;
    const int col = (int)(i % cols);
    const int64_t r = i / cols;
    const int ix = (int)(r % Wi);
    const int64_t t = r / Wi;
    const int iy = (int)(t % Hi), b = (int)(t / Hi);
    const float4 mu = reinterpret_cast<const float4*>(mean)[col], is = reinterpret_cast<const float4*>(invstd)[col];
    const float4 ga = reinterpret_cast<const float4*>(gamma)[col], be = reinterpret_cast<const float4*>(beta)[col];
    const float4 xv = reinterpret_cast<const float4*>(x)[i];
    const float4 g = pool_relu_grad(dpool, idx, b, iy, ix, col, cols, Ho, Wo, xv, mu, is, ga, be);
    const float4 a = reinterpret_cast<const float4*>(mg)[col], bb = reinterpret_cast<const float4*>(mgx)[col];
    float4 o;
    o.x = ga.x * is.x * (g.x - a.x - (xv.x - mu.x) * is.x * bb.x);
    o.y = ga.y * is.y * (g.y - a.y - (xv.y - mu.y) * is.y * bb.y);
    o.z = ga.z * is.z * (g.z - a.z - (xv.z - mu.z) * is.z * bb.z);
    o.w = ga.w * is.w * (g.w - a.w - (xv.w - mu.w) * is.w * bb.w);
    reinterpret_cast<float4*>(dx)[i] = o;
}

// backward of nearest x2 upsample: dx[b,y,x,c] = sum of the 2x2 block of dy (+ add, nullable)
template <typename T>
__global__ __launch_bounds__(256) void k_up2_bwd(const T* __restrict__ dy, const T* __restrict__ add, T* __restrict__ dx, int B,
                                                  int H, int W, int C) {
    const int cols = C >> 2;
    const int64_t n4 = (int64_t)B * H * W * cols;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const int col = (int)(i % cols);
    int64_t t = i / cols;
    const int x = (int)(t % W); t /= W;
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const T* base = dy + (((int64_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C;
    const float4 v0 = ld4(base, col), v1 = ld4(base + C, col);
    const float4 v2 = ld4(base + (int64_t)2 * W * C, col);
    const float4 v3 = ld4(base + (int64_t)2 * W * C + C, col);
    float4 o = make_float4(v0.x + v1.x + v2.x + v3.x, v0.y + v1.y + v2.y + v3.y, v0.z + v1.z + v2.z + v3.z, v0.w + v1.w + v2.w + v3.w);
    if (add) {
        const float4 a = ld4(add, i);
        o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
    }
    st4(dx, i, o);
}

// ------------------------------------------------------------------------------------------
// Head: 1x1 conv C -> Co (<= 16) with bias, NHWC in, NCHW out (network.py:22-29,57).
// 64 pixels per block: the [64][C] tile is one contiguous 64*C*4-byte span, staged in LDS with a
// 4-float row pad (ds_read_b128 conflict-free), then thread (pixel, group) accumulates its
// outputs with wave-uniform (broadcast) weight reads.  HBM-bound: AI ~ 3 flop/B.
// ------------------------------------------------------------------------------------------
constexpr int HEAD_MAX_CO = 32;

__global__ __launch_bounds__(256) void k_head_fwd(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                   float* __restrict__ y, int64_t M, int HW, int C, int Co) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = C + 4;
    float* xs = lds;                 // [64][C+4]
    float* ws = lds + 64 * LD;       // [Co][C]
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int c4 = C >> 2;
    for (int i = threadIdx.x; i < 64 * c4; i += 256) {
        const int row = i / c4, col = i - row * c4;
        const int64_t m = m0 + row;
        const float4 v = m < M ? reinterpret_cast<const float4*>(x + m * C)[col] : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(xs + row * LD + col * 4) = v;
    }
    for (int i = threadIdx.x; i < Co * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int px = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float acc[HEAD_MAX_CO / 4];
#pragma unroll
    for (int j = 0; j < HEAD_MAX_CO / 4; ++j) acc[j] = 0.f;
    for (int c = 0; c < C; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(xs + px * LD + c);
#pragma unroll
        for (int j = 0; j < HEAD_MAX_CO / 4; ++j) {
            const int co = grp + 4 * j;
            if (co < Co) {
                const float4 ww = *reinterpret_cast<const float4*>(ws + co * C + c);
                acc[j] += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
            }
        }
    }
    const int64_t m = m0 + px;
    if (m < M) {
        const int64_t b = m / HW, pix = m - b * HW;
#pragma unroll
        for (int j = 0; j < HEAD_MAX_CO / 4; ++j) {
            const int co = grp + 4 * j;
            if (co < Co) y[(b * Co + co) * HW + pix] = acc[j] + bias[co];
        }
    }
}

// Head forward on an fp32 NHWC input of 128 channels (the default FPN depth), Co <= 16: 537 MB in, 29 MB out at bs = 64 -- a pure HBM
// stream.  k_head_fwd above stages 64 pixels per block and re-reads every staged row once per output-channel group plus the weights
// per thread (13x the input bytes in LDS reads): 230 us, twice the stream's time.  Here, as in k_head_fwd_bf16_c128, every WAVE walks
// its own tiles of 16 pixels with a private ring of NST stages (8 KB each) filled by LDS-DMA NST - 1 tiles ahead (no staging registers,
// no block barrier, counted vmcnt), and multiplies on v_mfma_f32_16x16x4_f32: A = the tile (lane (row, kq) reads chunk 4 j + kq of its
// pixel: one ds_read_b128 per four MFMA steps), B = the weights in registers (lane (n, kq): w[n][16 j + 4 kq + t]), D = 16 pixels x 16
// output channels, lane (n, q) holding pixels 4 q .. 4 q + 3 of channel n: one 16-byte store into the NCHW plane.
//   LDS image of a tile: row r x 32 slots of 16 bytes, chunk c of row r at physical slot c ^ (r & 15) (swizzle applied to the DMA's
//   SOURCE address; the 16 lanes of a ds_read_b128 pass -- rows 0 .. 15, same chunk -- hit 16 different slots).
constexpr int HF_NST = 4;
__global__ __launch_bounds__(256) void k_head_fwd_f32_c128(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                            float* __restrict__ y, int HW, int Co, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float hf_lds[];             // 4 waves x HF_NST x 8 KB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const T = hf_lds + wave * (HF_NST * 2048);
    const uint32_t t_base = lds_addr(T);
    const int row = lane & 15, kq = lane >> 4;
    f32x4 bw[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        bw[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (row < Co) bw[j] = *reinterpret_cast<const f32x4*>(w + row * 128 + 16 * j + 4 * kq);
    }
    const float bv = row < Co ? bias[row] : 0.f;
    const int nwaves = gridDim.x * 4;
    const int dr = lane >> 5, dsl = lane & 31;                                 // DMA: lane -> (row within a 2-row piece, physical slot)
    // tile t -> ring stage st: eight 1 KB pieces; tiles past the end re-read the last tile (the vmcnt accounting stays fixed; never consumed)
#define HF_ISSUE(t_, st_)                                                                                          \
    {                                                                                                              \
        const int64_t m0_ = (int64_t)min((t_), ntiles - 1) * 16;                                                  \
        _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                            \
            const int r = 2 * d + dr;                                                                              \
            lds_dma16(x + (m0_ + r) * 128 + ((dsl ^ (r & 15)) << 2), T + (st_) * 2048 + d * 256);                   \
        }                                                                                                          \
    }
    int tile = blockIdx.x * 4 + wave;
#pragma unroll
    for (int s = 0; s < HF_NST - 1; ++s) HF_ISSUE(tile + s * nwaves, s)
    int stage = 0;
    for (; tile < ntiles; tile += nwaves) {
        const int nst = stage == 0 ? HF_NST - 1 : stage - 1;                   // the stage consumed in the previous trip (its reads were waited for)
        HF_ISSUE(tile + (HF_NST - 1) * nwaves, nst)
        // at most the pieces of the HF_NST - 1 younger tiles outstanding (this wave's interleaved output stores only make the wait stricter)
        wait_vmcnt<8 * (HF_NST - 1)>();
        const uint32_t a0 = t_base + (uint32_t)stage * 8192u + (uint32_t)row * 512u;
        f32x4 a[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] = lds_read128_async<0>(a0 + (uint32_t)(((4 * j + kq) ^ row) << 4));
        SD_LDS_WAIT4(0, a[0], a[1], a[2], a[3]);
        SD_LDS_WAIT4(0, a[4], a[5], a[6], a[7]);
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; j += 2)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][t], bw[j][t], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 1][t], bw[j + 1][t], acc1, 0, 0, 0);
            }
        if (row < Co) {
            const uint32_t m0 = (uint32_t)tile * 16u;                          // (M < 2^31)
            const uint32_t b = m0 / (uint32_t)HW, pix = m0 - b * (uint32_t)HW; // (HW % 16 == 0: a tile stays inside one image)
            const f32x4 o = {acc0[0] + acc1[0] + bv, acc0[1] + acc1[1] + bv, acc0[2] + acc1[2] + bv, acc0[3] + acc1[3] + bv};
            *reinterpret_cast<f32x4*>(y + ((int64_t)b * Co + row) * HW + pix + 4 * kq) = o;
        }
        stage = stage + 1 == HF_NST ? 0 : stage + 1;
    }
    wait_vmcnt<0>();                                                           // (past-the-end DMAs still landing in this wave's LDS)
#undef HF_ISSUE
}

// head data-gradient: dx[m][c] = sum_co dy[b][co][pix] * w[co][c]   (NCHW grad in, NHWC out)
constexpr int HEAD_WG_PIX = 1024;   // pixels per block of the head weight-gradient kernels
template <typename T>      // T = float, or uint16_t: dx is stored as bf16 (mixed-precision training)
__global__ __launch_bounds__(256) void k_head_dgrad(const float* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx,
                                                     int64_t M, int HW, int C, int Co) {
    __shared__ float gs[HEAD_MAX_CO][64];
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* ws = lds;                 // [Co][C]
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    for (int i = threadIdx.x; i < Co * 64; i += 256) {
        const int co = i >> 6, px = i & 63;
        const int64_t m = m0 + px;
        float v = 0.f;
        if (m < M) { const int64_t b = m / HW, pix = m - b * HW; v = dy[(b * Co + co) * HW + pix]; }
        gs[co][px] = v;
    }
    for (int i = threadIdx.x; i < Co * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int c4 = C >> 2;
    for (int i = threadIdx.x; i < 64 * c4; i += 256) {
        const int row = i / c4, col = i - row * c4;
        const int64_t m = m0 + row;
        if (m >= M) continue;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int co = 0; co < Co; ++co) {
            const float g = gs[co][row];
            const float4 ww = *reinterpret_cast<const float4*>(ws + co * C + col * 4);
            o.x += g * ww.x; o.y += g * ww.y; o.z += g * ww.z; o.w += g * ww.w;
        }
        if constexpr (sizeof(T) == 2) {
            uint2 pk;
            pk.x = (uint32_t)f32_to_bf16(o.x) | ((uint32_t)f32_to_bf16(o.y) << 16);
            pk.y = (uint32_t)f32_to_bf16(o.z) | ((uint32_t)f32_to_bf16(o.w) << 16);
            reinterpret_cast<uint2*>(dx + m * C)[col] = pk;
        } else {
            reinterpret_cast<float4*>(dx + m * C)[col] = o;
        }
    }
}

// head weight / bias gradient partials from a bf16 NHWC activation (mixed precision): a thread owns 8 consecutive channels (one 16-byte
// load per pixel: a wave-instruction reads 1 KB contiguous) and every 256 / (C / 8)-th pixel of a 64-pixel chunk, CO x 8 fp32 sums in
// registers; the pixel lanes are combined by wave shuffles and through LDS in a fixed order.  Same partial layout as k_head_wgrad.
template <int CO>
__global__ __launch_bounds__(256) void k_head_wgrad_bf16(const float* __restrict__ dy, const uint16_t* __restrict__ x, float* __restrict__ partial,
                                                          int64_t M, int HW, int C, int Co) {
    __shared__ float gs[CO][64];
    __shared__ float comb[4][CO * 128];               // per wave: [co][c] (C <= 128 here)
    const int64_t mb = (int64_t)blockIdx.x * HEAD_WG_PIX;
    const int CG = C >> 3, PL = 256 / CG;             // channel groups, pixel lanes (C = 128: 16 x 16; C = 64: 8 x 32)
    const int cg = threadIdx.x % CG, pl = threadIdx.x / CG;
    float acc[CO][8];
#pragma unroll
    for (int j = 0; j < CO; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[j][k] = 0.f;
    float bsum = 0.f;
    for (int64_t m0 = mb; m0 < min(mb + HEAD_WG_PIX, M); m0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < Co * 64; i += 256) {
            const int co = i >> 6, px = i & 63;
            const int64_t m = m0 + px;
            float v = 0.f;
            if (m < M) { const int64_t b = m / HW, pix = m - b * HW; v = dy[(b * Co + co) * HW + pix]; }
            gs[co][px] = v;
        }
        __syncthreads();
        for (int p0 = 0; p0 < 64; p0 += 4 * PL) {      // four 16-byte loads in flight per thread
            uint4 xv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = p0 + u * PL + pl;
                const int64_t m = m0 + px;
                xv[u] = reinterpret_cast<const uint4*>(x + (px < 64 && m < M ? m : mb) * C)[cg];      // masked rows meet gs == 0 (or are skipped below)
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int px = p0 + u * PL + pl;
                if (px >= 64) continue;
                const float f[8] = {__uint_as_float(xv[u].x << 16), __uint_as_float(xv[u].x & 0xffff0000u), __uint_as_float(xv[u].y << 16),
                                    __uint_as_float(xv[u].y & 0xffff0000u), __uint_as_float(xv[u].z << 16), __uint_as_float(xv[u].z & 0xffff0000u),
                                    __uint_as_float(xv[u].w << 16), __uint_as_float(xv[u].w & 0xffff0000u)};
#pragma unroll
                for (int j = 0; j < CO; ++j) {
                    if (j < Co) {
                        const float g = gs[j][px];
#pragma unroll
                        for (int k = 0; k < 8; ++k) acc[j][k] += g * f[k];
                    }
                }
            }
        }
        if (threadIdx.x < Co) {
            for (int px = 0; px < 64; ++px) bsum += gs[threadIdx.x][px];
        }
    }
    // pixel lanes of one wave (lanes with the same cg: stride CG), then the four waves in order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < CO; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = acc[j][k];
            for (int o = CG; o < 64; o <<= 1) v += __shfl_xor(v, o);
            acc[j][k] = v;
        }
    if (lane < CG) {
#pragma unroll
        for (int j = 0; j < CO; ++j)
            if (j < Co) {
#pragma unroll
                for (int k = 0; k < 8; ++k) comb[wave][j * C + lane * 8 + k] = acc[j][k];
            }
    }
    __syncthreads();
    float* dst = partial + (int64_t)blockIdx.x * (Co * C + Co);
    for (int i = threadIdx.x; i < Co * C; i += 256) dst[i] = (comb[0][i] + comb[1][i]) + (comb[2][i] + comb[3][i]);
    if (threadIdx.x < Co) dst[Co * C + threadIdx.x] = bsum;
}

// head weight/bias gradient partials: block handles HEAD_WG_PIX pixels in 64-pixel chunks; thread (c, half)
// accumulates Co sums over its half of every chunk (all 256 threads busy when C == 128), 8 loads in flight.
__global__ __launch_bounds__(256) void k_head_wgrad(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ partial,
                                                     int64_t M, int HW, int C, int Co) {
    __shared__ float gs[HEAD_MAX_CO][64];
    __shared__ float comb[HEAD_MAX_CO][256];
    const int64_t mb = (int64_t)blockIdx.x * HEAD_WG_PIX;
    const int halves = max(1, 256 / C);                 // pixel sub-ranges handled in parallel
    const int c = threadIdx.x % C, hf = threadIdx.x / C;
    const int span = 64 / halves;
    float acc[HEAD_MAX_CO];
#pragma unroll
    for (int j = 0; j < HEAD_MAX_CO; ++j) acc[j] = 0.f;
    float bsum = 0.f;                                   // thread co < Co also accumulates the bias gradient
    for (int64_t m0 = mb; m0 < min(mb + HEAD_WG_PIX, M); m0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < Co * 64; i += 256) {
            const int co = i >> 6, px = i & 63;
            const int64_t m = m0 + px;
            float v = 0.f;
            if (m < M) { const int64_t b = m / HW, pix = m - b * HW; v = dy[(b * Co + co) * HW + pix]; }
            gs[co][px] = v;
        }
        __syncthreads();
        if (hf < halves) {
            const int p0 = hf * span;
#pragma unroll 1
            for (int pb = 0; pb < span; pb += 8) {
                float xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t m = m0 + p0 + pb + u;
                    xv[u] = x[(m < M ? m : mb) * C + c];          // unconditional load; masked rows have gs == 0
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int j = 0; j < HEAD_MAX_CO; ++j)
                        if (j < Co) acc[j] += gs[j][p0 + pb + u] * xv[u];
            }
        }
        if (threadIdx.x < Co) {
            for (int px = 0; px < 64; ++px) bsum += gs[threadIdx.x][px];
        }
    }
    // combine the pixel halves
    for (int j = 0; j < Co; ++j) comb[j][threadIdx.x] = (hf < halves) ? acc[j] : 0.f;
    __syncthreads();
    float* dst = partial + (int64_t)blockIdx.x * (Co * C + Co);
    if (threadIdx.x < C) {
        for (int j = 0; j < Co; ++j) {
            float t = 0.f;
            for (int h2 = 0; h2 < halves; ++h2) t += comb[j][h2 * C + threadIdx.x];
            dst[j * C + threadIdx.x] = t;
        }
    }
    if (threadIdx.x < Co) dst[Co * C + threadIdx.x] = bsum;
}

// Head weight / bias gradient partials from an fp32 NHWC activation of 128 channels, Co <= 16, on v_mfma_f32_16x16x4_f32 (the reduction
// index of the MFMA is the PIXEL).  Same wave-private LDS-DMA ring as k_head_fwd_f32_c128 (tiles of 16 pixels, HF_NST stages, eight 1 KB
// pieces of x per tile, swizzled on the source side) plus one piece for dy: lane (co = n, kq) fetches dy[co][pixels 4 kq .. 4 kq + 3]
// from the NCHW plane into its own 16-byte slot.  Per tile: A = dy (one ds_read_b128 per lane), B = x: lane (n, kq) reads chunks n and
// 16 + n of pixels 4 kq + i (channels {4 n .. 4 n + 3, 64 + 4 n .. 64 + 4 n + 3}: 8 ds_read_b128, conflict-free), 32 MFMAs; x is read
// exactly once (537 MB at bs = 64).  k_head_wgrad (scalar loads, LDS-broadcast dy, 7 FMAs per loaded float): 343 us, 3x the stream's time.
// One partial row per WAVE, same layout as k_head_wgrad's (finished by k_head_wgrad_fin in a fixed order).
constexpr int HWG_STAGE = 2048 + 256;                                          // floats: 8 KB of x + 1 KB of dy
__global__ __launch_bounds__(256) void k_head_wgrad_f32_c128(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ partial,
                                                              int HW, int Co, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) float hw_lds[];             // 4 waves x HF_NST x 9 KB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* const T = hw_lds + wave * (HF_NST * HWG_STAGE);
    const uint32_t t_base = lds_addr(T);
    const int n = lane & 15, kq = lane >> 4;
    const int gw = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
    const int dr = lane >> 5, dsl = lane & 31;
    const int nco = min(n, Co - 1);                                            // (lanes n >= Co fetch a valid plane; their A operand is zeroed)
#define HWG_ISSUE(t_, st_)                                                                                         \
    {                                                                                                              \
        const uint32_t m0_ = (uint32_t)min((t_), ntiles - 1) * 16u;                      /* (M < 2^31) */        \
        const uint32_t b_ = m0_ / (uint32_t)HW, pix_ = m0_ - b_ * (uint32_t)HW;                                     \
        _Pragma("unroll") for (int d = 0; d < 8; ++d) {                                                            \
            const int r = 2 * d + dr;                                                                              \
            lds_dma16(x + ((int64_t)m0_ + r) * 128 + ((dsl ^ (r & 15)) << 2), T + (st_) * HWG_STAGE + d * 256);     \
        }                                                                                                          \
        lds_dma16(dy + ((int64_t)b_ * Co + nco) * HW + pix_ + 4 * kq, T + (st_) * HWG_STAGE + 2048);               \
    }
    f32x4 acc[8];
#pragma unroll
    for (int cb = 0; cb < 8; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    int tile = gw;
#pragma unroll
    for (int s = 0; s < HF_NST - 1; ++s) HWG_ISSUE(tile + s * nwaves, s)
    int stage = 0;
    for (; tile < ntiles; tile += nwaves) {
        const int nst = stage == 0 ? HF_NST - 1 : stage - 1;
        HWG_ISSUE(tile + (HF_NST - 1) * nwaves, nst)
        wait_vmcnt<9 * (HF_NST - 1)>();
        const uint32_t s0 = t_base + (uint32_t)stage * (HWG_STAGE * 4);
        f32x4 a = lds_read128_async<0>(s0 + 8192u + (uint32_t)lane * 16u);
        f32x4 xa[4], xb[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t row = (uint32_t)(4 * kq + i);
            xa[i] = lds_read128_async<0>(s0 + row * 512u + (((uint32_t)n ^ (row & 15u)) << 4));
            xb[i] = lds_read128_async<0>(s0 + row * 512u + (((uint32_t)(16 + n) ^ (row & 15u)) << 4));
        }
        SD_LDS_WAIT4(0, xa[0], xa[1], xa[2], xa[3]);
        SD_LDS_WAIT4(0, xb[0], xb[1], xb[2], xb[3]);
        SD_LDS_WAIT2(0, a, xa[0]);
        if (n >= Co) a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
                acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], xa[i][cb], acc[cb], 0, 0, 0);
                acc[4 + cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], xb[i][cb], acc[4 + cb], 0, 0, 0);
            }
        bsum += (a[0] + a[1]) + (a[2] + a[3]);
        stage = stage + 1 == HF_NST ? 0 : stage + 1;
    }
    wait_vmcnt<0>();
#undef HWG_ISSUE
    // D: lane (n' = n, q = kq) holds dW[co = 4 q + r][channel(n', cb)], channel = 4 n' + cb (cb < 4), 64 + 4 n' + cb - 4 (cb >= 4)
    float* dst = partial + (int64_t)gw * (Co * 128 + Co);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int co = 4 * kq + r;
        if (co < Co) {                                                         // (row length Co * 129: no 16-byte alignment -> scalar stores)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) { dst[co * 128 + 4 * n + cb] = acc[cb][r]; dst[co * 128 + 64 + 4 * n + cb] = acc[4 + cb][r]; }
        }
    }
    bsum += __shfl_xor(bsum, 16);
    bsum += __shfl_xor(bsum, 32);
    if (kq == 0 && n < Co) dst[Co * 128 + n] = bsum;
}

// 8 outputs x 32 lanes over the partial blocks per workgroup (deterministic order)
__global__ __launch_bounds__(256) void k_head_wgrad_fin(const float* __restrict__ partial, int nblocks, int n, float* __restrict__ dw,
                                                         float* __restrict__ db, int nw, int accumulate) {
    __shared__ double red[32][8];
    const int lo = threadIdx.x & 7, lr = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + lo;
    double s = 0.0;
    if (i < n)
        for (int b = lr; b < nblocks; b += 32) s += (double)partial[(int64_t)b * n + i];
    red[lr][lo] = s;
    __syncthreads();
    if (lr != 0 || i >= n) return;
    for (int k = 1; k < 32; ++k) s += red[k][lo];
    float* dst = i < nw ? dw + i : db + (i - nw);
    *dst = (accumulate ? *dst : 0.f) + (float)s;
}

// ------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam defaults: no weight decay, no amsgrad), one launch over the flat
// parameter buffer.  612 MB of traffic per step for 21.85 M parameters: HBM-bound.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int64_t n4, float lr, float b1, float b2, float eps, float bc1,
                                               float bc2_sqrt, float gscale) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        const float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
        float* P = &pp.x; const float* G = &gg.x; float* Mo = &mm.x; float* V = &vv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gr = G[j] * gscale;
            Mo[j] = b1 * Mo[j] + (1.f - b1) * gr;                     // exp_avg.lerp_(grad, 1 - beta1)
            V[j] = b2 * V[j] + (1.f - b2) * gr * gr;                  // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
            const float denom = sqrtf(V[j]) / bc2_sqrt + eps;
            P[j] -= (lr / bc1) * (Mo[j] / denom);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
    }
}

// ------------------------------------------------------------------------------------------
// bf16 backbone (inference): weight cast, max-pool and head reading bf16 NHWC activations
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f_(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf_(float v) { return __builtin_bit_cast(uint16_t, (__bf16)v); }

__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ x, uint16_t* __restrict__ y, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        reinterpret_cast<ushort4*>(y)[i] = make_ushort4(f2bf_(v.x), f2bf_(v.y), f2bf_(v.z), f2bf_(v.w));
    }
}

__global__ __launch_bounds__(256) void k_cast_f32(const uint16_t* __restrict__ x, float* __restrict__ y, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) st4(y, i, ld4(x, i));
}

__global__ __launch_bounds__(256) void k_maxpool_fwd_bf16(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int B, int Hi, int Wi,
                                                           int Ho, int Wo, int C) {
    const int cols = C >> 3;                       // 8 bf16 = 16 bytes per thread
    const int64_t n8 = (int64_t)B * Ho * Wo * cols;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int col = (int)(i % cols);
    int64_t t = i / cols;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int b = (int)(t / Ho);
    float m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = -INFINITY;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int iy = oy * 2 - 1 + r, ix = ox * 2 - 1 + s;
            if (iy < 0 || iy >= Hi || ix < 0 || ix >= Wi) continue;
            const uint4 v = reinterpret_cast<const uint4*>(x + (((int64_t)b * Hi + iy) * Wi + ix) * C)[col];
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                m[2 * j] = fmaxf(m[2 * j], __uint_as_float(w[j] << 16));
                m[2 * j + 1] = fmaxf(m[2 * j + 1], __uint_as_float(w[j] & 0xffff0000u));
            }
        }
    uint4 o;
    o.x = (uint32_t)f2bf_(m[0]) | ((uint32_t)f2bf_(m[1]) << 16); o.y = (uint32_t)f2bf_(m[2]) | ((uint32_t)f2bf_(m[3]) << 16);
    o.z = (uint32_t)f2bf_(m[4]) | ((uint32_t)f2bf_(m[5]) << 16); o.w = (uint32_t)f2bf_(m[6]) | ((uint32_t)f2bf_(m[7]) << 16);
    reinterpret_cast<uint4*>(y)[i] = o;
}

// head on a bf16 NHWC input: fp32 weights / accumulation / NCHW output (the decoder always runs in fp32)
__global__ __launch_bounds__(256) void k_head_fwd_bf16(const uint16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        float* __restrict__ y, int64_t M, int HW, int C, int Co) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int LD = C + 4;
    float* xs = lds;                 // [64][C+4] fp32
    float* ws = lds + 64 * LD;       // [Co][C]
    const int64_t m0 = (int64_t)blockIdx.x * 64;
    const int c8 = C >> 3;
    for (int i = threadIdx.x; i < 64 * c8; i += 256) {
        const int row = i / c8, col = i - row * c8;
        const int64_t m = m0 + row;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (m < M) v = reinterpret_cast<const uint4*>(x + m * C)[col];
        float* d = xs + row * LD + col * 8;
        d[0] = __uint_as_float(v.x << 16); d[1] = __uint_as_float(v.x & 0xffff0000u);
        d[2] = __uint_as_float(v.y << 16); d[3] = __uint_as_float(v.y & 0xffff0000u);
        d[4] = __uint_as_float(v.z << 16); d[5] = __uint_as_float(v.z & 0xffff0000u);
        d[6] = __uint_as_float(v.w << 16); d[7] = __uint_as_float(v.w & 0xffff0000u);
    }
    for (int i = threadIdx.x; i < Co * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int px = threadIdx.x & 63, grp = threadIdx.x >> 6;
    float acc[HEAD_MAX_CO / 4];
#pragma unroll
    for (int j = 0; j < HEAD_MAX_CO / 4; ++j) acc[j] = 0.f;
    for (int c = 0; c < C; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(xs + px * LD + c);
#pragma unroll
        for (int j = 0; j < HEAD_MAX_CO / 4; ++j) {
            const int co = grp + 4 * j;
            if (co < Co) {
                const float4 ww = *reinterpret_cast<const float4*>(ws + co * C + c);
                acc[j] += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
            }
        }
    }
    const int64_t m = m0 + px;
    if (m < M) {
        const int64_t b = m / HW, pix = m - b * HW;
#pragma unroll
        for (int j = 0; j < HEAD_MAX_CO / 4; ++j) {
            const int co = grp + 4 * j;
            if (co < Co) y[(b * Co + co) * HW + pix] = acc[j] + bias[co];
        }
    }
}

// Head on a bf16 NHWC input of 128 channels (the default FPN depth), wave-private pipeline: every wave walks tiles of 64 pixels; the
// tile (16 KB) comes in by LDS-DMA (16 wave-instructions of 1 KB, no staging registers, no block barrier) and is multiplied on the
// bf16 MFMA with the fp32 weights split into two bf16 terms, w = hi + lo (|w - hi - lo| <= 2^-17 |w|: the activations are bf16
// already, so x * hi + x * lo summed in fp32 equals the fp32 product to fp32 rounding).  Co <= 16: hi sits in columns 0 .. 15 and lo
// in columns 16 .. 31 of ONE 32-column B operand (the two halves are added across lanes at the end); Co <= 32: two MFMAs per step.
// The Co planes of the NCHW output are written 16 bytes (4 pixels) per lane.  k_head_fwd_bf16 staged 64 pixels per BLOCK, widened to
// fp32, and re-read every row once per output-channel group (24x the input bytes in LDS traffic): 186 us for 268 MB at bs=64.
//   LDS image of a tile: row r (pixel) x 16 slots of 16 bytes, slot c of row r at physical slot c ^ (r & 15): the DMA lane that lands in
//   physical slot s of row r fetches chunk s ^ (r & 15) (swizzle on the SOURCE), and the 16 lanes of a ds_read_b128 group (rows r ..
//   r+15, same chunk) hit 16 different slots.
typedef float head_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 head_bf16x8 __attribute__((ext_vector_type(8)));
template <bool SPLIT>
__global__ __launch_bounds__(256) void k_head_fwd_bf16_c128(const uint16_t* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                             float* __restrict__ y, int64_t M, int HW, int Co, int ntiles) {
    __shared__ __attribute__((aligned(16))) float lds[4 * 4096];              // 16 KB per wave
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* T = lds + wave * 4096;
    const int nwaves = gridDim.x * 4;
    const int rsub = lane >> 4, slot = lane & 15;                              // DMA: lane -> (row within a 4-row piece, physical slot)
    const int fr = lane & 31, fh = lane >> 5;
    // B operands: lane (n = fr, k-half fh) holds w[n][16 s + 8 fh .. + 7] of step s as bf16
    head_bf16x8 bh[8], bl[8];
    {
        const int n = SPLIT ? fr : (fr & 15);
        const bool lo_col = !SPLIT && fr >= 16;
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            uint16_t h[8], l[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float wv = n < Co ? w[n * 128 + 16 * st + 8 * fh + k] : 0.f;
                const uint16_t hi = f32_to_bf16(wv);
                const uint16_t lo = f32_to_bf16(wv - bf16_to_f32(hi));
                h[k] = lo_col ? lo : hi; l[k] = lo;
            }
            bh[st] = __builtin_bit_cast(head_bf16x8, h);
            bl[st] = __builtin_bit_cast(head_bf16x8, l);
        }
    }
    const float bv = (fr < Co && (SPLIT || fr < 16)) ? bias[fr] : 0.f;
    for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += nwaves) {
        const int64_t m0 = (int64_t)tile * 64;
#pragma unroll
        for (int pc = 0; pc < 16; ++pc) {
            const int r = pc * 4 + rsub;
            const int64_t m = min(m0 + r, M - 1);                              // (rows past the end re-read the last pixel; never stored)
#if defined(__HIP_DEVICE_COMPILE__)
            const uint16_t* src = x + m * 128 + ((slot ^ (r & 15)) << 3);
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(T + pc * 256), 16, 0, 0);
#else
            (void)m; (void)slot;
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        head_f32x16 acc[2], accl[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[mi][e] = 0.f; accl[mi][e] = 0.f; }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int r = mi * 32 + fr;
                const head_bf16x8 a = *reinterpret_cast<const head_bf16x8*>(reinterpret_cast<const char*>(T) + r * 256 + (((2 * st + fh) ^ (r & 15)) << 4));
                acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bh[st], acc[mi], 0, 0, 0);
                if (SPLIT) accl[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl[st], accl[mi], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // every LDS read of this tile is done before the next tile's DMA lands
        // C/D map of the 32x32 MFMA: column (output channel) = lane & 31, rows (pixels) (e & 3) + 8 (e >> 2) + 4 fh: four consecutive pixels per e >> 2
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 o;
                float* op = &o.x;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float hi = acc[mi][4 * g + k];
                    const float lo = SPLIT ? accl[mi][4 * g + k] : __shfl_down(hi, 16);      // column n + 16 holds x * lo
                    op[k] = hi + lo + bv;
                }
                const int64_t m = m0 + mi * 32 + 8 * g + 4 * fh;
                if (fr < Co && (SPLIT || fr < 16) && m < M) {
                    const int64_t b = m / HW, pix = m - b * HW;                // (HW % 4 == 0: the four pixels stay in one image)
                    *reinterpret_cast<float4*>(y + (b * Co + fr) * HW + pix) = o;
                }
            }
        }
    }
}

// ---- bf16 activations, 16 bytes per lane (8 elements): the mixed-precision training path's BatchNorm passes.  Same arithmetic
// as the templated 4-element kernels above (fp32, one rounding at the store; results are bit-identical to them) -- only the
// access width differs: 8-byte accesses left these HBM-bound passes at ~60 % of the rate of their fp32 (16-byte) versions.
struct F8 { float4 a, b; };
__device__ __forceinline__ F8 ld8(const uint16_t* p, int64_t i) {
    typedef unsigned sd_u32x4_nt __attribute__((ext_vector_type(4)));               // (read once: non-temporal, see ld4 in sd_common.h)
    const sd_u32x4_nt r_ = __builtin_nontemporal_load(reinterpret_cast<const sd_u32x4_nt*>(p) + i);
    const uint4 r = make_uint4(r_[0], r_[1], r_[2], r_[3]);
    F8 v;
    v.a = make_float4(bf16_to_f32((uint16_t)(r.x & 0xffff)), bf16_to_f32((uint16_t)(r.x >> 16)), bf16_to_f32((uint16_t)(r.y & 0xffff)), bf16_to_f32((uint16_t)(r.y >> 16)));
    v.b = make_float4(bf16_to_f32((uint16_t)(r.z & 0xffff)), bf16_to_f32((uint16_t)(r.z >> 16)), bf16_to_f32((uint16_t)(r.w & 0xffff)), bf16_to_f32((uint16_t)(r.w >> 16)));
    return v;
}
__device__ __forceinline__ void st8(uint16_t* p, int64_t i, const F8& v) {
    uint4 r;
    r.x = (uint32_t)f32_to_bf16(v.a.x) | ((uint32_t)f32_to_bf16(v.a.y) << 16); r.y = (uint32_t)f32_to_bf16(v.a.z) | ((uint32_t)f32_to_bf16(v.a.w) << 16);
    r.z = (uint32_t)f32_to_bf16(v.b.x) | ((uint32_t)f32_to_bf16(v.b.y) << 16); r.w = (uint32_t)f32_to_bf16(v.b.z) | ((uint32_t)f32_to_bf16(v.b.w) << 16);
    reinterpret_cast<uint4*>(p)[i] = r;
}

__global__ __launch_bounds__(256) void k_bn_apply_bf16x8(const uint16_t* __restrict__ x, uint16_t* __restrict__ y, int64_t n8, int C,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const uint16_t* __restrict__ res, int relu, uint8_t* __restrict__ mask) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int cols = C >> 3;
    const bool fixed = stride % cols == 0;                      // (see k_bn_apply: one column per thread; here EIGHT parameter vectors per element)
    const int col0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cols);
    float4 mu[2], is[2], g[2], b[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        mu[h] = reinterpret_cast<const float4*>(mean)[2 * col0 + h]; is[h] = reinterpret_cast<const float4*>(invstd)[2 * col0 + h];
        g[h] = reinterpret_cast<const float4*>(gamma)[2 * col0 + h]; b[h] = reinterpret_cast<const float4*>(beta)[2 * col0 + h];
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const F8 v = ld8(x, i);
        if (!fixed) {
            const int col = (int)(i % cols);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                mu[h] = reinterpret_cast<const float4*>(mean)[2 * col + h]; is[h] = reinterpret_cast<const float4*>(invstd)[2 * col + h];
                g[h] = reinterpret_cast<const float4*>(gamma)[2 * col + h]; b[h] = reinterpret_cast<const float4*>(beta)[2 * col + h];
            }
        }
        F8 o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 vv = h ? v.b : v.a;
            float4 oo;
            oo.x = (vv.x - mu[h].x) * is[h].x * g[h].x + b[h].x; oo.y = (vv.y - mu[h].y) * is[h].y * g[h].y + b[h].y;
            oo.z = (vv.z - mu[h].z) * is[h].z * g[h].z + b[h].z; oo.w = (vv.w - mu[h].w) * is[h].w * g[h].w + b[h].w;
            if (h) o.b = oo; else o.a = oo;
        }
        if (res) {
            const F8 r = ld8(res, i);
            o.a.x += r.a.x; o.a.y += r.a.y; o.a.z += r.a.z; o.a.w += r.a.w; o.b.x += r.b.x; o.b.y += r.b.y; o.b.z += r.b.z; o.b.w += r.b.w;
        }
        if (mask) {        // one byte per FOUR elements, as the 4-element kernels write it
            const uint8_t m0 = (uint8_t)((o.a.x > 0.f ? 1 : 0) | (o.a.y > 0.f ? 2 : 0) | (o.a.z > 0.f ? 4 : 0) | (o.a.w > 0.f ? 8 : 0));
            const uint8_t m1 = (uint8_t)((o.b.x > 0.f ? 1 : 0) | (o.b.y > 0.f ? 2 : 0) | (o.b.z > 0.f ? 4 : 0) | (o.b.w > 0.f ? 8 : 0));
            reinterpret_cast<uchar2*>(mask)[i] = make_uchar2(m0, m1);
        }
        if (relu) {
            o.a.x = fmaxf(o.a.x, 0.f); o.a.y = fmaxf(o.a.y, 0.f); o.a.z = fmaxf(o.a.z, 0.f); o.a.w = fmaxf(o.a.w, 0.f);
            o.b.x = fmaxf(o.b.x, 0.f); o.b.y = fmaxf(o.b.y, 0.f); o.b.z = fmaxf(o.b.z, 0.f); o.b.w = fmaxf(o.b.w, 0.f);
        }
        st8(y, i, o);
    }
}

// g = dy * relu mask (relu: 0 none, 2 recomputed from x, 3 mask bytes) for one float4 half
__device__ __forceinline__ float4 bn_mask4(float4 g, const float4 xv, const float4 mu, const float4 is, const float4 ga, const float4 be, int relu, uint8_t mb) {
    if (relu == 3) {
        g.x = (mb & 1) ? g.x : 0.f; g.y = (mb & 2) ? g.y : 0.f; g.z = (mb & 4) ? g.z : 0.f; g.w = (mb & 8) ? g.w : 0.f;
    } else if (relu == 2) {
        g.x = ((xv.x - mu.x) * is.x * ga.x + be.x) > 0.f ? g.x : 0.f; g.y = ((xv.y - mu.y) * is.y * ga.y + be.y) > 0.f ? g.y : 0.f;
        g.z = ((xv.z - mu.z) * is.z * ga.z + be.z) > 0.f ? g.z : 0.f; g.w = ((xv.w - mu.w) * is.w * ga.w + be.w) > 0.f ? g.w : 0.f;
    }
    return g;
}

__global__ __launch_bounds__(256) void k_bn_bwd_apply_bf16x8(const uint16_t* __restrict__ dy, const uint16_t* __restrict__ x, const uint8_t* __restrict__ maskb,
                                                              int relu, int64_t n8, int C, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ mg,
                                                              const float* __restrict__ mgx, uint16_t* __restrict__ dx, uint16_t* __restrict__ g_out) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int cols = C >> 3;
    const bool fixed = stride % cols == 0;                      // (see k_bn_apply: one column per thread; here up to TWELVE parameter vectors per element)
    const int col0 = (int)(((int64_t)blockIdx.x * 256 + threadIdx.x) % cols);
    float4 mu[2], is[2], ga[2], be[2], a[2], b[2];
    auto params = [&](int col) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            mu[h] = reinterpret_cast<const float4*>(mean)[2 * col + h]; is[h] = reinterpret_cast<const float4*>(invstd)[2 * col + h];
            ga[h] = reinterpret_cast<const float4*>(gamma)[2 * col + h];
            be[h] = relu == 2 ? reinterpret_cast<const float4*>(beta)[2 * col + h] : make_float4(0.f, 0.f, 0.f, 0.f);
            a[h] = reinterpret_cast<const float4*>(mg)[2 * col + h]; b[h] = reinterpret_cast<const float4*>(mgx)[2 * col + h];
        }
    };
    params(col0);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += stride) {
        const F8 gy = ld8(dy, i), xv = ld8(x, i);
        if (!fixed) params((int)(i % cols));
        uchar2 mb = make_uchar2(0, 0);
        if (relu == 3) mb = reinterpret_cast<const uchar2*>(maskb)[i];
        F8 o, gm;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float4 xx = h ? xv.b : xv.a;
            const float4 g = bn_mask4(h ? gy.b : gy.a, xx, mu[h], is[h], ga[h], be[h], relu, h ? mb.y : mb.x);
            float4 oo;
            oo.x = ga[h].x * is[h].x * (g.x - a[h].x - (xx.x - mu[h].x) * is[h].x * b[h].x);
            oo.y = ga[h].y * is[h].y * (g.y - a[h].y - (xx.y - mu[h].y) * is[h].y * b[h].y);
            oo.z = ga[h].z * is[h].z * (g.z - a[h].z - (xx.z - mu[h].z) * is[h].z * b[h].z);
            oo.w = ga[h].w * is[h].w * (g.w - a[h].w - (xx.w - mu[h].w) * is[h].w * b[h].w);
            if (h) { o.b = oo; gm.b = g; } else { o.a = oo; gm.a = g; }
        }
        st8(dx, i, o);
        if (g_out) st8(g_out, i, gm);
    }
}

// BatchNorm-backward reduction (k_col_reduce<1>) on bf16, 8 columns per lane: sum g, sum g * xhat per channel
__global__ __launch_bounds__(256) void k_col_reduce_bwd_bf16x8(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b, const uint8_t* __restrict__ maskb,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
                                                                int64_t M, int C, float* __restrict__ partial, int rpb) {
    __shared__ float4 red[4][256];
    const int cols = C >> 3;                       // 8-element columns (cols <= 128 for C <= 1024)
    const int lanes = 256 / cols;
    const int col = threadIdx.x % cols, rl = threadIdx.x / cols;
    const int64_t r0 = (int64_t)blockIdx.x * rpb, r1 = min(r0 + rpb, M);
    float4 s0[2], s1[2], mu[2], is[2], ga[2], be[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        s0[h] = s1[h] = ga[h] = be[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        mu[h] = reinterpret_cast<const float4*>(mean)[2 * col + h]; is[h] = reinterpret_cast<const float4*>(invstd)[2 * col + h];
        if (relu == 2) { ga[h] = reinterpret_cast<const float4*>(gamma)[2 * col + h]; be[h] = reinterpret_cast<const float4*>(beta)[2 * col + h]; }
    }
    if (rl < lanes) {
#pragma unroll 2
        for (int64_t r = r0 + rl; r < r1; r += lanes) {
            const F8 gy = ld8(a, r * cols + col), xv = ld8(b, r * cols + col);
            uchar2 mb = make_uchar2(0, 0);
            if (relu == 3) mb = reinterpret_cast<const uchar2*>(maskb)[r * cols + col];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 xx = h ? xv.b : xv.a;
                const float4 g = bn_mask4(h ? gy.b : gy.a, xx, mu[h], is[h], ga[h], be[h], relu, h ? mb.y : mb.x);
                s0[h].x += g.x; s0[h].y += g.y; s0[h].z += g.z; s0[h].w += g.w;
                s1[h].x += g.x * ((xx.x - mu[h].x) * is[h].x); s1[h].y += g.y * ((xx.y - mu[h].y) * is[h].y);
                s1[h].z += g.z * ((xx.z - mu[h].z) * is[h].z); s1[h].w += g.w * ((xx.w - mu[h].w) * is[h].w);
            }
        }
    }
    red[0][threadIdx.x] = s0[0]; red[1][threadIdx.x] = s0[1]; red[2][threadIdx.x] = s1[0]; red[3][threadIdx.x] = s1[1];
    __syncthreads();
    if (rl == 0) {
        for (int l = 1; l < lanes; ++l) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float4 u = red[h][l * cols + col], w = red[2 + h][l * cols + col];
                s0[h].x += u.x; s0[h].y += u.y; s0[h].z += u.z; s0[h].w += u.w;
                s1[h].x += w.x; s1[h].y += w.y; s1[h].z += w.z; s1[h].w += w.w;
            }
        }
        float* dst = partial + (int64_t)blockIdx.x * 2 * C;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            reinterpret_cast<float4*>(dst)[2 * col + h] = s0[h];
            reinterpret_cast<float4*>(dst + C)[2 * col + h] = s1[h];
        }
    }
}

static inline int ew_grid(int64_t n4) { return (int)std::min<int64_t>(cdiv(n4, 256), 256 * 16); }

}  // namespace sd

using namespace sd;

static thread_local int g_pool_pair = 1;              // sd_set_option("pool_fwd_pair", 0): one window per thread in sd_bn_relu_maxpool_fwd[_bf16] (A/B)
namespace sd { void sd_nn_set_pool_pair(int v) { g_pool_pair = v; } }  // (called by sd_set_option in sd_conv.hip)

extern "C" {

// rows per reduction block: 256 for the big maps, fewer for the deep layers so that the pass still spreads over >= 1024 blocks
// (layer4, M = 16384 at bs=64: 64 blocks of 256 rows read at 0.8 TB/s, 1024 blocks of 16 rows at HBM rate)
static int red_rows(int64_t M) {
    int r = RED_ROWS_PER_BLOCK;
    while (r > 16 && M / r < 1024) r >>= 1;
    return r;
}

static int fold_rows(int rows) { return rows > 2048 ? cdiv(rows, std::max(16, rows / 128)) : 0; }      // (<= 2048 rows: one finalize launch is faster)

// [nb partial rows][2][C] + the two per-channel means of the backward + the slab sums of the two-level finish
size_t sd_col_reduce_workspace_bytes(int64_t M, int C) {
    const int nb = cdiv(M, red_rows(M));
    return align_up(((size_t)nb + fold_rows(nb)) * 2 * C * sizeof(float) + 2 * (size_t)C * sizeof(float), 256);
}

// Many partial rows: fold them in coalesced slabs first (a column-slice finalize touches every row from every block); returns
// the rows the finalize kernel then reads (possibly moved to `scratch`).
static const float* fold_partials(const float* partial, int& rows, int C, float* scratch, hipStream_t st) {
    if (!scratch || fold_rows(rows) == 0 || 2 * C > 1024) return partial;
    const int slab = std::max(16, rows / 128), nb2 = cdiv(rows, slab);
    hipLaunchKernelGGL(k_rows_fold, dim3(nb2), dim3(256), 0, st, partial, rows, 2 * C, slab, scratch);
    rows = nb2;
    return scratch;
}

static int check_mc(const char* what, int64_t M, int C) {
    SD_REQUIRE(M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && (C / 4) <= 256 && 256 % (C / 4) == 0, SD_ERR_INVALID,
               "%s: needs M > 0 and C in {4..1024} with C/4 dividing 256 (got M=%lld C=%d)", what, (long long)M, C);
    return 0;
}

int sd_bn_train_stats(const float* x, int64_t M, int C, float eps, float momentum, float* running_mean, float* running_var,
                      float* mean, float* invstd, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_train_stats", M, C)) return e;
    SD_REQUIRE(x && mean && invstd && workspace, SD_ERR_INVALID, "sd_bn_train_stats: null pointer");
    SD_REQUIRE(workspace_bytes >= sd_col_reduce_workspace_bytes(M, C), SD_ERR_WORKSPACE, "sd_bn_train_stats: workspace too small");
    const int rpb = red_rows(M), nb = cdiv(M, rpb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_col_reduce<0>, dim3(nb), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, M, C, (float*)workspace, rpb);
    SD_LAUNCH_CHECK();
    int rows = nb;
    const float* fin = fold_partials((const float*)workspace, rows, C, (float*)workspace + ((size_t)nb + 1) * 2 * C, st);
    hipLaunchKernelGGL(k_col_finalize<0>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, eps, momentum,
                       mean, invstd, running_mean, running_var, (float*)nullptr, (float*)nullptr, 0);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_bn_train_stats_bf16(const void* x, int64_t M, int C, float eps, float momentum, float* running_mean, float* running_var,
                           float* mean, float* invstd, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_train_stats_bf16", M, C)) return e;
    SD_REQUIRE(x && mean && invstd && workspace, SD_ERR_INVALID, "sd_bn_train_stats_bf16: null pointer");
    SD_REQUIRE(workspace_bytes >= sd_col_reduce_workspace_bytes(M, C), SD_ERR_WORKSPACE, "sd_bn_train_stats_bf16: workspace too small");
    const int rpb = red_rows(M), nb = cdiv(M, rpb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((k_col_reduce<0, uint16_t>), dim3(nb), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)nullptr, (const uint16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, M, C, (float*)workspace, rpb);
    SD_LAUNCH_CHECK();
    int rows = nb;
    const float* fin = fold_partials((const float*)workspace, rows, C, (float*)workspace + ((size_t)nb + 1) * 2 * C, st);
    hipLaunchKernelGGL(k_col_finalize<0>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, eps, momentum,
                       mean, invstd, running_mean, running_var, (float*)nullptr, (float*)nullptr, 0);
    SD_LAUNCH_CHECK();
    return 0;
}

// rows of `scratch` the two-level finish of `rows` partial rows needs (0 = single level)
int sd_bn_finalize_scratch_rows(int rows) { return fold_rows(rows); }

int sd_bn_finalize_stats(const float* partial, int rows, int64_t M, int C, float eps, float momentum, float* running_mean,
                         float* running_var, float* mean, float* invstd, float* scratch, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_finalize_stats", M, C)) return e;
    SD_REQUIRE(partial && mean && invstd && rows > 0, SD_ERR_INVALID, "sd_bn_finalize_stats: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    // Many partial rows (one per conv tile: up to 8192): a column-slice finalize would touch every row from every block, so
    // the rows are first folded in coalesced slabs (full 2C-float rows per block), then the few slab sums are finished.
    partial = fold_partials(partial, rows, C, scratch, st);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_finalize<0>, dim3(cdiv(C, 4)), dim3(256), 0, st, partial, rows, C, (double)M, eps, momentum,
                       mean, invstd, running_mean, running_var, (float*)nullptr, (float*)nullptr, 0);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_bn_apply(const float* x, float* y, int64_t M, int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                const float* residual, int relu, uint8_t* relu_mask_out, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_apply", M, C)) return e;
    SD_REQUIRE(x && y && mean && invstd && gamma && beta, SD_ERR_INVALID, "sd_bn_apply: null pointer");
    const int64_t n4 = M * C / 4;
    hipLaunchKernelGGL(k_bn_apply<float>, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, x, y, n4, C, mean, invstd, gamma, beta, residual, relu,
                       relu_mask_out);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_bn_fold(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C, float* scale,
               float* shift, sd_stream_t stream) {
    SD_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, SD_ERR_INVALID, "sd_bn_fold: bad arguments");
    hipLaunchKernelGGL(k_bn_fold, dim3(cdiv(C, 256)), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, C, scale, shift);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_bn_bwd(const float* dy, const float* x, const float* y, int relu, int64_t M, int C, const float* mean, const float* invstd,
              const float* gamma, const float* beta, float* dx, float* g_out, float* dgamma, float* dbeta, int accumulate, void* workspace,
              size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_bwd", M, C)) return e;
    SD_REQUIRE(relu >= 0 && relu <= 3, SD_ERR_INVALID, "sd_bn_bwd: relu must be 0 (none), 1 (mask from y), 2 (mask recomputed from x) or 3 (mask bytes)");
    SD_REQUIRE(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && workspace && ((relu != 1 && relu != 3) || y) && (relu != 2 || beta),
               SD_ERR_INVALID, "sd_bn_bwd: null pointer");
    SD_REQUIRE(workspace_bytes >= sd_col_reduce_workspace_bytes(M, C), SD_ERR_WORKSPACE, "sd_bn_bwd: workspace too small");
    const int rpb = red_rows(M), nb = cdiv(M, rpb);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* mg = partial + (size_t)nb * 2 * C;
    float* mgx = mg + C;
    hipLaunchKernelGGL(k_col_reduce<1>, dim3(nb), dim3(256), 0, st, dy, x, y, mean, invstd, gamma, beta, relu, M, C, partial, rpb);
    SD_LAUNCH_CHECK();
    int rows = nb;
    const float* fin = fold_partials(partial, rows, C, mgx + C, st);
    hipLaunchKernelGGL(k_col_finalize<1>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, 0.f, 0.f, dgamma, dbeta,
                       (float*)nullptr, (float*)nullptr, mg, mgx, accumulate);
    SD_LAUNCH_CHECK();
    const int64_t n4 = M * C / 4;
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(ew_grid(n4)), dim3(256), 0, st, dy, x, y, relu, n4, C, mean, invstd, gamma, beta,
                       (const float*)mg, (const float*)mgx, dx, g_out);
    SD_LAUNCH_CHECK();
    return 0;
}

// second half of the reduction of sd_bn_bwd on caller-provided partial rows [rows][2][C] (sum g, sum g * xhat):
// dgamma / dbeta (+=) and means_out = [mean(g) (C), mean(g * xhat) (C)] for sd_bn_bwd_apply
int sd_bn_bwd_finalize(const float* partial, int rows, int64_t M, int C, float* dgamma, float* dbeta, int accumulate, float* means_out,
                       float* scratch, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_bwd_finalize", M, C)) return e;
    SD_REQUIRE(partial && dgamma && dbeta && means_out && rows > 0, SD_ERR_INVALID, "sd_bn_bwd_finalize: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const float* fin = fold_partials(partial, rows, C, scratch, st);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_col_finalize<1>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, 0.f, 0.f, dgamma, dbeta,
                       (float*)nullptr, (float*)nullptr, means_out, means_out + C, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

// apply pass of sd_bn_bwd with the two per-channel means already known (sd_conv2d_dgrad_bn_reduce / sd_bn_bwd_finalize)
int sd_bn_bwd_apply(const float* dy, const float* x, const float* y, int relu, int64_t M, int C, const float* mean, const float* invstd,
                    const float* gamma, const float* beta, const float* means, float* dx, float* g_out, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_bwd_apply", M, C)) return e;
    SD_REQUIRE(relu >= 0 && relu <= 3, SD_ERR_INVALID, "sd_bn_bwd_apply: relu must be 0 (none), 1 (mask from y), 2 (mask recomputed from x) or 3 (mask bytes)");
    SD_REQUIRE(dy && x && mean && invstd && gamma && means && dx && ((relu != 1 && relu != 3) || y) && (relu != 2 || beta), SD_ERR_INVALID,
               "sd_bn_bwd_apply: null pointer");
    const int64_t n4 = M * C / 4;
    hipLaunchKernelGGL(k_bn_bwd_apply<float>, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, dy, x, y, relu, n4, C, mean, invstd, gamma, beta,
                       means, means + C, dx, g_out);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_col_sum(const float* x, int64_t M, int C, float* out, int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_mc("sd_col_sum", M, C)) return e;
    SD_REQUIRE(x && out && workspace, SD_ERR_INVALID, "sd_col_sum: null pointer");
    SD_REQUIRE(workspace_bytes >= sd_col_reduce_workspace_bytes(M, C), SD_ERR_WORKSPACE, "sd_col_sum: workspace too small");
    const int rpb = red_rows(M), nb = cdiv(M, rpb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_col_reduce<2>, dim3(nb), dim3(256), 0, st, x, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, M, C, (float*)workspace, rpb);
    SD_LAUNCH_CHECK();
    int rows = nb;
    const float* fin = fold_partials((const float*)workspace, rows, C, (float*)workspace + ((size_t)nb + 1) * 2 * C, st);
    hipLaunchKernelGGL(k_col_finalize<2>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, 0.f, 0.f, out,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int B, int Hi, int Wi, int C, sd_stream_t stream) {
    SD_REQUIRE(x && y && idx && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 4 == 0, SD_ERR_INVALID, "sd_maxpool3x3s2_fwd: bad arguments");
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const int64_t n4 = (int64_t)B * Ho * Wo * C / 4;
    hipLaunchKernelGGL(k_maxpool_fwd, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, x, y, idx, B, Hi, Wi, Ho, Wo, C);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int B, int Hi, int Wi, int C, sd_stream_t stream) {
    SD_REQUIRE(dy && dx && idx && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 4 == 0, SD_ERR_INVALID, "sd_maxpool3x3s2_bwd: bad arguments");
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const int64_t n4 = (int64_t)B * Hi * Wi * C / 4;
    hipLaunchKernelGGL(k_maxpool_bwd, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, dy, idx, dx, B, Hi, Wi, Ho, Wo, C);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_bn_relu_maxpool_fwd(const float* x, int B, int Hi, int Wi, int C, const float* mean, const float* invstd, const float* gamma,
                           const float* beta, float* y_pool, uint8_t* idx, sd_stream_t stream) {
    SD_REQUIRE(x && mean && invstd && gamma && beta && y_pool && idx && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 4 == 0, SD_ERR_INVALID,
               "sd_bn_relu_maxpool_fwd: bad arguments");
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const int64_t n4 = (int64_t)B * Ho * Wo * C / 4;
    if (g_pool_pair && Wo % 2 == 0 && aligned16(x) && aligned16(y_pool) && (reinterpret_cast<uintptr_t>(idx) & 3u) == 0)
        hipLaunchKernelGGL((k_bn_relu_maxpool_fwd_pair<float, float, 4>), dim3(cdiv(n4 / 2, 256)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta,
                           y_pool, idx, B, Hi, Wi, Ho, Wo, C);
    else
    hipLaunchKernelGGL((k_bn_relu_maxpool_fwd<float, float>), dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta, y_pool, idx,
                       B, Hi, Wi, Ho, Wo, C);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_bn_relu_maxpool_fwd_bf16(const void* x_bf16, int B, int Hi, int Wi, int C, const float* mean, const float* invstd, const float* gamma,
                                const float* beta, void* y_pool_bf16, uint8_t* idx, sd_stream_t stream) {
    SD_REQUIRE(x_bf16 && mean && invstd && gamma && beta && y_pool_bf16 && idx && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 4 == 0, SD_ERR_INVALID,
               "sd_bn_relu_maxpool_fwd_bf16: bad arguments");
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const int64_t n4 = (int64_t)B * Ho * Wo * C / 4;
    if (g_pool_pair && Wo % 2 == 0 && C % 8 == 0 && aligned16(x_bf16) && aligned16(y_pool_bf16) && (reinterpret_cast<uintptr_t>(idx) & 7u) == 0)
        hipLaunchKernelGGL((k_bn_relu_maxpool_fwd_pair<uint16_t, uint16_t, 8>), dim3(cdiv(n4 / 4, 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16,
                           mean, invstd, gamma, beta, (uint16_t*)y_pool_bf16, idx, B, Hi, Wi, Ho, Wo, C);
    else
    hipLaunchKernelGGL((k_bn_relu_maxpool_fwd<uint16_t, uint16_t>), dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x_bf16, mean,
                       invstd, gamma, beta, (uint16_t*)y_pool_bf16, idx, B, Hi, Wi, Ho, Wo, C);
    SD_LAUNCH_CHECK();
    return 0;
}

extern "C++" {
template <typename XT, typename PT, typename DT = float>
static int maxpool_bn_relu_bwd_any(const char* what, const PT* dpool, const uint8_t* idx, const XT* x, int B, int Hi, int Wi, int C, const float* mean,
                                   const float* invstd, const float* gamma, const float* beta, DT* dx, float* dgamma, float* dbeta, int accumulate,
                                   void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    const int64_t M = (int64_t)B * Hi * Wi;
    if (int e = check_mc(what, M, C)) return e;
    SD_REQUIRE(dpool && idx && x && mean && invstd && gamma && beta && dx && dgamma && dbeta && workspace, SD_ERR_INVALID, "%s: null pointer", what);
    SD_REQUIRE(workspace_bytes >= sd_col_reduce_workspace_bytes(M, C), SD_ERR_WORKSPACE, "%s: workspace too small", what);
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const int nb = cdiv(M, RED_ROWS_PER_BLOCK);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* mg = partial + (size_t)nb * 2 * C;
    float* mgx = mg + C;
    const bool quad = Hi % 2 == 0 && Wi % 2 == 0;        // even maps: 2x2 quads share their four windows
    constexpr bool F32 = std::is_same<XT, float>::value && std::is_same<PT, float>::value && std::is_same<DT, float>::value;
    SD_REQUIRE(quad || F32, SD_ERR_INVALID, "%s: the bf16 form needs even Hi, Wi", what);
    if (quad) hipLaunchKernelGGL((k_pool_bn_bwd_reduce_quad<XT, PT>), dim3(nb), dim3(256), 0, st, dpool, idx, x, mean, invstd, gamma, beta, Hi, Wi, Ho, Wo, M / 4, C, partial);
    else if constexpr (F32) hipLaunchKernelGGL(k_pool_bn_bwd_reduce, dim3(nb), dim3(256), 0, st, dpool, idx, x, mean, invstd, gamma, beta, Hi, Wi, Ho, Wo, M, C, partial);
    SD_LAUNCH_CHECK();
    int rows = nb;
    const float* fin = fold_partials(partial, rows, C, mgx + C, st);
    hipLaunchKernelGGL(k_col_finalize<1>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, 0.f, 0.f, dgamma, dbeta,
                       (float*)nullptr, (float*)nullptr, mg, mgx, accumulate);
    SD_LAUNCH_CHECK();
    const int64_t n4 = M * C / 4;
    if (quad) hipLaunchKernelGGL((k_pool_bn_bwd_apply_quad<XT, PT, DT>), dim3(cdiv(n4 / 4, 256)), dim3(256), 0, st, dpool, idx, x, mean, invstd, gamma, beta,
                                 (const float*)mg, (const float*)mgx, Hi, Wi, Ho, Wo, n4 / 4, C, dx);
    else if constexpr (F32) hipLaunchKernelGGL(k_pool_bn_bwd_apply, dim3(cdiv(n4, 256)), dim3(256), 0, st, dpool, idx, x, mean, invstd, gamma, beta, (const float*)mg,
                                               (const float*)mgx, Hi, Wi, Ho, Wo, n4, C, dx);
    SD_LAUNCH_CHECK();
    return 0;
}
}  // extern "C++"

int sd_maxpool_bn_relu_bwd(const float* dpool, const uint8_t* idx, const float* x, int B, int Hi, int Wi, int C, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, int accumulate,
                           void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    return maxpool_bn_relu_bwd_any<float, float>("sd_maxpool_bn_relu_bwd", dpool, idx, x, B, Hi, Wi, C, mean, invstd, gamma, beta, dx, dgamma, dbeta,
                                                 accumulate, workspace, workspace_bytes, stream);
}

int sd_maxpool_bn_relu_bwd_bf16(const void* dpool_bf16, const uint8_t* idx, const void* x_bf16, int B, int Hi, int Wi, int C, const float* mean,
                                const float* invstd, const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta, int accumulate,
                                void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    return maxpool_bn_relu_bwd_any<uint16_t, uint16_t>("sd_maxpool_bn_relu_bwd_bf16", (const uint16_t*)dpool_bf16, idx, (const uint16_t*)x_bf16, B, Hi, Wi, C,
                                                       mean, invstd, gamma, beta, dx, dgamma, dbeta, accumulate, workspace, workspace_bytes, stream);
}

// the same with a bf16 input gradient out (what sd_conv2d_stem_wgrad_bf16 reads: under autocast the stem conv's output gradient is bf16)
int sd_maxpool_bn_relu_bwd_bf16_dx16(const void* dpool_bf16, const uint8_t* idx, const void* x_bf16, int B, int Hi, int Wi, int C, const float* mean,
                                     const float* invstd, const float* gamma, const float* beta, void* dx_bf16, float* dgamma, float* dbeta, int accumulate,
                                     void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    return maxpool_bn_relu_bwd_any<uint16_t, uint16_t, uint16_t>("sd_maxpool_bn_relu_bwd_bf16_dx16", (const uint16_t*)dpool_bf16, idx, (const uint16_t*)x_bf16,
                                                                 B, Hi, Wi, C, mean, invstd, gamma, beta, (uint16_t*)dx_bf16, dgamma, dbeta, accumulate,
                                                                 workspace, workspace_bytes, stream);
}

int sd_upsample2x_bwd(const float* dy, const float* add, float* dx, int B, int H, int W, int C, sd_stream_t stream) {
    SD_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, SD_ERR_INVALID, "sd_upsample2x_bwd: bad arguments");
    const int64_t n4 = (int64_t)B * H * W * C / 4;
    hipLaunchKernelGGL(k_up2_bwd<float>, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, dy, add, dx, B, H, W, C);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_cast_f32_to_bf16(const float* x, void* y, int64_t n, sd_stream_t stream) {
    SD_REQUIRE(x && y && n > 0 && n % 4 == 0 && aligned16(x), SD_ERR_INVALID, "sd_cast_f32_to_bf16: bad arguments (n %% 4 == 0, 16-byte aligned source)");
    hipLaunchKernelGGL(k_cast_bf16, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, x, (uint16_t*)y, n / 4);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_cast_bf16_to_f32(const void* x, float* y, int64_t n, sd_stream_t stream) {
    SD_REQUIRE(x && y && n > 0 && n % 4 == 0 && aligned16(y) && (reinterpret_cast<uintptr_t>(x) & 7u) == 0, SD_ERR_INVALID,
               "sd_cast_bf16_to_f32: bad arguments (n %% 4 == 0, 8-byte aligned source, 16-byte aligned destination)");
    hipLaunchKernelGGL(k_cast_f32, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, y, n / 4);
    SD_LAUNCH_CHECK();
    return 0;
}

// ---- bf16 activations (mixed-precision training): the same kernels with 2-byte loads / stores, all arithmetic in fp32 --------
int sd_bn_apply_bf16(const void* x, void* y, int64_t M, int C, const float* mean, const float* invstd, const float* gamma, const float* beta,
                     const void* residual, int relu, uint8_t* relu_mask_out, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_apply_bf16", M, C)) return e;
    SD_REQUIRE(x && y && mean && invstd && gamma && beta, SD_ERR_INVALID, "sd_bn_apply_bf16: null pointer");
    const int64_t n4 = M * C / 4;
    if (C % 8 == 0 && aligned16(x) && aligned16(y) && aligned16(residual)) {
        hipLaunchKernelGGL(k_bn_apply_bf16x8, dim3(ew_grid(n4 / 2)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n4 / 2, C, mean,
                           invstd, gamma, beta, (const uint16_t*)residual, relu, relu_mask_out);
        SD_LAUNCH_CHECK();
        return 0;
    }
    hipLaunchKernelGGL(k_bn_apply<uint16_t>, dim3(ew_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, n4, C, mean,
                       invstd, gamma, beta, (const uint16_t*)residual, relu, relu_mask_out);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_bn_bwd_bf16(const void* dy, const void* x, const void* y, int relu, int64_t M, int C, const float* mean, const float* invstd,
                   const float* gamma, const float* beta, void* dx, void* g_out, float* dgamma, float* dbeta, int accumulate, void* workspace,
                   size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_mc("sd_bn_bwd_bf16", M, C)) return e;
    SD_REQUIRE(relu >= 0 && relu <= 3, SD_ERR_INVALID, "sd_bn_bwd_bf16: relu must be 0 (none), 1 (mask from y), 2 (mask recomputed from x) or 3 (mask bytes)");
    SD_REQUIRE(dy && x && mean && invstd && gamma && dx && dgamma && dbeta && workspace && ((relu != 1 && relu != 3) || y) && (relu != 2 || beta),
               SD_ERR_INVALID, "sd_bn_bwd_bf16: null pointer");
    SD_REQUIRE(workspace_bytes >= sd_col_reduce_workspace_bytes(M, C), SD_ERR_WORKSPACE, "sd_bn_bwd_bf16: workspace too small");
    const int rpb = red_rows(M), nb = cdiv(M, rpb);
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)workspace;
    float* mg = partial + (size_t)nb * 2 * C;
    float* mgx = mg + C;
    const bool wide = C % 8 == 0 && relu != 1 && aligned16(dy) && aligned16(x) && aligned16(dx) && aligned16(g_out);
    if (wide) hipLaunchKernelGGL(k_col_reduce_bwd_bf16x8, dim3(nb), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x, (const uint8_t*)y, mean,
                                 invstd, gamma, beta, relu, M, C, partial, rpb);
    else hipLaunchKernelGGL((k_col_reduce<1, uint16_t>), dim3(nb), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x, (const uint16_t*)y, mean,
                            invstd, gamma, beta, relu, M, C, partial, rpb);
    SD_LAUNCH_CHECK();
    int rows = nb;
    const float* fin = fold_partials(partial, rows, C, mgx + C, st);
    hipLaunchKernelGGL(k_col_finalize<1>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, 0.f, 0.f, dgamma, dbeta,
                       (float*)nullptr, (float*)nullptr, mg, mgx, accumulate);
    SD_LAUNCH_CHECK();
    const int64_t n4 = M * C / 4;
    if (wide) hipLaunchKernelGGL(k_bn_bwd_apply_bf16x8, dim3(ew_grid(n4 / 2)), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x, (const uint8_t*)y,
                                 relu, n4 / 2, C, mean, invstd, gamma, beta, (const float*)mg, (const float*)mgx, (uint16_t*)dx, (uint16_t*)g_out);
    else hipLaunchKernelGGL(k_bn_bwd_apply<uint16_t>, dim3(ew_grid(n4)), dim3(256), 0, st, (const uint16_t*)dy, (const uint16_t*)x, (const uint16_t*)y,
                       relu, n4, C, mean, invstd, gamma, beta, (const float*)mg, (const float*)mgx, (uint16_t*)dx, (uint16_t*)g_out);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_col_sum_bf16(const void* x, int64_t M, int C, float* out, int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_mc("sd_col_sum_bf16", M, C)) return e;
    SD_REQUIRE(x && out && workspace, SD_ERR_INVALID, "sd_col_sum_bf16: null pointer");
    SD_REQUIRE(workspace_bytes >= sd_col_reduce_workspace_bytes(M, C), SD_ERR_WORKSPACE, "sd_col_sum_bf16: workspace too small");
    const int rpb = red_rows(M), nb = cdiv(M, rpb);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((k_col_reduce<2, uint16_t>), dim3(nb), dim3(256), 0, st, (const uint16_t*)x, (const uint16_t*)nullptr, (const uint16_t*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, M, C, (float*)workspace, rpb);
    SD_LAUNCH_CHECK();
    int rows = nb;
    const float* fin = fold_partials((const float*)workspace, rows, C, (float*)workspace + ((size_t)nb + 1) * 2 * C, st);
    hipLaunchKernelGGL(k_col_finalize<2>, dim3(cdiv(C, 4)), dim3(256), 0, st, fin, rows, C, (double)M, 0.f, 0.f, out,
                       (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_upsample2x_bwd_bf16(const void* dy, const void* add, void* dx, int B, int H, int W, int C, sd_stream_t stream) {
    SD_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, SD_ERR_INVALID, "sd_upsample2x_bwd_bf16: bad arguments");
    const int64_t n4 = (int64_t)B * H * W * C / 4;
    hipLaunchKernelGGL(k_up2_bwd<uint16_t>, dim3(cdiv(n4, 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)dy, (const uint16_t*)add,
                       (uint16_t*)dx, B, H, W, C);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_maxpool3x3s2_fwd_bf16(const void* x, void* y, int B, int Hi, int Wi, int C, sd_stream_t stream) {
    SD_REQUIRE(x && y && B > 0 && Hi > 0 && Wi > 0 && C > 0 && C % 8 == 0, SD_ERR_INVALID, "sd_maxpool3x3s2_fwd_bf16: bad arguments");
    const int Ho = (Hi + 2 - 3) / 2 + 1, Wo = (Wi + 2 - 3) / 2 + 1;
    const int64_t n8 = (int64_t)B * Ho * Wo * C / 8;
    hipLaunchKernelGGL(k_maxpool_fwd_bf16, dim3(cdiv(n8, 256)), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, (uint16_t*)y, B, Hi, Wi, Ho, Wo, C);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_head_fwd_bf16(const void* x, const float* w, const float* bias, float* y, int B, int HW, int C, int Co, sd_stream_t stream) {
    SD_REQUIRE(x && w && bias && y && B > 0 && HW > 0, SD_ERR_INVALID, "sd_head_fwd_bf16: bad arguments");
    SD_REQUIRE(C % 8 == 0 && C <= 512 && Co > 0 && Co <= HEAD_MAX_CO, SD_ERR_INVALID, "sd_head_fwd_bf16: needs C %% 8 == 0, C <= 512, Co <= %d", HEAD_MAX_CO);
    const int64_t M = (int64_t)B * HW;
    if (C == 128 && HW % 4 == 0 && aligned16(x) && aligned16(y)) {       // wave-private LDS-DMA + bf16 MFMA pipeline (the default FPN depth)
        const int ntiles = (int)cdiv(M, 64), blocks = std::max(1, std::min(cdiv(ntiles, 4), 1024));
        if (Co <= 16) hipLaunchKernelGGL(k_head_fwd_bf16_c128<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, w, bias, y, M, HW, Co, ntiles);
        else hipLaunchKernelGGL(k_head_fwd_bf16_c128<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x, w, bias, y, M, HW, Co, ntiles);
        SD_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = ((size_t)64 * (C + 4) + (size_t)Co * C) * sizeof(float);
    hipLaunchKernelGGL(k_head_fwd_bf16, dim3(cdiv(M, 64)), dim3(256), lds, (hipStream_t)stream, (const uint16_t*)x, w, bias, y, M, HW, C, Co);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_head_fwd(const float* x, const float* w, const float* bias, float* y, int B, int HW, int C, int Co, sd_stream_t stream) {
    SD_REQUIRE(x && w && bias && y && B > 0 && HW > 0, SD_ERR_INVALID, "sd_head_fwd: bad arguments");
    SD_REQUIRE(C % 4 == 0 && C <= 512 && Co > 0 && Co <= HEAD_MAX_CO, SD_ERR_INVALID, "sd_head_fwd: needs C %% 4 == 0, C <= 512, Co <= %d", HEAD_MAX_CO);
    const int64_t M = (int64_t)B * HW;
    if (C == 128 && Co <= 16 && HW % 16 == 0 && M < (1ll << 31) && aligned16(x) && aligned16(w) && aligned16(y)) {
        const int ntiles = (int)(M / 16);
        const size_t lds_b = (size_t)4 * HF_NST * 2048 * sizeof(float);
        static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_fwd_f32_c128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        SD_HIP(attr_once);
        hipLaunchKernelGGL(k_head_fwd_f32_c128, dim3(std::min(cdiv(ntiles, 4), 256)), dim3(256), lds_b, (hipStream_t)stream, x, w, bias, y, HW, Co, ntiles);
        SD_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = ((size_t)64 * (C + 4) + (size_t)Co * C) * sizeof(float);
    hipLaunchKernelGGL(k_head_fwd, dim3(cdiv(M, 64)), dim3(256), lds, (hipStream_t)stream, x, w, bias, y, M, HW, C, Co);
    SD_LAUNCH_CHECK();
    return 0;
}

// partial rows of the head weight gradient: one per block of k_head_wgrad (1024 pixels), or one per WAVE of k_head_wgrad_f32_c128
// (up to 1024 waves = one block of four waves per CU, each wave on tiles of 16 pixels)
static int head_wgrad_wave_rows(int64_t M) { return (int)std::min<int64_t>(1024, (cdiv(M, 16) + 3) / 4 * 4); }
size_t sd_head_bwd_workspace_bytes(int B, int HW, int C, int Co) {
    const int64_t M = (int64_t)B * HW;
    return align_up((size_t)std::max<int64_t>(cdiv(M, HEAD_WG_PIX), head_wgrad_wave_rows(M)) * (Co * C + Co) * sizeof(float), 256);
}

int sd_head_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* dbias, int B, int HW, int C, int Co,
                int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    SD_REQUIRE(dy && x && w && dx && dw && dbias && workspace && B > 0 && HW > 0, SD_ERR_INVALID, "sd_head_bwd: bad arguments");
    SD_REQUIRE(C % 4 == 0 && C <= 256 && (256 % C == 0) && (64 % (256 / C) == 0) && ((64 / (256 / C)) % 8 == 0) && Co > 0 && Co <= HEAD_MAX_CO,
               SD_ERR_INVALID, "sd_head_bwd: needs C in {64, 128, 256} and Co <= %d", HEAD_MAX_CO);
    SD_REQUIRE(workspace_bytes >= sd_head_bwd_workspace_bytes(B, HW, C, Co), SD_ERR_WORKSPACE, "sd_head_bwd: workspace too small");
    const int64_t M = (int64_t)B * HW;
    hipStream_t st = (hipStream_t)stream;
    // weight gradient first: it streams x while HBM is quiet; behind the data-gradient it shared the memory system with the write-back
    // of that kernel's 537 MB (144 us instead of ~105)
    int nb = cdiv(M, HEAD_WG_PIX);
    if (C == 128 && Co <= 16 && HW % 16 == 0 && M < (1ll << 31) && aligned16(x) && aligned16(dy)) {
        nb = head_wgrad_wave_rows(M);                                          // one partial row per wave
        const int blocks = nb / 4;
        const size_t lds_b = (size_t)4 * HF_NST * HWG_STAGE * sizeof(float);
        static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_head_wgrad_f32_c128), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        SD_HIP(attr_once);
        hipLaunchKernelGGL(k_head_wgrad_f32_c128, dim3(blocks), dim3(256), lds_b, st, dy, x, (float*)workspace, HW, Co, (int)(M / 16));
    } else {
        hipLaunchKernelGGL(k_head_wgrad, dim3(nb), dim3(256), 0, st, dy, x, (float*)workspace, M, HW, C, Co);
    }
    SD_LAUNCH_CHECK();
    const int n = Co * C + Co;
    hipLaunchKernelGGL(k_head_wgrad_fin, dim3(cdiv(n, 8)), dim3(256), 0, st, (const float*)workspace, nb, n, dw, dbias, Co * C, accumulate);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_head_dgrad<float>, dim3(cdiv(M, 64)), dim3(256), (size_t)Co * C * sizeof(float), st, dy, w, dx, M, HW, C, Co);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_head_bwd_bf16(const float* dy, const void* x_bf16, const float* w, void* dx_bf16, float* dw, float* dbias, int B, int HW, int C, int Co,
                     int accumulate, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    SD_REQUIRE(dy && x_bf16 && w && dx_bf16 && dw && dbias && workspace && B > 0 && HW > 0, SD_ERR_INVALID, "sd_head_bwd_bf16: bad arguments");
    SD_REQUIRE((C == 64 || C == 128) && Co > 0 && Co <= 8, SD_ERR_INVALID, "sd_head_bwd_bf16: needs C in {64, 128} and Co <= 8 (got %d, %d)", C, Co);
    SD_REQUIRE(aligned16(x_bf16) && aligned16(dx_bf16), SD_ERR_ALIGN, "sd_head_bwd_bf16: x and dx must be 16-byte aligned");
    SD_REQUIRE(workspace_bytes >= sd_head_bwd_workspace_bytes(B, HW, C, Co), SD_ERR_WORKSPACE, "sd_head_bwd_bf16: workspace too small");
    const int64_t M = (int64_t)B * HW;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_head_dgrad<uint16_t>, dim3(cdiv(M, 64)), dim3(256), (size_t)Co * C * sizeof(float), st, dy, w, (uint16_t*)dx_bf16, M, HW, C, Co);
    SD_LAUNCH_CHECK();
    const int nb = cdiv(M, HEAD_WG_PIX);
    hipLaunchKernelGGL(k_head_wgrad_bf16<8>, dim3(nb), dim3(256), 0, st, dy, (const uint16_t*)x_bf16, (float*)workspace, M, HW, C, Co);
    SD_LAUNCH_CHECK();
    const int n = Co * C + Co;
    hipLaunchKernelGGL(k_head_wgrad_fin, dim3(cdiv(n, 8)), dim3(256), 0, st, (const float*)workspace, nb, n, dw, dbias, Co * C, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int step, float lr, float beta1, float beta2,
                 float eps, float grad_scale, sd_stream_t stream) {
    SD_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && n % 4 == 0 && step >= 1, SD_ERR_INVALID, "sd_adam_step: bad arguments (n %% 4 == 0, step >= 1)");
    SD_REQUIRE(aligned16(param) && aligned16(grad) && aligned16(exp_avg) && aligned16(exp_avg_sq), SD_ERR_ALIGN, "sd_adam_step: pointers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(k_adam, dim3(ew_grid(n / 4)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, n / 4, lr, beta1, beta2, eps,
                       (float)bc1, (float)sqrt(bc2), grad_scale);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
