// fp32 convolution for the SDNet backbone / FPN on gfx950 as an implicit GEMM on the f32-input
// MFMA (v_mfma_f32_32x32x2_f32: exact fp32, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
// Replaces the ATen/cuDNN convolutions the reference reaches through torch.nn.Conv2d in
// src/sdnet/model/network.py:6-57 (Fpn, Head, Network) and torchvision's resnet34.
//
// Data layout: activations NHWC (torch channels_last), weights [Cout][R][S][Cin] (torch
// channels_last OIHW), so that every K-chunk of the GEMM (one filter tap x 32 input channels) is
// 128 contiguous bytes per output pixel (A operand) and per output channel (B operand).
//
//   GEMM:  Y[m][n] = sum_k A[m][k] * Wt[n][k],   m = (b, oy, ox), n = cout, k = (r, s, cin)
//   block tile 128 (m) x BN (n, 128 or 64) x 32 (k); 4 waves, each a 2x2 / 2x1 grid of 32x32 MFMA tiles
//   LDS: double-buffered A[128][32+4] and W[BN][32+4] (the +4 pad makes the ds_read_b128
//        fragment reads conflict-free: slot = 9*row mod 16), register-staged prefetch of chunk
//        k+1 while chunk k is multiplied (one barrier per chunk).
//
// The same kernel is the data-gradient: dX = conv(dY, W^T) with the coordinate map inverted
// (out = (in + pad - r) / stride when divisible), weights pre-transposed to [Cin][R][S][Cout].
#include <string.h>
#include "sd_common.h"
#include "sd_mfma.h"
#include "sd_conv_rows.h"

#ifndef SD_S2_LPT
#define SD_S2_LPT 1            // stride-2 data-gradient: parity classes in descending tap count (0: interleaved, the round-1 order)
#endif
namespace sd {


#ifndef SD_IGEMM_DMA
// 1 = stage the igemm tiles with global_load_lds (LDS-DMA): no staging VGPRs (167 -> 146), no ds_write pass.  For the prefetch
// to overlap the MFMAs each pipeline stage must be its OWN __shared__ object and the loop must name the stage statically
// (SD_ITER(0) / SD_ITER(1)): with one LDS array hipcc (ROCm 7.2) puts `s_waitcnt vmcnt(0)` in front of the first ds_read of
// every chunk, because an in-flight LDS-DMA is a pending write that may alias any read of the same object (measured: 2 %
// slower than register staging in that form, 1.5 % faster in this one).  0 = register staging + ds_write_b128.
#define SD_IGEMM_DMA 1
#endif
#ifndef SD_ABLATE_HOT
#define SD_ABLATE_HOT 0       // timing-only experiment: every staging load reads the (cache-hot) zero line -- WRONG RESULTS
#endif
#ifndef SD_ABLATE_STORE
#define SD_ABLATE_STORE 0     // timing-only experiment: the epilogue stores (almost) nothing -- WRONG RESULTS
#endif
#ifndef SD_ABLATE_PATCH
#define SD_ABLATE_PATCH 0     // timing-only experiment: stage the A tile for ~1.6 of the 9 taps only (what patch staging would need) -- WRONG RESULTS
#endif
#if SD_ABLATE_HOT || SD_ABLATE_STORE || SD_ABLATE_PATCH || defined(SD_PP_ABL) || defined(SD_RS_ABL)
#warning "timing-only ablation build: this libsdnet_hip.so computes WRONG RESULTS; the Python loader refuses it unless SDNET_ALLOW_ABLATION=1"
#endif
// Reported through the C ABI (sd_build_flags): the loader, build() and the CPU tests assert 0, so an experiment build can never
// be mistaken for the product library.
#ifdef SD_DECODE_TRACE
#define SD_TRACE_FLAG 8       // csrc/sd_decode.hip compiled with in-kernel timestamps (slower, extra global stores)
#else
#define SD_TRACE_FLAG 0
#endif
#if defined(SD_PP_TRACE) || defined(SD_SB_TRACE)
#undef SD_TRACE_FLAG
#define SD_TRACE_FLAG 8
#endif
#if defined(SD_PP_ABL) || defined(SD_RS_ABL) || defined(SD_SB_ABL) || defined(SD_SHAPE_EXP) || defined(SD_W16_ABL) || defined(SD_SR_ABL) || defined(SD_SF_ABL)
#define SD_EXPERIMENT_FLAG 16  // timing-only ablations of the bf16 two-group / row-stream kernels (WRONG RESULTS)
#else
#define SD_EXPERIMENT_FLAG 0
#endif
void sd_nn_set_pool_pair(int v);      // sd_nn.hip
extern "C" int sd_build_flags(void) { return (SD_ABLATE_HOT ? 1 : 0) | (SD_ABLATE_STORE ? 2 : 0) | (SD_ABLATE_PATCH ? 4 : 0) | SD_TRACE_FLAG | SD_EXPERIMENT_FLAG; }

// Timing experiment (WRONG RESULTS; make SUFFIX=_shape EXTRA=-DSD_SHAPE_EXP=<mask>): the 32x32x16 bf16 MFMAs of the selected kernels are replaced
// by two 16x16x32 on the same operand registers (same flops, same LDS traffic) -- decides whether a kernel is worth converting to the shape the
// chip clocks higher on.  mask: 1 k_conv_igemm<bf16>, 2 k_conv3x3_patch<bf16>, 4 k_wgrad3x3_bf16, 8 k_wgrad_tap_bf16
#ifndef SD_SHAPE_EXP
#define SD_SHAPE_EXP 0
#else
#define SD_SHAPE_EXP_BUILD 1
#endif
__device__ __forceinline__ f32x16 mfma_shape_exp(bf16x8 a, bf16x8 b, f32x16 c) {
    f32x4 c0 = __builtin_shufflevector(c, c, 0, 1, 2, 3), c1 = __builtin_shufflevector(c, c, 4, 5, 6, 7);
    c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, c1, 0, 0, 0);
    c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3]; c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
    return c;
}
#define SD_MFMA_BF16(BIT, a, b, c) (((SD_SHAPE_EXP) & (BIT)) ? mfma_shape_exp(a, b, c) : __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0))

#ifndef SD_IGEMM_LATE_DMA
#define SD_IGEMM_LATE_DMA 0   // 1 = issue the next stage's DMA after the first MFMA group of the chunk (experiment)
#endif
constexpr int NBUF = 2;   // LDS stages of the igemm tiles (a single-buffer / 2-barrier variant was measured: no gain)
constexpr int BM = 128, BK = 32, LDK = BK;   // LDS rows are unpadded; 16-byte slots are XOR-swizzled by ((row >> 1) & 7)

struct ConvArgs {
    // x / w / y / res are fp32, or bf16 in the BF16 instantiation (inference with a bf16 backbone); the accumulator,
    // scale / shift and the split-K partials are always fp32
    const void* x;        // A source, NHWC [B][Hi][Wi][Ck]   (STEM: NCHW [B][3][Hi][Wi])
    const void* w;        // [Nn][R*S][Ck]
    void* y;              // [M][Nn], M = B*Ho*Wo
    const float* scale;   // per-n multiplier (nullable)
    const float* shift;   // per-n addend (nullable): bias or folded BN
    const void* res;      // residual, [M][Nn] or (res_up2 = 1: upsampled x2, 2: added at even (y, x) only) [B][Ho/2][Wo/2][Nn] (nullable)
    int B, Hi, Wi, Ck, Ho, Wo, Nn, R, S;
    int mul, div, off, rsign;   // input coord t = o*mul + off + rsign*r ; valid iff t>=0, t%div==0, t/div < Hi
    int relu, res_up2;
    int M, nk, kchunks;   // kchunks = Ck / (elements per 128-byte chunk: 32 fp32 or 64 bf16); nk = number of K chunks
    int splits;           // split-K (small-batch inference): blockIdx.y = K slice, raw partial tiles go to `part`
    float* part;          // [splits][M][Nn]
    // data-gradient + BatchNorm-backward reduction: the tensor this launch writes is the gradient w.r.t. the OUTPUT of a
    // BatchNorm(+ReLU) whose input was bn_x; the epilogue then also accumulates sum(g) and sum(g * xhat) per channel into
    // `stat` (g = v * relu mask, xhat = (bn_x - mean) * invstd) -- the reduction pass of sd_bn_bwd without re-reading dy
    const float* bn_x;    // [M][Nn] input of that BatchNorm (nullable = no fused reduction)
    const float* bn_y;    // its output, for the ReLU mask of residual layers (bn_relu == 1)
    const float *bn_mean, *bn_invstd, *bn_gamma, *bn_beta;
    int bn_relu;          // 0 none, 1 mask = bn_y > 0, 2 mask recomputed from bn_x
    float* stat;          // forward + BatchNorm statistics: per (m-tile, wave row) partial column sums [rows][2][Nn] of the
                          // raw conv output (sum, sum of squares), finished by sd_bn_finalize / k_col_finalize<0> (nullable)
    // k_conv3x3_patch geometry (host-computed): tile = 256 consecutive output pixels = 256/Wo whole rows of a 2^pt_tw_log2-wide map
    int pt_tw_log2, pt_pw, pt_pieces, pt_rolling, pt_flip;
    int pt_strip_log2;       // k_conv3x3_bf16_pp: log2 of the column strips a map row is cut into (0: a sub-tile's rows span the map)
    int par;              // stride-2 data-gradient: output pixels are grouped by (y&1, x&1) so that a tile only
                          // walks the filter taps that can reach its parity class (9/4 instead of 9 taps for 3x3)
    // k_conv3x3_bf16_pp with the 1x1 head fused into its epilogue (inference: sd_conv2d_fwd_bf16_head).  head_y != nullptr: the conv output
    // (all Nn = 128 channels of a pixel sit in one wave) is NOT stored; the head's head_co <= 32 channels go to head_y as NCHW fp32 planes
    const uint16_t* head_w;   // [2][32][128] bf16: hi and lo halves of the fp32 head weights (rows >= head_co zero): w = hi + lo to 2^-17
    const float* head_b;      // [32] bias (zero padded)
    float* head_y;            // [B][head_co][Ho * Wo]
    int head_co;
};

__device__ __attribute__((aligned(128))) float g_zero_line[64];   // zero-initialised: source of padded (out-of-image) rows


// Stride-2 data-gradient (MODE 2): tile of block `bid`.  The four output-parity classes have 4, 2, 2 and 1 filter taps: with the classes
// interleaved over the tile index the last blocks to start were as likely 4-tap tiles as 1-tap ones and ran next to idle slots.  Here
// every XCD (blocks bid, bid + 8, ... in dispatch order) walks its share of the 4-tap class first, then the 2-tap classes, then the 1-tap
// class -- longest first -- and inside a class a contiguous range of tiles (shared input rows hit the same L2).  `tile_m` keeps the old
// numbering 4 * (tile inside the class) + class: it only names the tile's statistics row.
__device__ __forceinline__ void s2_tile(int bid, int nwg, int n_tiles, int off, int& cls, int& tq, int& n_idx, int& tile_m) {
    if (SD_S2_LPT && (nwg & 31) == 0) {
        const int xcd = bid & 7, pos = bid >> 3, per = nwg >> 5;            // tiles per XCD and class
        const int k4 = pos / per, within = xcd * per + (pos - k4 * per);   // position inside the class, n fastest
        cls = (off & 1) ? (k4 == 0 ? 3 : k4 == 3 ? 0 : k4) : k4;           // taps per class: (2 - r0) * (2 - s0), r0 = (py + off) & 1
        tq = within / n_tiles; n_idx = within - tq * n_tiles;
    } else {
        const int tile = xcd_remap(bid, nwg), tm = tile / n_tiles;
        cls = tm & 3; tq = tm >> 2; n_idx = tile - tm * n_tiles;
    }
    tile_m = 4 * tq + cls;
}


__device__ __forceinline__ float f4c(const float4& v, int t) { return t == 0 ? v.x : (t == 1 ? v.y : (t == 2 ? v.z : v.w)); }

// K order: channel chunk outermost, filter taps innermost.  The nine taps of one channel chunk re-read (shifted) the same
// 128-byte input lines, so they hit the XCD's L2; with the taps outermost every tap walked the tile's whole Cin x pixels
// footprint (256 KB per block, 64 blocks per 4 MB L2) and the PMC showed 4-9x the input size in L2 fills per launch.
// MODE 0: unit "div" (forward conv of any stride, stride-1 data-gradient)   1: stem (NCHW image, K = 147 -> 160)
// MODE 2: stride-2 data-gradient, parity classes                             3: generic strided data-gradient
//
// NOTE on style: the staging registers are individual named variables filled by macros, not arrays written
// inside lambdas -- hipcc left such arrays in scratch memory (scratch_store after every global_load and a
// vmcnt(0) wait per load), which serialised the prefetch.
// float4 form of the fused BatchNorm-backward reduction term for one row (m) and four consecutive channels (n .. n+3)
struct BnRed4 { float4 mu, is, ga, be, s, q; };
__device__ __forceinline__ void bnred4_init(BnRed4& b, const ConvArgs& p, int n) {
    b.mu = *reinterpret_cast<const float4*>(p.bn_mean + n); b.is = *reinterpret_cast<const float4*>(p.bn_invstd + n);
    b.ga = b.be = b.s = b.q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bn_relu == 2) { b.ga = *reinterpret_cast<const float4*>(p.bn_gamma + n); b.be = *reinterpret_cast<const float4*>(p.bn_beta + n); }
}
__device__ __forceinline__ void bnred4_add(BnRed4& b, const ConvArgs& p, float4 g, int64_t m, int n) {
    const int64_t i = m * p.Nn + n;
    const float4 xv = *reinterpret_cast<const float4*>(p.bn_x + i);
    const float4 xh = make_float4((xv.x - b.mu.x) * b.is.x, (xv.y - b.mu.y) * b.is.y, (xv.z - b.mu.z) * b.is.z, (xv.w - b.mu.w) * b.is.w);
    if (p.bn_relu == 1) {
        const float4 yy = *reinterpret_cast<const float4*>(p.bn_y + i);
        g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f; g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
    } else if (p.bn_relu == 3) {
        const uint8_t mb = reinterpret_cast<const uint8_t*>(p.bn_y)[i >> 2];
        g.x = (mb & 1) ? g.x : 0.f; g.y = (mb & 2) ? g.y : 0.f; g.z = (mb & 4) ? g.z : 0.f; g.w = (mb & 8) ? g.w : 0.f;
    } else if (p.bn_relu == 2) {
        g.x = (xh.x * b.ga.x + b.be.x) > 0.f ? g.x : 0.f; g.y = (xh.y * b.ga.y + b.be.y) > 0.f ? g.y : 0.f;
        g.z = (xh.z * b.ga.z + b.be.z) > 0.f ? g.z : 0.f; g.w = (xh.w * b.ga.w + b.be.w) > 0.f ? g.w : 0.f;
    }
    b.s.x += g.x; b.s.y += g.y; b.s.z += g.z; b.s.w += g.w;
    b.q.x += g.x * xh.x; b.q.y += g.y * xh.y; b.q.z += g.z * xh.z; b.q.w += g.w * xh.w;
}
// sum over the lanes that share a column group (lane % LPR), result valid in lanes 0 .. LPR-1
template <int LPR>
__device__ __forceinline__ void bnred4_wave_sum(BnRed4& b) {
#pragma unroll
    for (int o = LPR; o < 64; o <<= 1) {
        b.s.x += __shfl_xor(b.s.x, o); b.s.y += __shfl_xor(b.s.y, o); b.s.z += __shfl_xor(b.s.z, o); b.s.w += __shfl_xor(b.s.w, o);
        b.q.x += __shfl_xor(b.q.x, o); b.q.y += __shfl_xor(b.q.y, o); b.q.z += __shfl_xor(b.q.z, o); b.q.w += __shfl_xor(b.q.w, o);
    }
}

// Epilogue helpers shared by k_conv_igemm and k_conv_igemm_big.
// SD_BNRED_TERM: contribution of one stored value v at (m, n) to the fused BatchNorm-backward reduction.
#define SD_BNRED_TERM(v, m, n, S, Q)                                                               \
    {                                                                                              \
        const float xv_ = p.bn_x[(int64_t)(m) * p.Nn + (n)];                                       \
        const float xh_ = (xv_ - bmu) * bis;                                                       \
        float g_ = (v);                                                                            \
        if (p.bn_relu == 1) g_ = p.bn_y[(int64_t)(m) * p.Nn + (n)] > 0.f ? g_ : 0.f;               \
        else if (p.bn_relu == 3) g_ = ((reinterpret_cast<const uint8_t*>(p.bn_y)[((int64_t)(m) * p.Nn + (n)) >> 2] >> ((n) & 3)) & 1) ? g_ : 0.f; \
        else if (p.bn_relu == 2) g_ = (xh_ * bga + bbe) > 0.f ? g_ : 0.f;                          \
        S += g_; Q += g_ * xh_;                                                                    \
    }

template <int BN, int MODE, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void k_conv_igemm(ConvArgs p) {
    constexpr bool STEM = (MODE == 1);
    using T = typename std::conditional<BF16, uint16_t, float>::type;
    constexpr int KE = BF16 ? 64 : 32;     // K elements per 128-byte chunk row
    constexpr int VE = BF16 ? 8 : 4;       // elements per 16-byte staging vector
    const T* const px_ = reinterpret_cast<const T*>(p.x);
    const T* const pw_ = reinterpret_cast<const T*>(p.w);
    const T* const zero_ = reinterpret_cast<const T*>(g_zero_line);
    constexpr int NT = BN / 64;            // 32-wide MFMA tiles per wave along n (wave tile = 64 x BN/2)
    // One __shared__ object per pipeline stage: hipcc's wait insertion treats an in-flight LDS-DMA as a pending write that
    // may alias any later ds_read OF THE SAME OBJECT; with separate objects the prefetch of stage k+1 is not waited for
    // before the fragment reads of stage k.
    __shared__ __attribute__((aligned(16))) float As0[BM * LDK];
    __shared__ __attribute__((aligned(16))) float As1[BM * LDK];
    __shared__ __attribute__((aligned(16))) float Bs0[BN * LDK];
    __shared__ __attribute__((aligned(16))) float Bs1[BN * LDK];
    __shared__ int orow[BM];               // output pixel of each tile row, -1 = none
    __shared__ int rrow[BM];               // its row in a half-size residual map (res_up2), -1 = no residual for this pixel

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_tiles = p.Nn / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tile_m = tile / n_tiles;
    int n0 = (tile % n_tiles) * BN;                      // n fastest: the A tile is reused from L2
    int m0 = tile_m * BM;

    int r0 = 0, s0 = 0, tstep = 1, nk = p.nk;
    int cls_base = 0, py = 0, px = 0;
    if (MODE == 2) {
        int cls, tq, n_idx;
        s2_tile(blockIdx.x, gridDim.x, n_tiles, p.off, cls, tq, n_idx, tile_m);
        n0 = n_idx * BN;
        const int Mq = p.M >> 2;
        cls_base = cls * Mq; m0 = cls_base + tq * BM;
        py = cls >> 1; px = cls & 1;
        r0 = (py + p.off) & 1; s0 = (px + p.off) & 1; tstep = 2;         // r = oy + pad (mod 2), likewise s
        nk = ((p.R - r0 + 1) >> 1) * ((p.S - s0 + 1) >> 1) * p.kchunks;
    }

    // ---- staging assignment: thread -> (row = tid/8 + 32*i, 4 consecutive k = (tid%8)*4)
    const int srow = tid >> 3, sk = (tid & 7) * 4;        // sk: LDS float offset of the staged 16-byte slot
    const int ske = (tid & 7) * VE;                        // the same slot in elements of the source tensors
    const int img_stride = p.Hi * p.Wi * (STEM ? 3 : p.Ck);
#define SD_ROW_SETUP(i)                                                                           \
    int aty##i = 0, atx##i = 0;                                                                   \
    const T* aptr##i = px_;                                                                       \
    bool aok##i;                                                                                  \
    {                                                                                             \
        const int m = m0 + srow + 32 * i;                                                         \
        int pix = -1;                                                                             \
        aok##i = m < p.M;                                                                         \
        if (aok##i) {                                                                             \
            int ox, oy, b;                                                                        \
            if (MODE == 2) {                                                                      \
                const int ml = m - cls_base, hw = p.Wo >> 1, hh = p.Ho >> 1;                      \
                const int qx = ml % hw, t = ml / hw, qy = t % hh;                                 \
                b = t / hh; oy = 2 * qy + py; ox = 2 * qx + px;                                   \
            } else {                                                                              \
                ox = m % p.Wo; const int t = m / p.Wo; oy = t % p.Ho; b = t / p.Ho;               \
            }                                                                                     \
            aty##i = oy * p.mul + p.off;                                                          \
            atx##i = ox * p.mul + p.off;                                                          \
            aptr##i = px_ + (int64_t)b * img_stride;                                              \
            pix = (b * p.Ho + oy) * p.Wo + ox;                                                    \
            if ((tid & 7) == 0)      /* once per tile row instead of a div/mod chain per output element */ \
                rrow[srow + 32 * i] = (p.res_up2 == 2 && ((ox | oy) & 1)) ? -1 : (b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1); \
        }                                                                                         \
        if ((tid & 7) == 0) orow[srow + 32 * i] = pix;                                            \
    }
    SD_ROW_SETUP(0) SD_ROW_SETUP(1) SD_ROW_SETUP(2) SD_ROW_SETUP(3)
#undef SD_ROW_SETUP
    const int wk = p.R * p.S * (STEM ? 3 : p.Ck);
    const T* wrow0 = pw_ + (int64_t)(n0 + srow) * wk;
    const T* wrow1 = wrow0 + (int64_t)32 * wk;
    const T* wrow2 = wrow0 + (int64_t)64 * wk;          // used when BN == 128
    const T* wrow3 = wrow0 + (int64_t)96 * wk;

    float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
    int ld_c0 = 0, ld_r = r0, ld_s = s0, ld_kc = 0;    // chunk cursor of the loader (chunks are loaded strictly in order)
    if (MODE == 0 && p.splits > 1) {
        // split-K: this block multiplies chunks [kbeg, kend) only
        const int per = (p.nk + p.splits - 1) / p.splits;
        const int kbeg = min((int)blockIdx.y * per, p.nk), kend = min(kbeg + per, p.nk);
        nk = kend - kbeg;
        const int ntap = p.R * p.S, cc = kbeg / ntap, tap = kbeg - cc * ntap;       // chunk index = channel chunk * taps + tap
        ld_c0 = cc * KE; ld_r = tap / p.S; ld_s = tap - ld_r * p.S;
    }

    // Padding rows read a zero line instead of being predicated: there is no select after the load, so the compiler
    // cannot turn the load back into a branch with a wait per load.
    constexpr bool DMA = (SD_IGEMM_DMA != 0) && !STEM && (NBUF == 2);
    // LDS-DMA: lane l of a wave-instruction writes 16 bytes at (wave-uniform base + 16 l) = row l/8, physical slot l%8 of
    // an 8-row group; the swizzle therefore goes on the SOURCE: this lane fetches logical k-slot (l%8) ^ swizzle(row).
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const int qe = (((tid & 7) ^ ((srow >> 1) & 7)) * VE);       // element offset of the logical slot this lane fetches
#define SD_DMA_A(i, buf)                                                                          \
    {                                                                                             \
        int ty = aty##i + p.rsign * ld_r, tx = atx##i + p.rsign * ld_s;                           \
        bool ok = aok##i;                                                                         \
        if (MODE == 2) { ty >>= 1; tx >>= 1; }                                                    \
        if (MODE == 3) {                                                                          \
            ok = ok && ty >= 0 && tx >= 0 && (ty % p.div == 0) && (tx % p.div == 0);              \
            ty /= p.div; tx /= p.div;                                                             \
        }                                                                                         \
        ok = ok && (unsigned)ty < (unsigned)p.Hi && (unsigned)tx < (unsigned)p.Wi;                \
        const T* src = (ok && !SD_ABLATE_HOT) ? aptr##i + ((ty * p.Wi + tx) * p.Ck + ld_c0 + qe) : zero_ + qe; \
        lds_dma16(src, ((buf) ? As1 : As0) + (32 * i + 8 * wave_u) * LDK);                        \
    }
#define SD_DMA_B(i, buf)                                                                          \
    lds_dma16(SD_ABLATE_HOT ? zero_ + qe : wrow##i + woffd, ((buf) ? Bs1 : Bs0) + (32 * i + 8 * wave_u) * LDK);
#define SD_DMA_CHUNK(buf)                                                                         \
    {                                                                                             \
        SD_DMA_A(0, buf) SD_DMA_A(1, buf) SD_DMA_A(2, buf) SD_DMA_A(3, buf)                       \
        const int woffd = (ld_r * p.S + ld_s) * p.Ck + ld_c0 + qe;                                \
        SD_DMA_B(0, buf) SD_DMA_B(1, buf)                                                         \
        if (BN == 128) { SD_DMA_B(2, buf) SD_DMA_B(3, buf) }                                      \
        ld_s += tstep;             /* taps innermost: see the note on the K order above the kernel */ \
        if (ld_s >= p.S) { ld_s = s0; ld_r += tstep; if (ld_r >= p.R) { ld_r = r0; ld_c0 += KE; } }  \
    }
#define SD_LOAD_A(i)                                                                              \
    {                                                                                             \
        int ty = aty##i + p.rsign * ld_r, tx = atx##i + p.rsign * ld_s;                           \
        bool ok = aok##i;                                                                         \
        if (MODE == 2) { ty >>= 1; tx >>= 1; }                                                    \
        if (MODE == 3) {                                                                          \
            ok = ok && ty >= 0 && tx >= 0 && (ty % p.div == 0) && (tx % p.div == 0);              \
            ty /= p.div; tx /= p.div;                                                             \
        }                                                                                         \
        ok = ok && (unsigned)ty < (unsigned)p.Hi && (unsigned)tx < (unsigned)p.Wi;                \
        const T* src = ok ? aptr##i + ((ty * p.Wi + tx) * p.Ck + ld_c0 + ske) : zero_ + ske;     \
        ra##i = *reinterpret_cast<const float4*>(src);                                            \
    }
#define SD_LOAD_A_STEM(i)                                                                         \
    {                                                                                             \
        float v[4];                                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            const int k = ld_kc * BK + sk + j;                                                    \
            const int tap = k / 3, ci = k - tap * 3;                                              \
            const int r = tap / p.S, s = tap - r * p.S;                                           \
            const int ty = aty##i + r, tx = atx##i + s;                                           \
            const bool ok = aok##i && k < wk && (unsigned)ty < (unsigned)p.Hi && (unsigned)tx < (unsigned)p.Wi; \
            const float* src = ok ? reinterpret_cast<const float*>(aptr##i) + ((ci * p.Hi + ty) * p.Wi + tx) : g_zero_line; \
            v[j] = *src;                                                                          \
        }                                                                                         \
        ra##i = make_float4(v[0], v[1], v[2], v[3]);                                              \
    }
#define SD_LOAD_B_STEM(i)                                                                         \
    {                                                                                             \
        float v[4];                                                                               \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                           \
            const int k = ld_kc * BK + sk + j;                                                    \
            const float* src = k < wk ? reinterpret_cast<const float*>(wrow##i) + k : g_zero_line; \
            v[j] = *src;                                                                          \
        }                                                                                         \
        rb##i = make_float4(v[0], v[1], v[2], v[3]);                                              \
    }
#define SD_LOAD_CHUNK()                                                                           \
    if (!STEM) {                                                                                  \
        SD_LOAD_A(0) SD_LOAD_A(1) SD_LOAD_A(2) SD_LOAD_A(3)                                       \
        const int woff = (ld_r * p.S + ld_s) * p.Ck + ld_c0 + ske;                                \
        rb0 = *reinterpret_cast<const float4*>(wrow0 + woff);                                     \
        rb1 = *reinterpret_cast<const float4*>(wrow1 + woff);                                     \
        if (BN == 128) {                                                                          \
            rb2 = *reinterpret_cast<const float4*>(wrow2 + woff);                                 \
            rb3 = *reinterpret_cast<const float4*>(wrow3 + woff);                                 \
        }                                                                                         \
        ld_s += tstep;             /* taps innermost: see the note on the K order above the kernel */ \
        if (ld_s >= p.S) { ld_s = s0; ld_r += tstep; if (ld_r >= p.R) { ld_r = r0; ld_c0 += KE; } }  \
    } else {                                                                                      \
        SD_LOAD_A_STEM(0) SD_LOAD_A_STEM(1) SD_LOAD_A_STEM(2) SD_LOAD_A_STEM(3)                   \
        SD_LOAD_B_STEM(0) SD_LOAD_B_STEM(1)                                                       \
        ++ld_kc;                                                                                  \
    }
#define SD_STORE_A(buf)                                                                           \
    {                                                                                             \
        float* ad = ((buf) ? As1 : As0) + srow * LDK + st_sk;                                     \
        *reinterpret_cast<float4*>(ad) = ra0;                                                     \
        *reinterpret_cast<float4*>(ad + 32 * LDK) = ra1;                                          \
        *reinterpret_cast<float4*>(ad + 64 * LDK) = ra2;                                          \
        *reinterpret_cast<float4*>(ad + 96 * LDK) = ra3;                                          \
    }
#define SD_STORE_B(buf)                                                                           \
    {                                                                                             \
        float* bd = ((buf) ? Bs1 : Bs0) + srow * LDK + st_sk;                                     \
        *reinterpret_cast<float4*>(bd) = rb0;                                                     \
        *reinterpret_cast<float4*>(bd + 32 * LDK) = rb1;                                          \
        if (BN == 128) {                                                                          \
            *reinterpret_cast<float4*>(bd + 64 * LDK) = rb2;                                      \
            *reinterpret_cast<float4*>(bd + 96 * LDK) = rb3;                                      \
        }                                                                                         \
    }
#define SD_STORE_CHUNK(buf) { SD_STORE_A(buf) SD_STORE_B(buf) }

    // LDS image: row r keeps its 16-byte k-slot q at slot q ^ ((r >> 1) & 7): a ds_read_b128 lane group (16 rows
    // distinct mod 16, same q) then touches 16 different slots of the 256-byte bank row -- conflict-free without
    // row padding (staged rows srow + 32 i and fragment rows tile + fr all share the swizzle of srow / fr).
    const int st_sk = (((sk >> 2) ^ ((srow >> 1) & 7)) << 2);

    // ---- wave tile: 64 (m) x BN/2 (n)
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * (BN / 2);
    const int fr = lane & 31, fh = lane >> 5;
    const int rd_swz = (fr >> 1) & 7;
    f32x16 acc[2][NT];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    if (nk > 0) {
        if (DMA) { SD_DMA_CHUNK(0) }
        else { SD_LOAD_CHUNK() SD_STORE_CHUNK(0) }
    }
    __syncthreads();                               // (also waits for the LDS-DMA: vmcnt(0) before the barrier)

#define SD_SLOT(ks) ((((ks) * 2 + fh) ^ rd_swz) << 2)
    // multiply one staged chunk: fragments of k-group ks+1 are read while the MFMAs of group ks run
#define SD_COMPUTE(AB, BB, CUR)                                                                                   \
    {                                                                                                             \
        const float* Ab = (AB) + (wm0 + fr) * LDK;                                                                \
        const float* Bb = (BB) + (wn0 + fr) * LDK;                                                                \
        float4 na0 = *reinterpret_cast<const float4*>(Ab + SD_SLOT(0)), na1 = *reinterpret_cast<const float4*>(Ab + 32 * LDK + SD_SLOT(0)); \
        float4 nb0 = *reinterpret_cast<const float4*>(Bb + SD_SLOT(0)), nb1 = nb0;                               \
        if (NT == 2) nb1 = *reinterpret_cast<const float4*>(Bb + 32 * LDK + SD_SLOT(0));                         \
        _Pragma("unroll") for (int ks = 0; ks < BK / 8; ++ks) {                                                   \
            const float4 a0 = na0, a1 = na1, b0 = nb0, b1 = nb1;                                                  \
            if (ks + 1 < BK / 8) {                                                                                \
                na0 = *reinterpret_cast<const float4*>(Ab + SD_SLOT(ks + 1));                                     \
                na1 = *reinterpret_cast<const float4*>(Ab + 32 * LDK + SD_SLOT(ks + 1));                          \
                nb0 = *reinterpret_cast<const float4*>(Bb + SD_SLOT(ks + 1));                                     \
                if (NT == 2) nb1 = *reinterpret_cast<const float4*>(Bb + 32 * LDK + SD_SLOT(ks + 1));             \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);      /* keep the prefetch reads ahead of this group's MFMAs */      \
            if (BF16) {   /* one 16-byte slot = 8 bf16 = this lane's k-half of a 32x32x16 MFMA step */            \
                const bf16x8 xa0 = __builtin_bit_cast(bf16x8, a0), xa1 = __builtin_bit_cast(bf16x8, a1);          \
                const bf16x8 xb0 = __builtin_bit_cast(bf16x8, b0), xb1 = __builtin_bit_cast(bf16x8, b1);          \
                acc[0][0] = SD_MFMA_BF16(1, xa0, xb0, acc[0][0]);                                              \
                acc[1][0] = SD_MFMA_BF16(1, xa1, xb0, acc[1][0]);                                              \
                if (NT == 2) {                                                                                    \
                    acc[0][NT - 1] = SD_MFMA_BF16(1, xa0, xb1, acc[0][NT - 1]);                                \
                    acc[1][NT - 1] = SD_MFMA_BF16(1, xa1, xb1, acc[1][NT - 1]);                                \
                }                                                                                                 \
            } else {      /* k-step outermost: consecutive MFMAs hit different accumulators */                    \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a0, t), f4c(b0, t), acc[0][0], 0, 0, 0); \
                    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a1, t), f4c(b0, t), acc[1][0], 0, 0, 0); \
                    if (NT == 2) {                                                                                \
                        acc[0][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a0, t), f4c(b1, t), acc[0][NT - 1], 0, 0, 0); \
                        acc[1][NT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a1, t), f4c(b1, t), acc[1][NT - 1], 0, 0, 0); \
                    }                                                                                             \
                }                                                                                                 \
            }                                                                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                                    \
            if (DMA && SD_IGEMM_LATE_DMA && ks == 0 && kc + 1 < nk) { SD_DMA_CHUNK(1 - CUR) }                      \
        }                                                                                                         \
    }
    // one pipeline step with the stage index as a literal, so that every LDS access names its __shared__ object
#define SD_ITER(CUR)                                                                                              \
    {                                                                                                             \
        if (!(DMA && SD_IGEMM_LATE_DMA) && kc + 1 < nk) {                                                         \
            if (DMA) { SD_DMA_CHUNK(1 - CUR) }      /* stage 1-CUR was last read before the previous barrier */   \
            else { SD_LOAD_CHUNK() }                                                                              \
        }                                                                                                         \
        SD_COMPUTE((CUR) ? As1 : As0, (CUR) ? Bs1 : Bs0, CUR)                                                     \
        if (!DMA && kc + 1 < nk) { SD_STORE_CHUNK(1 - CUR) }                                                      \
        __syncthreads();                                                                                          \
        ++kc;                                                                                                     \
    }
    int kc = 0;
    while (kc < nk) {
        SD_ITER(0)
        if (kc < nk) SD_ITER(1)
    }
#undef SD_ITER
#undef SD_COMPUTE
#undef SD_SLOT
#undef SD_LOAD_A
#undef SD_LOAD_A_STEM
#undef SD_LOAD_B_STEM
#undef SD_LOAD_CHUNK
#undef SD_STORE_CHUNK
#undef SD_STORE_A
#undef SD_STORE_B
#undef SD_DMA_A
#undef SD_DMA_B
#undef SD_DMA_CHUNK

    // ---- epilogue: C/D map of the 32x32 MFMA: n = lane&31, m = (e&3) + 8*(e>>2) + 4*(lane>>5)
    if (MODE == 0 && p.splits > 1) {
        // split-K partial tile, stored as rows through the wave's LDS region (16-byte stores, see the epilogue below)
        float* dst = p.part + (int64_t)blockIdx.y * p.M * p.Nn;
        constexpr int TWC = BN / 2;
        float* T = reinterpret_cast<float*>(BN == 128 ? (wave == 0 ? As0 : wave == 1 ? As1 : wave == 2 ? Bs0 : Bs1)
                                                      : (wave == 0 ? Bs0 : wave == 1 ? Bs1 : wave == 2 ? As0 : As1));
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) T[(mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * TWC + ni * 32 + fr] = acc[mi][ni][e];
        constexpr int LPR = TWC / 4, RPI = 64 / LPR;
        const int c4 = (lane % LPR) * 4;
#pragma unroll 4
        for (int it = 0; it < 64 / RPI; ++it) {
            const int row = it * RPI + lane / LPR;
            const int m = orow[wm0 + row];
            if (m >= 0) *reinterpret_cast<float4*>(dst + (int64_t)m * p.Nn + n0 + wn0 + c4) = *reinterpret_cast<const float4*>(T + row * TWC + c4);
        }
        return;
    }
    // Column sums for a BatchNorm that is fused with this launch (p.stat): forward = statistics of the raw conv output, from
    // the accumulators (rows past M staged zeros: they add nothing); data-gradient = the BatchNorm-backward reduction over
    // the values being stored.  Per wave over its 64 rows, the two wave rows combined through LDS, one partial row per tile.
    // (bf16 output: the statistics are those of the ROUNDED values, i.e. of the tensor the BatchNorm will normalise)
    const bool fwd_stat = MODE == 0 && p.stat && !p.bn_x;
    const bool bwd_red = !BF16 && p.stat && p.bn_x;
    float sv[NT], qv[NT];
#pragma unroll
    for (int ni = 0; ni < NT; ++ni) {
        sv[ni] = qv[ni] = 0.f;
        if (fwd_stat) {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float a = BF16 ? bf2f(f2bf(acc[mi][ni][e])) : acc[mi][ni][e]; sv[ni] += a; qv[ni] += a * a; }
        }
    }
    bool red_done = false;
    __shared__ float statred[2][BN];
    {
        // 16-byte epilogue (bf16 output: 8 bytes = four values per lane): a 32x32 MFMA accumulator holds 16 ROWS of one column per lane, so the direct form stores (and reads
        // the residual) 4 bytes per lane -- 64 + 64 vector-memory instructions per lane and tile, which is what bounds the 1x1
        // convs (FPN lateral with its upsample-add: 773 us vs 343 us for the plain conv).  The wave tile goes through the
        // (now idle) LDS stage buffers once -- each wave has a 64 x BN/2 float region of its own, no block barrier -- and comes
        // back as rows: 16 + 16 instructions of 16 bytes per lane.
        constexpr int TWC = BN / 2;                    // wave tile columns
        constexpr int LPR = TWC / 4, RPI = 64 / LPR;   // lanes per row, rows per pass
        const int c4 = (lane % LPR) * 4, n = n0 + wn0 + c4;
        float* T = reinterpret_cast<float*>(BN == 128 ? (wave == 0 ? As0 : wave == 1 ? As1 : wave == 2 ? Bs0 : Bs1)
                                                      : (wave == 0 ? Bs0 : wave == 1 ? Bs1 : wave == 2 ? As0 : As1));
        // bf16: the residual rows of the whole wave tile are requested BEFORE the trip through LDS (one memory round trip per
        // tile instead of one per four rows: the residual convs were 25 % slower than the plain ones for nothing but this latency)
        uint2 rr[64 / RPI] = {};
        float4 rf[64 / RPI];                           // the fp32 twin (FPN laterals: the upsample-add's rows come from a quarter-size map)
        const bool pre32 = !BF16 && p.res != nullptr;
        if constexpr (BF16) {
            if (p.res) {
#pragma unroll
                for (int it = 0; it < 64 / RPI; ++it) {
                    const int row = it * RPI + lane / LPR;
                    const int m = orow[wm0 + row];
                    const int64_t rm = m < 0 ? -1 : (p.res_up2 ? (int64_t)rrow[wm0 + row] : (int64_t)m);
                    // (no row: any valid address -- the value is dropped below; a select, not a branch around the load)
                    rr[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.res) + (rm >= 0 ? rm * p.Nn : 0) + n);
                }
            }
        } else if (pre32) {
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) {
                const int row = it * RPI + lane / LPR;
                const int m = orow[wm0 + row];
                const int64_t rm = m < 0 ? -1 : (p.res_up2 ? (int64_t)rrow[wm0 + row] : (int64_t)m);
                rf[it] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + (rm >= 0 ? rm * p.Nn : 0) + n);
            }
        }
#pragma unroll
        for (int ni = 0; ni < NT; ++ni)
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) T[(mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * TWC + ni * 32 + fr] = acc[mi][ni][e];
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n);
        if (p.shift) sh4 = *reinterpret_cast<const float4*>(p.shift + n);
        BnRed4 br;
        if (bwd_red) bnred4_init(br, p, n);
        auto out_row = [&](int it, uint2 r16, const float4* r32) {
            const int row = it * RPI + lane / LPR;
            const int m = orow[wm0 + row];
            if (m < 0) return;
            float4 v = *reinterpret_cast<const float4*>(T + row * TWC + c4);
            v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
            if (p.res) {
                const int64_t rm = p.res_up2 ? rrow[wm0 + row] : m;
                if (rm >= 0) {
                    if (BF16) {
                        v.x += bf2f((uint16_t)(r16.x & 0xffff)); v.y += bf2f((uint16_t)(r16.x >> 16));
                        v.z += bf2f((uint16_t)(r16.y & 0xffff)); v.w += bf2f((uint16_t)(r16.y >> 16));
                    } else {
                        const float4 r = r32 ? *r32 : *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + rm * p.Nn + n);
                        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                    }
                }
            }
            if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            if (BF16) {
                uint2 pk;
                pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
                pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
                *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (int64_t)m * p.Nn + n) = pk;
            } else {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (int64_t)m * p.Nn + n) = v;
                if (bwd_red) bnred4_add(br, p, v, m, n);
            }
        };
        if constexpr (BF16) {
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) out_row(it, rr[it], nullptr);
        } else if (pre32) {
#pragma unroll
            for (int it = 0; it < 64 / RPI; ++it) out_row(it, make_uint2(0u, 0u), &rf[it]);
        } else {
#pragma unroll 4
            for (int it = 0; it < 64 / RPI; ++it) out_row(it, make_uint2(0u, 0u), nullptr);
        }
        if (bwd_red) {      // BatchNorm-backward reduction: lanes -> wave (64 rows) -> the two wave rows -> one partial row per tile
            bnred4_wave_sum<LPR>(br);
            if ((wave >> 1) == 1 && lane < LPR) {
                *reinterpret_cast<float4*>(&statred[0][wn0 + c4]) = br.s; *reinterpret_cast<float4*>(&statred[1][wn0 + c4]) = br.q;
            }
            __syncthreads();
            if ((wave >> 1) == 0 && lane < LPR) {
                float* dst = p.stat + (int64_t)tile_m * 2 * p.Nn + n;
                const float4 a = *reinterpret_cast<const float4*>(&statred[0][wn0 + c4]), bq = *reinterpret_cast<const float4*>(&statred[1][wn0 + c4]);
                *reinterpret_cast<float4*>(dst) = make_float4(br.s.x + a.x, br.s.y + a.y, br.s.z + a.z, br.s.w + a.w);
                *reinterpret_cast<float4*>(dst + p.Nn) = make_float4(br.q.x + bq.x, br.q.y + bq.y, br.q.z + bq.z, br.q.w + bq.w);
            }
            red_done = true;
        }
    }
    if (fwd_stat || (bwd_red && !red_done)) {
#pragma unroll
        for (int ni = 0; ni < NT; ++ni) {
            sv[ni] += __shfl_xor(sv[ni], 32); qv[ni] += __shfl_xor(qv[ni], 32);
            if ((wave >> 1) == 1 && fh == 0) { statred[0][wn0 + ni * 32 + fr] = sv[ni]; statred[1][wn0 + ni * 32 + fr] = qv[ni]; }
        }
        __syncthreads();
        if ((wave >> 1) == 0 && fh == 0) {
            float* dst = p.stat + (int64_t)tile_m * 2 * p.Nn + n0;
#pragma unroll
            for (int ni = 0; ni < NT; ++ni) {
                const int c = wn0 + ni * 32 + fr;
                dst[c] = sv[ni] + statred[0][c]; dst[p.Nn + c] = qv[ni] + statred[1][c];
            }
        }
    }
}

// Epilogue of the 256-row tile kernels: y = [relu](acc * scale + shift [+ residual]) and the fused BatchNorm column sums (see
// k_conv_igemm): one partial row per tile.  row_to_m maps a tile row to its output pixel (-1 = none), row_to_res to its row in a
// half-size residual map (res_up2).
// T0 / T1 (VEC): two 32-row x 64-float LDS regions private to the wave (idle stage buffers) for the 16-byte form of the epilogue
// (see k_conv_igemm); the caller guarantees that no LDS-DMA is still landing in them.
template <int BN, int WM, int WN, int MT, int NTW, bool FWD, bool BF16 = false, bool VEC = false, typename RowMap, typename ResMap>
__device__ __forceinline__ void tile_epilogue(const ConvArgs& p, f32x16 (&acc)[MT][NTW], RowMap row_to_m, ResMap row_to_res, int tid, int wave, int fr,
                                              int fh, int wm0, int wn0, int n0, int tile_m, float* T0 = nullptr, float* T1 = nullptr) {
    const bool fwd_stat = FWD && p.stat && !p.bn_x;
    const bool bwd_red = !BF16 && p.stat && p.bn_x;
    float sv[NTW], qv[NTW];
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
        sv[ni] = qv[ni] = 0.f;
        if (fwd_stat) {
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float a = BF16 ? bf2f(f2bf(acc[mi][ni][e])) : acc[mi][ni][e]; sv[ni] += a; qv[ni] += a * a; }
        }
    }
    const int lane = tid & 63;
    bool red_done = false;
    __shared__ float statred[2][BN];
    if (VEC && NTW == 2) {
        const int c4 = (lane & 15) * 4, n = n0 + wn0 + c4;
        float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n);
        if (p.shift) sh4 = *reinterpret_cast<const float4*>(p.shift + n);
        BnRed4 br;
        if (bwd_red) bnred4_init(br, p, n);
        // passes of 64 rows (two 32-row LDS regions per wave) or, when the caller only has 8 KB per wave (T1 == nullptr), of 32 rows
        const bool two = T1 != nullptr;
        const int npass = two ? MT / 2 : MT, rpp = two ? 64 : 32;
        for (int h = 0; h < npass; ++h) {
            // bf16: the residual rows of the pass are requested before its trip through LDS (see k_conv_igemm's epilogue)
            uint2 rr[16] = {};
            if constexpr (BF16) {
                if (p.res) {
#pragma unroll
                    for (int it = 0; it < 16; ++it) {
                        if (it * 4 < rpp) {
                            const int trow = wm0 + h * rpp + it * 4 + (lane >> 4);
                            const int m = row_to_m(trow);
                            const int64_t rm = m < 0 ? -1 : (p.res_up2 ? (int64_t)row_to_res(trow, m) : (int64_t)m);
                            rr[it] = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.res) + (rm >= 0 ? rm * p.Nn : 0) + n);
                        }
                    }
                }
            }
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int r = (e & 3) + 8 * (e >> 2) + 4 * fh;
                    // (static accumulator indices: select by pass)
                    float v0 = 0.f, v1 = 0.f;
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi) {
                        if (two ? mi == 2 * h : mi == h) v0 = acc[mi][ni][e];
                        if (two && mi == 2 * h + 1) v1 = acc[mi][ni][e];
                    }
                    T0[r * 64 + ni * 32 + fr] = v0;
                    if (two) T1[r * 64 + ni * 32 + fr] = v1;
                }
            auto out_row = [&](int it, uint2 r16) {
                const int rl = it * 4 + (lane >> 4);                      // row inside this pass
                const int trow = wm0 + h * rpp + rl;
                const int m = row_to_m(trow);
                if (m < 0) return;
                float4 v = *reinterpret_cast<const float4*>((rl < 32 ? T0 : T1) + (rl & 31) * 64 + c4);
                v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
                if (p.res) {
                    const int64_t rm = p.res_up2 ? (int64_t)row_to_res(trow, m) : (int64_t)m;
                    if (rm >= 0) {
                        if (BF16) {
                            v.x += bf2f((uint16_t)(r16.x & 0xffff)); v.y += bf2f((uint16_t)(r16.x >> 16));
                            v.z += bf2f((uint16_t)(r16.y & 0xffff)); v.w += bf2f((uint16_t)(r16.y >> 16));
                        } else {
                            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + rm * p.Nn + n);
                            v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
                        }
                    }
                }
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (BF16) {
                    uint2 pk;
                    pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
                    pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (int64_t)m * p.Nn + n) = pk;
                } else {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(p.y) + (int64_t)m * p.Nn + n) = v;
                    if (bwd_red) bnred4_add(br, p, v, m, n);
                }
            };
            if constexpr (BF16) {
#pragma unroll
                for (int it = 0; it < 16; ++it)
                    if (it * 4 < rpp) out_row(it, rr[it]);
            } else {
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    if (it * 4 >= rpp) break;
                    out_row(it, make_uint2(0u, 0u));
                }
            }
        }
        if (bwd_red) {      // lanes -> wave -> wave rows 1 .. WM-1 into LDS in a fixed order -> wave row 0 writes the partial row
            bnred4_wave_sum<16>(br);
            if (tid < 2 * BN) statred[tid / BN][tid % BN] = 0.f;
            __syncthreads();
            for (int wr = 1; wr < WM; ++wr) {
                if (wave / WN == wr && lane < 16) {
                    float4* a = reinterpret_cast<float4*>(&statred[0][wn0 + c4]);
                    float4* bq = reinterpret_cast<float4*>(&statred[1][wn0 + c4]);
                    *a = make_float4(a->x + br.s.x, a->y + br.s.y, a->z + br.s.z, a->w + br.s.w);
                    *bq = make_float4(bq->x + br.q.x, bq->y + br.q.y, bq->z + br.q.z, bq->w + br.q.w);
                }
                __syncthreads();
            }
            if (wave / WN == 0 && lane < 16) {
                float* dst = p.stat + (int64_t)tile_m * 2 * p.Nn + n;
                const float4 a = *reinterpret_cast<const float4*>(&statred[0][wn0 + c4]), bq = *reinterpret_cast<const float4*>(&statred[1][wn0 + c4]);
                *reinterpret_cast<float4*>(dst) = make_float4(br.s.x + a.x, br.s.y + a.y, br.s.z + a.z, br.s.w + a.w);
                *reinterpret_cast<float4*>(dst + p.Nn) = make_float4(br.q.x + bq.x, br.q.y + bq.y, br.q.z + bq.z, br.q.w + bq.w);
            }
            red_done = true;
        }
    } else {
#pragma unroll
    for (int ni = 0; ni < NTW; ++ni) {
        const int n = n0 + wn0 + ni * 32 + fr;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float sh = p.shift ? p.shift[n] : 0.f;
        float bmu = 0.f, bis = 0.f, bga = 0.f, bbe = 0.f;
        if (bwd_red) {
            bmu = p.bn_mean[n]; bis = p.bn_invstd[n];
            if (p.bn_relu == 2) { bga = p.bn_gamma[n]; bbe = p.bn_beta[n]; }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int trow = wm0 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const int m = row_to_m(trow);
                if (m < 0) continue;
                float v = acc[mi][ni][e] * sc + sh;
                if (p.res) {
                    const int64_t rm = p.res_up2 ? row_to_res(trow, m) : m;      // half-size residual map: its row, -1 = none here
                    const bool has = rm >= 0;
                    if (has) v += BF16 ? bf2f(reinterpret_cast<const uint16_t*>(p.res)[rm * p.Nn + n]) : reinterpret_cast<const float*>(p.res)[rm * p.Nn + n];
                }
                if (p.relu) v = fmaxf(v, 0.f);
                if (BF16) reinterpret_cast<uint16_t*>(p.y)[(int64_t)m * p.Nn + n] = f2bf(v);
                else if (!SD_ABLATE_STORE || v == 123.456f) reinterpret_cast<float*>(p.y)[(int64_t)m * p.Nn + n] = v;
                if (bwd_red) SD_BNRED_TERM(v, m, n, sv[ni], qv[ni])
            }
        }
    }
    }
    if (fwd_stat || (bwd_red && !red_done)) {
        static_assert(WM == 2 || WM == 4, "wave rows 1 .. WM-1 are combined into row 0");
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) { sv[ni] += __shfl_xor(sv[ni], 32); qv[ni] += __shfl_xor(qv[ni], 32); }
        // wave rows 1 .. WM-1 add into LDS one after the other (fixed order -> deterministic), wave row 0 finishes
        if (tid < 2 * BN) statred[tid / BN][tid % BN] = 0.f;
        __syncthreads();
        for (int wr = 1; wr < WM; ++wr) {
            if (wave / WN == wr && fh == 0) {
#pragma unroll
                for (int ni = 0; ni < NTW; ++ni) { statred[0][wn0 + ni * 32 + fr] += sv[ni]; statred[1][wn0 + ni * 32 + fr] += qv[ni]; }
            }
            __syncthreads();
        }
        if (wave / WN == 0 && fh == 0) {
            float* dst = p.stat + (int64_t)tile_m * 2 * p.Nn + n0;
#pragma unroll
            for (int ni = 0; ni < NTW; ++ni) {
                const int c = wn0 + ni * 32 + fr;
                dst[c] = sv[ni] + statred[0][c]; dst[p.Nn + c] = qv[ni] + statred[1][c];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Large-tile fp32 implicit GEMM (MODE 0 / 2): block tile 256 (m) x BN, BK = 16, three LDS stages, counted vmcnt waits.
// Why: the staging path is as busy as the MFMA pipe in k_conv_igemm -- two resident 128x128x32 blocks ask a CU for 8 B/clk of
// staged data at full MFMA rate (128x64 tiles: 12 B/clk, which is why the 64-channel layers are the slowest).  Measured:
// every staging load redirected to one cache-hot line still ran 13 % below the no-load time; a dedicated producer wave
// (warp specialisation) was SLOWER, because one wave cannot feed more than ~1 piece per 300 clk.  A 256-row tile needs
// 25 % (BN = 128) / 17 % (BN = 64) fewer staged bytes per MFMA, and a 128x64 wave tile 25 % fewer ds_read_b128 per MFMA.
//   waves: BN = 128 -> 2 (m) x 2 (n), wave tile 128 x 64 (4 x 2 MFMA tiles, 128 accumulator VGPRs)
//          BN = 64  -> 4 (m) x 1 (n), wave tile  64 x 64
//   LDS rows are 16 floats (64 B); 16-byte slot q of row r lives at slot q ^ ((r >> 2) & 3) (conflict-free b128 reads).
//   Pipeline: chunk kc+2 is issued at the top of iteration kc, `s_waitcnt vmcnt(<pieces of chunk kc+2>)` + a bare s_barrier
//   end it (a __syncthreads() would carry a workgroup fence = vmcnt(0) and drain the stage in flight).
// ---------------------------------------------------------------------------------------------

constexpr int BMB = 256, BKB = 16;


// BF16 (round 4, opt-in: sd_set_option("igemm_big_bf16", 1)): the same tiles and pipeline on bf16 operands (64-byte LDS rows = 32 channels per chunk,
// one v_mfma_f32_32x32x16_bf16 per tile and k-group) for the stride-2 3x3 convs of the mixed-precision step / bf16 forward.  Not faster than
// k_conv_igemm<.., true> there (see g_igemm_big_bf16).
template <int BN, int MODE, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void k_conv_igemm_big(ConvArgs p) {
    using T = typename std::conditional<BF16, uint16_t, float>::type;
    constexpr int EPS = BF16 ? 8 : 4;                                    // elements per 16-byte slot
    constexpr int KC = BF16 ? 32 : BKB;                                  // channels per chunk
    constexpr int WM = BN == 128 ? 2 : 4, WN = BN == 128 ? 2 : 1;       // wave grid
    constexpr int MT = BMB / WM / 32, NTW = BN / WN / 32;                // 32x32 MFMA tiles per wave
    constexpr int PAW = BMB / 16 / 4, PBW = BN / 16 / 4;                 // 1 KB pieces (16 rows x 64 B) per wave and chunk
    constexpr int PW = PAW + PBW;
    constexpr int A_ST = BMB * BKB, B_ST = BN * BKB;
    constexpr int KSCALE = BK / BKB;                                     // ConvArgs counts 32-wide chunks (bf16: 64-wide, of 32 here)
    static_assert(MODE == 0 || MODE == 2, "unit-div coordinate maps only");
    static_assert(NTW == 2 && (MT == 4 || MT == 2), "wave tile 128x64 or 64x64");
    __shared__ __attribute__((aligned(16))) float As0[A_ST];
    __shared__ __attribute__((aligned(16))) float As1[A_ST];
    __shared__ __attribute__((aligned(16))) float As2[A_ST];
    __shared__ __attribute__((aligned(16))) float Bs0[B_ST];
    __shared__ __attribute__((aligned(16))) float Bs1[B_ST];
    __shared__ __attribute__((aligned(16))) float Bs2[B_ST];
    __shared__ int orow[BMB];
    __shared__ int rrow[BMB];              // row in a half-size residual map (res_up2), -1 = none
    const T* const px_ = reinterpret_cast<const T*>(p.x);
    const T* const pw_ = reinterpret_cast<const T*>(p.w);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = p.Nn / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    int tile_m = tile / n_tiles;
    int n0 = (tile % n_tiles) * BN;
    int m0 = tile_m * BMB;
    int r0 = 0, s0 = 0, tstep = 1, nk = p.nk * KSCALE;
    int cls_base = 0, py = 0, pxc = 0, m_end = p.M;
    if (MODE == 2) {
        int cls, tq, n_idx;
        s2_tile(blockIdx.x, gridDim.x, n_tiles, p.off, cls, tq, n_idx, tile_m);       // longest tiles first
        n0 = n_idx * BN;
        const int Mq = p.M >> 2;
        cls_base = cls * Mq; m0 = cls_base + tq * BMB; m_end = cls_base + Mq;
        py = cls >> 1; pxc = cls & 1;
        r0 = (py + p.off) & 1; s0 = (pxc + p.off) & 1; tstep = 2;
        nk = ((p.R - r0 + 1) >> 1) * ((p.S - s0 + 1) >> 1) * p.kchunks * KSCALE;
    }
    int ld_c0 = 0, ld_r = r0, ld_s = s0;
    // pixel of tile row `row`: (b, oy, ox); ok_ = the row exists
#define SD_BIG_PIXEL(row, b_, oy_, ox_, ok_)                                                       \
    int b_ = 0, oy_ = 0, ox_ = 0;                                                                  \
    const bool ok_ = m0 + (row) < m_end;                                                           \
    if (ok_) {                                                                                     \
        const int m = m0 + (row);                                                                  \
        if (MODE == 2) {                                                                           \
            const int ml = m - cls_base, hw = p.Wo >> 1, hh = p.Ho >> 1;                           \
            const int qx = ml % hw, t = ml / hw, qy = t % hh;                                      \
            b_ = t / hh; oy_ = 2 * qy + py; ox_ = 2 * qx + pxc;                                    \
        } else {                                                                                   \
            ox_ = m % p.Wo; const int t = m / p.Wo; oy_ = t % p.Ho; b_ = t / p.Ho;                 \
        }                                                                                          \
    }
    {
        SD_BIG_PIXEL(tid, b, oy, ox, ok)
        orow[tid] = ok ? (b * p.Ho + oy) * p.Wo + ox : -1;
        rrow[tid] = (!ok || (p.res_up2 == 2 && ((ox | oy) & 1))) ? -1 : (b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
    }

    // ---- staging: wave w owns A pieces 4w .. 4w+3 and B pieces PBW*w ..; lane -> row (lane / 4) of the piece, slot lane % 4.
    // Per row: pointer to the pixel of the FIRST tap at channel 0 (+ the lane's swizzled slot); per chunk a wave-uniform
    // offset (tap step, channel chunk) is added and the coordinates are range-checked; padding reads the zero line.
    const int prow = lane >> 2, pslot = lane & 3;
    const int qe = (pslot ^ ((prow >> 2) & 3)) * EPS;        // (row >> 2) & 3 == (prow >> 2) & 3: pieces start at multiples of 16
    const T* const zsrc = reinterpret_cast<const T*>(g_zero_line) + qe;
    const T* abase[PAW];
    int aty[PAW], atx[PAW];
#pragma unroll
    for (int j = 0; j < PAW; ++j) {
        const int row = (wave * PAW + j) * 16 + prow;
        SD_BIG_PIXEL(row, b, oy, ox, ok)
        int ty = oy * p.mul + p.off + p.rsign * r0, tx = ox * p.mul + p.off + p.rsign * s0;
        if (MODE == 2) { ty >>= 1; tx >>= 1; }               // (exact: the parity class makes both sums even)
        aty[j] = ok ? ty : -(1 << 28);                        // rows past the end fail every range check
        atx[j] = tx;
        abase[j] = px_ + (int64_t)b * p.Hi * p.Wi * p.Ck + ((int64_t)ty * p.Wi + tx) * p.Ck + qe;
    }
#undef SD_BIG_PIXEL
    const int wk = p.R * p.S * p.Ck;
    const T* bbase[PBW];
#pragma unroll
    for (int j = 0; j < PBW; ++j) bbase[j] = pw_ + (int64_t)(n0 + (wave * PBW + j) * 16 + prow) * wk + qe;
    const int tdiv = MODE == 2 ? 2 : 1;                       // taps advance by tstep, input coordinates by rsign * tstep / tdiv
    int issued = 0;                                           // chunks issued so far (past-the-end chunks stage zeros)
#define SD_BIG_ISSUE(AD, BD)                                                                       \
    {                                                                                              \
        if (issued < nk) {                                                                         \
            const int dy = p.rsign * ((ld_r - r0) / tdiv), dx = p.rsign * ((ld_s - s0) / tdiv);    \
            const int64_t aoff = ((int64_t)dy * p.Wi + dx) * p.Ck + ld_c0;                         \
            if (!SD_ABLATE_PATCH || (ld_r == r0 && ld_s == s0) || (ld_r == r0 + 1 && ld_s == s0 && ld_c0 == 0)) {   \
            _Pragma("unroll") for (int j = 0; j < PAW; ++j) {                                      \
                const bool ok = (unsigned)(aty[j] + dy) < (unsigned)p.Hi && (unsigned)(atx[j] + dx) < (unsigned)p.Wi; \
                lds_dma16(ok ? abase[j] + aoff : zsrc, (AD) + (wave * PAW + j) * 256);             \
            } }                                                                                    \
            const int woff = (ld_r * p.S + ld_s) * p.Ck + ld_c0;                                   \
            _Pragma("unroll") for (int j = 0; j < PBW; ++j) lds_dma16(bbase[j] + woff, (BD) + (wave * PBW + j) * 256); \
            ld_s += tstep;         /* taps innermost (K order note above k_conv_igemm) */           \
            if (ld_s >= p.S) { ld_s = s0; ld_r += tstep; if (ld_r >= p.R) { ld_r = r0; ld_c0 += KC; } } \
        } else {                                                                                   \
            _Pragma("unroll") for (int j = 0; j < PAW; ++j) lds_dma16(zsrc, (AD) + (wave * PAW + j) * 256); \
            _Pragma("unroll") for (int j = 0; j < PBW; ++j) lds_dma16(zsrc, (BD) + (wave * PBW + j) * 256); \
        }                                                                                          \
        ++issued;                                                                                  \
    }

    const int wm0 = (wave / WN) * (BMB / WM), wn0 = (wave % WN) * (BN / WN);
    const int fr = lane & 31, fh = lane >> 5;
    const int rd_swz = (fr >> 2) & 3;
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    SD_BIG_ISSUE(As0, Bs0)
    SD_BIG_ISSUE(As1, Bs1)
    wait_vmcnt<PW>();                                         // chunk 0 has landed (chunk 1 may be in flight)
    __builtin_amdgcn_s_barrier();

    // per-lane LDS byte offsets of the two k-groups' fragments inside a stage (A: rows wm0+fr+32 mi, B: rows wn0+fr+32 ni)
    const uint32_t a_k0 = ((wm0 + fr) * BKB + (((0 * 2 + fh) ^ rd_swz) << 2)) * 4, a_k1 = ((wm0 + fr) * BKB + (((1 * 2 + fh) ^ rd_swz) << 2)) * 4;
    const uint32_t b_k0 = ((wn0 + fr) * BKB + (((0 * 2 + fh) ^ rd_swz) << 2)) * 4, b_k1 = ((wn0 + fr) * BKB + (((1 * 2 + fh) ^ rd_swz) << 2)) * 4;
    constexpr int TSTR = 32 * BKB * 4;                        // bytes between consecutive 32-row MFMA tiles
#define SD_BIG_MFMA16(FA, FB, mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FA), __builtin_bit_cast(bf16x8, FB), acc[mi][ni], 0, 0, 0);
#define SD_BIG_MFMA(FA, FB, mi, ni)                                                                \
    _Pragma("unroll") for (int t = 0; t < 4; ++t) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[t], FB[t], acc[mi][ni], 0, 0, 0);
    // one chunk: all fragment reads of both k-groups are issued, then each group's MFMAs wait only for their own reads
#define SD_BIG_COMPUTE(AB, BB)                                                                     \
    {                                                                                              \
        const uint32_t ab = lds_addr(AB), bb = lds_addr(BB);                                       \
        f32x4 a00 = lds_read128_async<0>(ab + a_k0), a01 = lds_read128_async<TSTR>(ab + a_k0);     \
        f32x4 a02 = a00, a03 = a00;                                                                \
        if (MT == 4) { a02 = lds_read128_async<2 * TSTR>(ab + a_k0); a03 = lds_read128_async<3 * TSTR>(ab + a_k0); } \
        f32x4 b00 = lds_read128_async<0>(bb + b_k0), b01 = lds_read128_async<TSTR>(bb + b_k0);     \
        f32x4 a10 = lds_read128_async<0>(ab + a_k1), a11 = lds_read128_async<TSTR>(ab + a_k1);     \
        f32x4 a12 = a10, a13 = a10;                                                                \
        if (MT == 4) { a12 = lds_read128_async<2 * TSTR>(ab + a_k1); a13 = lds_read128_async<3 * TSTR>(ab + a_k1); } \
        f32x4 b10 = lds_read128_async<0>(bb + b_k1), b11 = lds_read128_async<TSTR>(bb + b_k1);     \
        if (MT == 4) { SD_LDS_WAIT6(6, a00, a01, a02, a03, b00, b01); } else { SD_LDS_WAIT4(4, a00, a01, b00, b01); } \
        if constexpr (BF16) {                                                                      \
            SD_BIG_MFMA16(a00, b00, 0, 0) SD_BIG_MFMA16(a01, b00, 1, 0)                            \
            if (MT == 4) { SD_BIG_MFMA16(a02, b00, MT - 2, 0) SD_BIG_MFMA16(a03, b00, MT - 1, 0) } \
            SD_BIG_MFMA16(a00, b01, 0, 1) SD_BIG_MFMA16(a01, b01, 1, 1)                            \
            if (MT == 4) { SD_BIG_MFMA16(a02, b01, MT - 2, 1) SD_BIG_MFMA16(a03, b01, MT - 1, 1) } \
        } else                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                            \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00[t], b00[t], acc[0][0], 0, 0, 0);  \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[t], b00[t], acc[1][0], 0, 0, 0);  \
            if (MT == 4) {                                                                         \
                acc[MT - 2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a02[t], b00[t], acc[MT - 2][0], 0, 0, 0); \
                acc[MT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a03[t], b00[t], acc[MT - 1][0], 0, 0, 0); \
            }                                                                                      \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a00[t], b01[t], acc[0][1], 0, 0, 0);  \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a01[t], b01[t], acc[1][1], 0, 0, 0);  \
            if (MT == 4) {                                                                         \
                acc[MT - 2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a02[t], b01[t], acc[MT - 2][1], 0, 0, 0); \
                acc[MT - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a03[t], b01[t], acc[MT - 1][1], 0, 0, 0); \
            }                                                                                      \
        }                                                                                          \
        if (MT == 4) { SD_LDS_WAIT6(0, a10, a11, a12, a13, b10, b11); } else { SD_LDS_WAIT4(0, a10, a11, b10, b11); } \
        if constexpr (BF16) {                                                                      \
            SD_BIG_MFMA16(a10, b10, 0, 0) SD_BIG_MFMA16(a11, b10, 1, 0)                            \
            if (MT == 4) { SD_BIG_MFMA16(a12, b10, MT - 2, 0) SD_BIG_MFMA16(a13, b10, MT - 1, 0) } \
            SD_BIG_MFMA16(a10, b11, 0, 1) SD_BIG_MFMA16(a11, b11, 1, 1)                            \
            if (MT == 4) { SD_BIG_MFMA16(a12, b11, MT - 2, 1) SD_BIG_MFMA16(a13, b11, MT - 1, 1) } \
        } else                                                                                     \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                            \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10[t], b10[t], acc[0][0], 0, 0, 0);  \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11[t], b10[t], acc[1][0], 0, 0, 0);  \
            if (MT == 4) {                                                                         \
                acc[MT - 2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a12[t], b10[t], acc[MT - 2][0], 0, 0, 0); \
                acc[MT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a13[t], b10[t], acc[MT - 1][0], 0, 0, 0); \
            }                                                                                      \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a10[t], b11[t], acc[0][1], 0, 0, 0);  \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a11[t], b11[t], acc[1][1], 0, 0, 0);  \
            if (MT == 4) {                                                                         \
                acc[MT - 2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a12[t], b11[t], acc[MT - 2][1], 0, 0, 0); \
                acc[MT - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a13[t], b11[t], acc[MT - 1][1], 0, 0, 0); \
            }                                                                                      \
        }                                                                                          \
    }
    // iteration with literal stage names: issue chunk kc+2 into the stage read at iteration kc-1, multiply chunk kc,
    // then make sure this wave's pieces of chunk kc+1 have landed before the barrier publishes them
#define SD_BIG_ITER(AC, BC, AN, BN_)                                                               \
    {                                                                                              \
        SD_BIG_ISSUE(AN, BN_)                                                                      \
        SD_BIG_COMPUTE(AC, BC)                                                                     \
        wait_vmcnt_and_lds<PW>();                                                                  \
        __builtin_amdgcn_s_barrier();                                                              \
        ++kc;                                                                                      \
    }
    int kc = 0;
    while (kc < nk) {
        SD_BIG_ITER(As0, Bs0, As2, Bs2)
        if (kc < nk) SD_BIG_ITER(As1, Bs1, As0, Bs0)
        if (kc < nk) SD_BIG_ITER(As2, Bs2, As1, Bs1)
    }
    wait_vmcnt<0>();
#undef SD_BIG_ITER
#undef SD_BIG_COMPUTE
#undef SD_BIG_MFMA
#undef SD_BIG_MFMA16
#undef SD_BIG_ISSUE

    // 16-byte epilogue through the idle stage buffers (BN = 128: three 16 KB A stages for waves 0..2, two 8 KB B stages for wave 3);
    // the barrier makes sure no other wave's (past-the-end) DMA is still landing in them
    __syncthreads();
    float* T0 = wave == 0 ? As0 : wave == 1 ? As1 : wave == 2 ? As2 : Bs0;
    float* T1 = wave == 3 ? Bs1 : T0 + 32 * 64;
    tile_epilogue<BN, WM, WN, MT, NTW, MODE == 0, BF16, BN == 128>(p, acc, [&](int row) { return orow[row]; }, [&](int row, int) { return rrow[row]; },
                                                                   tid, wave, fr, fh, wm0, wn0, n0, tile_m, T0, T1);
}
// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution (forward, and data-gradient with the taps flipped) with PATCH STAGING: the nine taps of
// a channel chunk read (shifted) the same input pixels, so the (rows + 2) x (cols + 2) input patch of the tile is staged ONCE per
// channel chunk and every tap's A fragments are read from it; only the weights are staged per tap.  Staged bytes per MFMA
// drop to 0.47x (BN = 128) / 0.56x (BN = 64) of k_conv_igemm_big's (staged bytes per flop).
//   tile  : 256 consecutive output pixels = 256/Wo whole rows (Wo in {16, 32, 64, 128}, Ho*Wo % 256 == 0) x BN channels
//   patch : LDS rows of 16 floats, one per patch pixel pp = py * PW + px, slot swizzle q ^ ((pp >> 2) & 3)
//           Wo <= 64: double-buffered (next chunk's patch lands while the nine taps of this one run)
//           Wo = 128: ONE buffer of 4 rows x 144 (padded) pixels, refilled row by row behind the taps that are done with a row
//                     (tap row r reads patch rows r, r+1): rows 2, 3 of THIS chunk go out at taps 0..3, rows 0, 1 of the NEXT
//                     chunk at taps 4..7
//   B     : three stages of BN rows x 16 floats, one per tap, ring position = tap % 3 (9 taps = 3 turns)
//   sync  : per tap one counted `s_waitcnt vmcnt(n)` (n = what this wave issued during the tap) + bare s_barrier
// ---------------------------------------------------------------------------------------------
#ifndef SD_PATCH_READS_FIRST_F32
#define SD_PATCH_READS_FIRST_F32 0      // fp32 patch kernel: 1 = fragment reads before the tap's LDS-DMA issue, as the bf16 instantiation does (A/B switch)
#endif
constexpr int PT_STAGE_FLOATS = 25 * 256;          // one double-buffer stage: 25 pieces of 1 KB (Wo = 64: 6 x 66 = 396 patch rows)
constexpr int PT_FLOATS = 2 * PT_STAGE_FLOATS;     // 51.2 KB; the rolling mode (36 pieces) uses it as one buffer
constexpr int PT_MAXP = 12;                        // patch pieces a wave can own (rolling: 4 rows x 3 slots)

// BF16: elements are bf16, a chunk is 32 channels (the same 64-byte LDS rows, one 32x32x16 MFMA per tile and k-group), forward only
// ROLL (the single rolling buffer of 128-pixel-wide maps, 12 pieces per wave) is a compile-time flag and its kernel a separate entry
// point (k_conv3x3_patch_roll): with the mode decided at run time every launch carried the 12 source pointers of the rolling mode
// (24 registers; the double-buffer mode owns at most 7 pieces per wave), the fp32 128-channel instantiation spilled four of them, and
// the reload sat in the tap loop: `scratch_load` + `s_waitcnt vmcnt(0)` in front of wave 0's seventh piece -- a drain of the whole
// prefetch stream once per chunk on maps with more than 24 pieces (layer2: 136 TFLOP/s against layer3's 140, same kernel).
template <int BN, bool BF16, bool ROLL>
__device__ __forceinline__ void conv3x3_patch_body(const ConvArgs& p) {
    using T = typename std::conditional<BF16, uint16_t, float>::type;
    constexpr int EPS = BF16 ? 8 : 4;                // elements per 16-byte slot
    constexpr int KC = BF16 ? 32 : BKB;              // channels per chunk
    constexpr int WM = BN == 128 ? 2 : 4, WN = BN == 128 ? 2 : 1;
    constexpr int MT = BMB / WM / 32, NTW = BN / WN / 32;
    constexpr int PBW = BN / 16 / 4;                 // weight pieces per wave and tap
    constexpr int B_ST = BN * BKB;
    static_assert(NTW == 2 && (MT == 4 || MT == 2), "wave tile 128x64 or 64x64");
    __shared__ __attribute__((aligned(16))) float Pt[PT_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[3 * B_ST];
    const T* const px_ = reinterpret_cast<const T*>(p.x);
    const T* const pw_ = reinterpret_cast<const T*>(p.w);
    const T* const zero_ = reinterpret_cast<const T*>(g_zero_line);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_tiles = p.Nn / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / n_tiles;
    const int n0 = (tile % n_tiles) * BN;
    const int m0 = tile_m * BMB;
    const int TWl = p.pt_tw_log2, TW = 1 << TWl, PW = p.pt_pw, TH = BMB >> TWl;
    // (the rolling entry point keeps the flag a run-time value: with it folded, hipcc's schedule of that instantiation spills 60 registers)
    const bool rolling = ROLL && p.pt_rolling != 0;
    constexpr int NPC = ROLL ? PT_MAXP : 7;           // patch pieces a wave can own in this mode (double buffer: <= 25 pieces per stage)
    const int hw = p.Ho * p.Wo;
    const int bimg = m0 / hw, y0 = (m0 - bimg * hw) >> TWl;
    const int nchunks = p.Ck / KC;

    // ---- patch pieces of this wave.  double-buffer: piece j = wave + 4 i;  rolling: i = 3 r + u -> j = 9 r + wave + 4 u (u = 2: wave 0)
    const int prow = lane >> 2, pslot = lane & 3;
    const T* pbase[NPC];
    unsigned okmask = 0, ownmask = 0;                 // okmask: the piece reads the image (else the zero line); ownmask: piece exists
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int j = rolling ? 9 * (i / 3) + wave + 4 * (i % 3) : wave + 4 * i;
        const bool own = rolling ? (wave + 4 * (i % 3) < 9) : (j < p.pt_pieces);
        const int pp = j * 16 + prow;
        const int py = pp / PW, pxx = pp - py * PW;
        const int iy = y0 - 1 + py, ix = pxx - 1;
        const bool ok = own && py < TH + 2 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const int qe = (pslot ^ ((pp >> 2) & 3)) * EPS;
        pbase[i] = ok ? px_ + (((int64_t)bimg * p.Hi + iy) * p.Wi + ix) * p.Ck + qe : zero_ + qe;
        // (ok differs per lane: keep it per lane in bit i; own is wave-uniform)
        okmask |= (ok ? 1u : 0u) << i;
        ownmask |= (own ? 1u : 0u) << i;
    }
    const int wk = 9 * p.Ck;
    const int qeb = (pslot ^ ((prow >> 2) & 3)) * EPS;
    const T* bbase[PBW];
#pragma unroll
    for (int j = 0; j < PBW; ++j) bbase[j] = pw_ + (int64_t)(n0 + (wave * PBW + j) * 16 + prow) * wk + qeb;
    const T* const zsrc = zero_ + qeb;

#define PT_PB(i) pbase[(i) < NPC ? (i) : 0]          /* (the other mode's piece indices only occur in compiled-out branches) */
#define PT_PATCH(i, dst, c0) { lds_dma16(((okmask >> (i)) & 1u) ? PT_PB(i) + (c0) : PT_PB(i), (dst) + (rolling ? 9 * ((i) / 3) + wave + 4 * ((i) % 3) : wave + 4 * (i)) * 256); }
#define PT_OWN(i) ((ownmask >> (i)) & 1u)
    // weights of (chunk ccn, loop tap t2) -> ring stage st; past the end: the zero line (keeps the per-tap issue count fixed).
    // The data-gradient walks the patch in the same order (rows 0..2: the rolling refill depends on it) with the weight taps
    // reversed: input pixel o + (1 - r, 1 - s) pairs with weight tap (r, s).
#define PT_ISSUE_B(ccn, t2, st)                                                                    \
    {                                                                                              \
        float* bd = Bs + (st) * B_ST;                                                              \
        if ((ccn) < nchunks) {                                                                     \
            const int woff = (p.pt_flip ? 8 - (t2) : (t2)) * p.Ck + (ccn) * KC;                    \
            _Pragma("unroll") for (int j = 0; j < PBW; ++j) lds_dma16(bbase[j] + woff, bd + (wave * PBW + j) * 256); \
        } else {                                                                                   \
            _Pragma("unroll") for (int j = 0; j < PBW; ++j) lds_dma16(zsrc, bd + (wave * PBW + j) * 256); \
        }                                                                                          \
    }

    // ---- MFMA side
    const int wm0 = (wave / WN) * (BMB / WM), wn0 = (wave % WN) * (BN / WN);
    const int fr = lane & 31, fh = lane >> 5;
    int bpp[MT];                                       // patch pixel of this lane's output pixel (tap 0, 0), per m-tile
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int ml = wm0 + mi * 32 + fr;
        bpp[mi] = (ml >> TWl) * PW + (ml & (TW - 1));
    }
    const int rd_swz_b = (fr >> 2) & 3;
    const uint32_t b_k0 = ((wn0 + fr) * BKB + (((0 * 2 + fh) ^ rd_swz_b) << 2)) * 4;
    constexpr int TSTR = 32 * BKB * 4;
    f32x16 acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mi][ni][e] = 0.f;

    // ---- prologue: weights of taps 0, 1 and the patch of chunk 0 (rolling: its rows 0 and 1)
    PT_ISSUE_B(0, 0, 0)
    PT_ISSUE_B(0, 1, 1)
#pragma unroll
    for (int i = 0; i < NPC; ++i)
        if (PT_OWN(i) && (!rolling || i < 6)) PT_PATCH(i, Pt, 0)
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();

    const uint32_t pt_base = lds_addr(Pt), bs_base = lds_addr(Bs);
    // the MFMAs of one k-group: fp32 = four 32x32x2 steps over the slot's four k values, bf16 = one 32x32x16 step over its eight
#define PT_MFMA1(A, B, mi, ni)                                                                     \
    if (BF16) acc[mi][ni] = SD_MFMA_BF16(2, __builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[mi][ni]); \
    else { _Pragma("unroll") for (int k = 0; k < 4; ++k) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[k], B[k], acc[mi][ni], 0, 0, 0); }
#define PT_MFMA_GROUP(A0, A1, A2, A3, B0, B1)                                                      \
    if (BF16) {                                                                                    \
        PT_MFMA1(A0, B0, 0, 0) PT_MFMA1(A1, B0, 1, 0)                                              \
        if (MT == 4) { PT_MFMA1(A2, B0, MT - 2, 0) PT_MFMA1(A3, B0, MT - 1, 0) }                   \
        PT_MFMA1(A0, B1, 0, 1) PT_MFMA1(A1, B1, 1, 1)                                              \
        if (MT == 4) { PT_MFMA1(A2, B1, MT - 2, 1) PT_MFMA1(A3, B1, MT - 1, 1) }                   \
    } else {                                                                                       \
        _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                            \
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[k], B0[k], acc[0][0], 0, 0, 0);    \
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[k], B0[k], acc[1][0], 0, 0, 0);    \
            if (MT == 4) {                                                                         \
                acc[MT - 2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[k], B0[k], acc[MT - 2][0], 0, 0, 0); \
                acc[MT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(A3[k], B0[k], acc[MT - 1][0], 0, 0, 0); \
            }                                                                                      \
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A0[k], B1[k], acc[0][1], 0, 0, 0);    \
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A1[k], B1[k], acc[1][1], 0, 0, 0);    \
            if (MT == 4) {                                                                         \
                acc[MT - 2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A2[k], B1[k], acc[MT - 2][1], 0, 0, 0); \
                acc[MT - 1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(A3[k], B1[k], acc[MT - 1][1], 0, 0, 0); \
            }                                                                                      \
        }                                                                                          \
    }
    // one tap: issue (weights of tap t+2, this tap's share of the patch traffic), multiply, publish
#define PT_TAP(t)                                                                                  \
    {                                                                                              \
        int n_iss = PBW;                                                                           \
        if (!(BF16 || SD_PATCH_READS_FIRST_F32)) {                                                 \
            PT_ISSUE_B(cc + ((t) + 2) / 9, ((t) + 2) % 9, ((t) + 2) % 3)                               \
        if (rolling) {                                                                             \
            /* taps 0..3: rows 2, 3 of this chunk; taps 4..7: rows 0, 1 of the next one; even tap: slot 0, odd: slots 1 (+2) */ \
            if ((t) < 8) {                                                                         \
                const int row = (t) < 4 ? 2 + (t) / 2 : ((t) - 4) / 2;                             \
                const int cch = (t) < 4 ? cc : cc + 1;                                             \
                if (cch < nchunks) {                                                               \
                    if (((t) & 1) == 0) { PT_PATCH(3 * row, Pt, cch * KC) ++n_iss; }               \
                    else {                                                                         \
                        if (PT_OWN(3 * row + 1)) { PT_PATCH(3 * row + 1, Pt, cch * KC) ++n_iss; }  \
                        if (PT_OWN(3 * row + 2)) { PT_PATCH(3 * row + 2, Pt, cch * KC) ++n_iss; }  \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        } else if ((t) < 7 && cc + 1 < nchunks && PT_OWN(t)) {                                     \
            PT_PATCH(t, pt_nxt, (cc + 1) * KC) ++n_iss;                                            \
        }                                                                                          \
        }                                                                                          \
        const int tapoff = ((t) / 3) * PW + ((t) % 3);      /* data-gradient: same walk, weight taps reversed */ \
        uint32_t aa[MT];                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) {                                        \
            const int pp = bpp[mi] + tapoff;                                                       \
            aa[mi] = pt_cur + (uint32_t)pp * 64u + (uint32_t)((fh ^ ((pp >> 2) & 3)) << 4);        \
        }                                                                                          \
        const uint32_t bb = bs_base + ((t) % 3) * (B_ST * 4) + b_k0;                               \
        f32x4 a00 = lds_read128_async<0>(aa[0]), a01 = lds_read128_async<0>(aa[1]);                \
        f32x4 a02 = a00, a03 = a00;                                                                \
        if (MT == 4) { a02 = lds_read128_async<0>(aa[MT - 2]); a03 = lds_read128_async<0>(aa[MT - 1]); } \
        f32x4 b00 = lds_read128_async<0>(bb), b01 = lds_read128_async<TSTR>(bb);                   \
        f32x4 a10 = lds_read128_async<0>(aa[0] ^ 32u), a11 = lds_read128_async<0>(aa[1] ^ 32u);    \
        f32x4 a12 = a10, a13 = a10;                                                                \
        if (MT == 4) { a12 = lds_read128_async<0>(aa[MT - 2] ^ 32u); a13 = lds_read128_async<0>(aa[MT - 1] ^ 32u); } \
        f32x4 b10 = lds_read128_async<0>(bb ^ 32u), b11 = lds_read128_async<TSTR>(bb ^ 32u);       \
        /* bf16: a wave gets one LDS-DMA instruction out per ~66 cycles and waits at it in order, and a bf16 tap is only 512 MFMA \
           cycles long -- the fragment reads go first (fp32 taps are 8x longer: the order does not matter there) */ \
        if (BF16 || SD_PATCH_READS_FIRST_F32) {                                                    \
            PT_ISSUE_B(cc + ((t) + 2) / 9, ((t) + 2) % 9, ((t) + 2) % 3)                               \
        if (rolling) {                                                                             \
            /* taps 0..3: rows 2, 3 of this chunk; taps 4..7: rows 0, 1 of the next one; even tap: slot 0, odd: slots 1 (+2) */ \
            if ((t) < 8) {                                                                         \
                const int row = (t) < 4 ? 2 + (t) / 2 : ((t) - 4) / 2;                             \
                const int cch = (t) < 4 ? cc : cc + 1;                                             \
                if (cch < nchunks) {                                                               \
                    if (((t) & 1) == 0) { PT_PATCH(3 * row, Pt, cch * KC) ++n_iss; }               \
                    else {                                                                         \
                        if (PT_OWN(3 * row + 1)) { PT_PATCH(3 * row + 1, Pt, cch * KC) ++n_iss; }  \
                        if (PT_OWN(3 * row + 2)) { PT_PATCH(3 * row + 2, Pt, cch * KC) ++n_iss; }  \
                    }                                                                              \
                }                                                                                  \
            }                                                                                      \
        } else if ((t) < 7 && cc + 1 < nchunks && PT_OWN(t)) {                                     \
            PT_PATCH(t, pt_nxt, (cc + 1) * KC) ++n_iss;                                            \
        }                                                                                          \
        }                                                                                          \
        if (MT == 4) { SD_LDS_WAIT6(6, a00, a01, a02, a03, b00, b01); } else { SD_LDS_WAIT4(4, a00, a01, b00, b01); } \
        PT_MFMA_GROUP(a00, a01, a02, a03, b00, b01)                                                \
        if (MT == 4) { SD_LDS_WAIT6(0, a10, a11, a12, a13, b10, b11); } else { SD_LDS_WAIT4(0, a10, a11, b10, b11); } \
        PT_MFMA_GROUP(a10, a11, a12, a13, b10, b11)                                                \
        /* everything this wave issued BEFORE this tap has landed; its LDS reads are done (see k_conv_igemm_big) */ \
        if (n_iss == PBW) wait_vmcnt_and_lds<PBW>();                                               \
        else if (n_iss == PBW + 1) wait_vmcnt_and_lds<PBW + 1>();                                  \
        else wait_vmcnt_and_lds<PBW + 2>();                                                        \
        __builtin_amdgcn_s_barrier();                                                              \
    }
    for (int cc = 0; cc < nchunks; ++cc) {
        const uint32_t pt_cur = pt_base + (rolling ? 0u : (uint32_t)(cc & 1) * (PT_STAGE_FLOATS * 4));
        float* const pt_nxt = Pt + (rolling ? 0 : ((cc + 1) & 1) * PT_STAGE_FLOATS);
        PT_TAP(0) PT_TAP(1) PT_TAP(2) PT_TAP(3) PT_TAP(4) PT_TAP(5) PT_TAP(6) PT_TAP(7) PT_TAP(8)
    }
    wait_vmcnt<0>();
#undef PT_TAP
#undef PT_MFMA_GROUP
#undef PT_MFMA1
#undef PT_ISSUE_B
#undef PT_OWN
#undef PT_PATCH
#undef PT_PB
    // 16-byte epilogue through the idle patch / weight buffers (BN = 128: 16 KB per wave); the barrier makes sure no other
    // wave's (past-the-end) DMA is still landing in them
    __syncthreads();
    float* T0 = BN == 128 ? (wave < 3 ? Pt + wave * 4096 : Bs) : Pt + wave * 2048;      // BN = 64: 8 KB per wave, 32-row passes
    tile_epilogue<BN, WM, WN, MT, NTW, true, BF16, true>(
        p, acc, [&](int row) { return m0 + row; },
        [&](int, int m) {      // half-size residual map (not on this kernel's hot uses: 3x3 / stride 1 layers join full-size residuals)
            const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            return (p.res_up2 == 2 && ((ox | oy) & 1)) ? -1 : (b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
        },
        tid, wave, fr, fh, wm0, wn0, n0, tile_m, T0, BN == 128 ? T0 + 2048 : nullptr);
}

template <int BN, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_patch(ConvArgs p) { conv3x3_patch_body<BN, BF16, false>(p); }
template <int BN, bool BF16 = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_patch_roll(ConvArgs p) { conv3x3_patch_body<BN, BF16, true>(p); }

// ---------------------------------------------------------------------------------------------
// bf16 3x3 / stride 1 / pad 1 convolution, TWO-GROUP form of the patch-staging kernel (forward and flipped-tap data-gradient).
// Why: a bf16 tap is 16 MFMAs of 32 cycles per wave -- 8x shorter than the fp32 tap -- so in k_conv3x3_patch<128, true> the
// per-tap barrier, the fragment reads and above all the 2-tap prefetch distance of the weight ring (shorter than one L2 round trip)
// are exposed: 31 % of the MFMA peak.  Here ONE 512-thread block per CU holds two 256-pixel sub-tiles that share the weights:
//   group g = wave / 4 owns sub-tile g (its own double-buffered patch); its four waves take 64 pixels x all 128 channels each;
//   the weight ring has 7 stages of one tap (8 KB), tap t + 5 is issued during tap t (all 8 waves, one 1 KB piece each):
//     staged weight bytes per MFMA are half of the 256-pixel tile's and a piece has > 5 tap times to land;
//   the groups run half a tap apart (group 1 does one extra s_barrier first, group 0 one last): every tap is
//     LOAD [12 ds_read_b128, then the tap's LDS-DMA, counted vmcnt] - s_barrier - MFMA [32 x 16x16x32 (round 4; 16 x 32x32x16 before), raised priority] - s_barrier
//     so that one group's LOAD always runs under the other group's MFMAs on the same SIMDs (2 waves per SIMD).
//   Hazards (t = tap, interval = time between two barriers; group 0 LOADs tap t in interval 2t, group 1 in 2t + 1):
//     RAW  a piece of tap t+1 is covered by its issuer's vmcnt wait in LOAD(t), which ends with a barrier every reader passes
//          before its LOAD(t+1);
//     WAR  stage (t+5) % 7 = stage of tap t-2, whose last reads (group 1, interval 2t-3) were waited for (lgkmcnt) in interval
//          2t-2, two barriers before group 0 issues in interval 2t.  The next chunk's patch is written by its own group only.
//   vmcnt: per tap a wave issues p(tap) patch pieces (2, 2, 2, 1, 0 ... of the NEXT chunk; non-existent pieces re-load piece
//          wave % 4, past the last chunk the zero line: the count stays fixed) and then 1 weight piece;
//          LOAD(t) needs the weight piece issued last in LOAD(t-4): N(t) = 4 + p(t) + p(t-1) + p(t-2) + p(t-3).
//   LDS: 4 x 25 KB patches + 7 x 8 KB weights = 156 KB (dynamic) + 1 KB statistics; epilogue scratch = the same array.
// ---------------------------------------------------------------------------------------------
// (packed bf16 helpers rs_pack2 / rs_relu2: sd_conv_rows.h)
// Lean epilogue of the two-group bf16 kernel for everything but a half-size residual map (those launches keep tile_epilogue).  The
// generic epilogue costs a wave ~1100 vector instructions per tile (per-row validity branches, 64-bit index chains that also serve the
// half-size residual, scalar fp32 math): 12-14 k cycles with the eight waves of the block in it at once, a fifth of a layer2 launch.
// Here: the same trip of the accumulators through the wave's two 32-row LDS regions (64 rows per pass), then per 16-byte run
// v_pk_fma_f32 for the affine, the residual added as shifted halves, one v_cvt_pk_bf16_f32 per pair and the ReLU as a signed 16-bit
// integer max on the packed pairs (rounding is monotonic: max(round(x), 0) = round(max(x, 0))).  Bit-identical to tile_epilogue
// (same fp32 operations per element in the same order).
typedef float pp_f32x2 __attribute__((ext_vector_type(2)));
// acc[mi][ni] = 16 x 16 tile (mi: 16 pixels, ni: 16 channels) of v_mfma_f32_16x16x32_bf16: lane (r16 = lane & 15, kq = lane >> 4) holds
// channel column r16, pixel rows kq * 4 + e.  Wave tile: 64 pixels x all 128 channels; a pass = one 64-channel half of it.
// HEAD (inference, network.py:22-29 fused behind network.py:17-18): the rounded bf16 rows of a pass are written back IN PLACE over the
// fp32 scratch rows they were read from (a wave's LDS operations execute in order: every lane of a row has read its 16 bytes before the
// row's 8-byte pieces land) at a position that makes the 16 x 16 x 32 operand reads conflict-free -- logical 16-byte slot s of row r at
// physical slot (s & 1) * 8 + ((r + 2 (s >> 1)) & 7) of the row's 256 bytes -- and multiplied by the head weights (hi + lo bf16 halves of
// the fp32 weights, fp32 accumulation: x * hi + x * lo equals the fp32 product to fp32 rounding because x is bf16 already).
template <bool HEAD>
__device__ __forceinline__ void pp_epilogue_bf16(const ConvArgs& p, f32x4 (&acc)[4][8], int mbase, int TWl, int tid, int wave, int n0, int tile_m,
                                                 float* T0, float* T1) {
    constexpr int BN = 128;
    const int lane = tid & 63, r16 = lane & 15, kq = lane >> 4;
    const int wml = (wave & 3) * 64;                                          // row origin of the wave tile inside its group's sub-tile
    const bool fwd_stat = p.stat && !p.bn_x;
    float sv[8], qv[8];
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
        sv[ni] = qv[ni] = 0.f;
        if (fwd_stat) {
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int e = 0; e < 4; ++e) { const float a = bf2f(f2bf(acc[mi][ni][e])); sv[ni] += a; qv[ni] += a * a; }
        }
    }
    __shared__ float pp_statred[2][BN];
    const int c4 = (lane & 15) * 4;
    const bool affine = p.scale || p.shift;
    const int TWm = (1 << TWl) - 1, rsub = lane >> 4;
    f32x4 hacc[4][2];
    if constexpr (HEAD) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) { hacc[mi][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[mi][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    }
    const bool head2 = HEAD && p.head_co > 16;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int n = n0 + h * 64 + c4;
        // head weights of this pass's two k-steps (channels 64 h + 32 ks + 8 kq ..): lane = (output channel r16 (+ 16), k group kq)
        bf16x8 hw[2][2][2];                                     // [k-step][output tile][hi / lo]
        if constexpr (HEAD) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int no = 0; no < 2; ++no)
#pragma unroll
                    for (int hl = 0; hl < 2; ++hl)
                        hw[ks][no][hl] = *reinterpret_cast<const bf16x8*>(p.head_w + ((hl * 32 + (head2 ? no : 0) * 16 + r16) * 128 + h * 64 + ks * 32 + kq * 8));
        }
        pp_f32x2 sc01 = {1.f, 1.f}, sc23 = {1.f, 1.f}, sh01 = {0.f, 0.f}, sh23 = {0.f, 0.f};
        if (p.scale) { const float4 t = *reinterpret_cast<const float4*>(p.scale + n); sc01 = pp_f32x2{t.x, t.y}; sc23 = pp_f32x2{t.z, t.w}; }
        if (p.shift) { const float4 t = *reinterpret_cast<const float4*>(p.shift + n); sh01 = pp_f32x2{t.x, t.y}; sh23 = pp_f32x2{t.z, t.w}; }
        uint16_t* const yb = reinterpret_cast<uint16_t*>(p.y) + n;
        const uint16_t* const rb = reinterpret_cast<const uint16_t*>(p.res) + n;
        // the residual runs of the pass are requested before its trip through LDS
        uint2 rr[16];
        if (p.res) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int ml = wml + it * 4 + rsub;
                const int m = mbase + (ml >> TWl) * p.Wo + (ml & TWm);
                rr[it] = *reinterpret_cast<const uint2*>(rb + (int64_t)m * p.Nn);
            }
        }
        // 64 pixel rows x 64 channels of the pass -> the wave's two 32-row x 64-float regions (two lane groups share a bank set: free
        // on ds_write_b32)
#pragma unroll
        for (int nj = 0; nj < 4; ++nj)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = kq * 4 + e;
                T0[r * 64 + nj * 16 + r16] = acc[0][4 * h + nj][e];
                T0[(r + 16) * 64 + nj * 16 + r16] = acc[1][4 * h + nj][e];
                T1[r * 64 + nj * 16 + r16] = acc[2][4 * h + nj][e];
                T1[(r + 16) * 64 + nj * 16 + r16] = acc[3][4 * h + nj][e];
            }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int rl = it * 4 + rsub;                                     // row of the wave tile: it < 8 -> T0, else T1
            const int ml = wml + rl;
            const int m = mbase + (ml >> TWl) * p.Wo + (ml & TWm);
            const f32x4 v = *reinterpret_cast<const f32x4*>((it < 8 ? T0 : T1) + (rl & 31) * 64 + c4);
            pp_f32x2 v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
            if (affine) { v01 = v01 * sc01 + sh01; v23 = v23 * sc23 + sh23; }
            if (p.res) {
                v01 += pp_f32x2{__uint_as_float(rr[it].x << 16), __uint_as_float(rr[it].x & 0xffff0000u)};
                v23 += pp_f32x2{__uint_as_float(rr[it].y << 16), __uint_as_float(rr[it].y & 0xffff0000u)};
            }
            uint2 pk;
            pk.x = rs_pack2(v01[0], v01[1]); pk.y = rs_pack2(v23[0], v23[1]);
            if (p.relu) { pk.x = rs_relu2(pk.x); pk.y = rs_relu2(pk.y); }
            if constexpr (HEAD) {
                const int rr = rl & 31, i = lane & 15, sl = i >> 1;            // piece i = channels 4 i .. 4 i + 3 of the pass: logical slot i / 2
                const int ps = (sl & 1) * 8 + ((rr + 2 * (sl >> 1)) & 7);
                *reinterpret_cast<uint2*>(reinterpret_cast<char*>(it < 8 ? T0 : T1) + rr * 256 + ps * 16 + (i & 1) * 8) = pk;
            } else {
                *reinterpret_cast<uint2*>(yb + (int64_t)m * p.Nn) = pk;
            }
        }
        if constexpr (HEAD) {
            // 64 pixels x 64 channels of bf16 rows -> A operands (pixel r16 of tile mi, k group kq), two k-steps
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int rr = (mi & 1) * 16 + r16;
                const char* const tb = reinterpret_cast<const char*>(mi < 2 ? T0 : T1) + rr * 256;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int sl = ks * 4 + kq;
                    const int ps = (sl & 1) * 8 + ((rr + 2 * (sl >> 1)) & 7);
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(tb + ps * 16);
                    hacc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hw[ks][0][0], hacc[mi][0], 0, 0, 0);
                    hacc[mi][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hw[ks][0][1], hacc[mi][0], 0, 0, 0);
                    if (head2) {
                        hacc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hw[ks][1][0], hacc[mi][1], 0, 0, 0);
                        hacc[mi][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, hw[ks][1][1], hacc[mi][1], 0, 0, 0);
                    }
                }
            }
        }
    }
    if constexpr (HEAD) {
        // D: column = output channel r16 (+ 16), rows = pixels 4 kq + e of tile mi: four consecutive pixels of one map row -> one 16-byte
        // store into the channel's NCHW plane
        const int HW = p.Ho * p.Wo;
#pragma unroll
        for (int no = 0; no < 2; ++no) {
            const int co = no * 16 + r16;
            if (co >= p.head_co) continue;
            const float bv = p.head_b[co];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int ml = wml + mi * 16 + kq * 4;
                const int m = mbase + (ml >> TWl) * p.Wo + (ml & TWm);
                const int b = m / HW, pix = m - b * HW;
                f32x4 o = hacc[mi][no];
                o[0] += bv; o[1] += bv; o[2] += bv; o[3] += bv;
                *reinterpret_cast<f32x4*>(p.head_y + ((int64_t)b * p.head_co + co) * HW + pix) = o;
            }
        }
    }
    if (fwd_stat) {
        // lane groups (fixed order), then waves 1 .. 7 into LDS one after the other (fixed order), wave 0 finishes: the 512-pixel tile is
        // one statistics row
#pragma unroll
        for (int ni = 0; ni < 8; ++ni) {
            sv[ni] += __shfl_xor(sv[ni], 16); qv[ni] += __shfl_xor(qv[ni], 16);
            sv[ni] += __shfl_xor(sv[ni], 32); qv[ni] += __shfl_xor(qv[ni], 32);
        }
        if (tid < 2 * BN) pp_statred[tid / BN][tid % BN] = 0.f;
        __syncthreads();
        for (int wr = 1; wr < 8; ++wr) {
            if (wave == wr && kq == 0) {
#pragma unroll
                for (int ni = 0; ni < 8; ++ni) { pp_statred[0][ni * 16 + r16] += sv[ni]; pp_statred[1][ni * 16 + r16] += qv[ni]; }
            }
            __syncthreads();
        }
        if (wave == 0 && kq == 0) {
            float* dst = p.stat + (int64_t)tile_m * 2 * p.Nn + n0;
#pragma unroll
            for (int ni = 0; ni < 8; ++ni) {
                const int c = ni * 16 + r16;
                dst[c] = sv[ni] + pp_statred[0][c]; dst[p.Nn + c] = qv[ni] + pp_statred[1][c];
            }
        }
    }
}

#ifdef SD_PP_TRACE
// timing experiment (make SUFFIX=_pptrace EXTRA=-DSD_PP_TRACE): shader-clock time of the phases of every tap, summed per wave of block 0
__device__ unsigned long long g_pp_trace[8][8];
__device__ unsigned long long g_pp_tl[4096][4];      // per block (wave 0): 100 MHz times of start, loop begin, loop end, epilogue end
#define PP_T(v) const unsigned long long v = __builtin_readcyclecounter();
#define PP_ACC(k, a, b) tr[k] += (b) - (a);
#else
#define PP_T(v)
#define PP_ACC(k, a, b)
#endif
#ifndef SD_PP_ABL
#define SD_PP_ABL 0          // timing experiments (WRONG RESULTS): 1 no fragment reads, 2 no DMA in the loop, 3 no MFMA, 4 no s_setprio, 5 no vmcnt wait, 6 DMA from the zero line only, 7 DMA only (no reads, no MFMA)
#endif
constexpr int PP_NST = 7, PP_D = 5;
constexpr int PP_B_FLOATS = 128 * 16;                                            // one tap of 128 channels x 32 k (bf16) = 8 KB
constexpr int PP_LDS_FLOATS = 4 * PT_STAGE_FLOATS + PP_NST * PP_B_FLOATS;        // 39936 floats = 159744 B
constexpr int PP_BM = 512;

__device__ __forceinline__ f32x4 pp_fake_read(uint32_t addr) { f32x4 v; asm volatile("v_mov_b32 %0, %1" : "=v"(v[0]) : "v"(addr)); v[1] = v[2] = v[3] = v[0]; return v; }
template <bool HEAD>
__device__ __forceinline__ void conv3x3_bf16_pp_body(const ConvArgs& p) {
    using T = uint16_t;
    constexpr int BN = 128, EPS = 8, KC = 32, MT = 4, NTW = 8, NPP = 7;     // wave tile: 4 (pixels) x 8 (channels) MFMA tiles of 16 x 16
    extern __shared__ __attribute__((aligned(16))) float pp_lds[];
    PP_T(tr_start)
#ifdef SD_PP_TRACE
    const unsigned long long tl_start = __builtin_amdgcn_s_memrealtime();
#endif
    const T* const px_ = reinterpret_cast<const T*>(p.x);
    const T* const pw_ = reinterpret_cast<const T*>(p.w);
    const T* const zero_ = reinterpret_cast<const T*>(g_zero_line);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, wl = wave & 3;
    const int n_tiles = p.Nn / BN;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = tile / n_tiles;
    const int n0 = (tile % n_tiles) * BN;
    // this group's 256-pixel sub-tile: TH rows of TW pixels -- whole map rows, or (maps of 128 pixels and wider) a 64-pixel column
    // strip of four rows, so that its patch still fits a double-buffered 25 KB stage; the two groups of a block take neighbours
    const int TWl = p.pt_tw_log2, TW = 1 << TWl, PW = p.pt_pw, TH = BMB >> TWl;
    const int q = tile_m * 2 + grp, spi = (p.Ho / TH) << p.pt_strip_log2;      // sub-tile, sub-tiles per image
    const int bimg = q / spi, rq = q - bimg * spi;
    const int y0 = (rq >> p.pt_strip_log2) * TH, x0 = (rq & ((1 << p.pt_strip_log2) - 1)) << TWl;
    const int mbase = (bimg * p.Ho + y0) * p.Wo + x0;           // output pixel of the sub-tile's first row / column
    const int nchunks = p.Ck / KC;
    float* const Pt = pp_lds + grp * (2 * PT_STAGE_FLOATS);
    float* const Bs = pp_lds + 4 * PT_STAGE_FLOATS;

    // ---- patch pieces of this wave: piece j = wl + 4 i (i < 7); a piece past the patch re-loads piece wl
    const int prow = lane >> 2, pslot = lane & 3;
    const T* pbase[NPP];
    int pdst[NPP];
    unsigned okmask = 0;
#pragma unroll
    for (int i = 0; i < NPP; ++i) {
        const int j = (wl + 4 * i < p.pt_pieces) ? wl + 4 * i : wl;
        const int pp = j * 16 + prow;
        const int py = pp / PW, pxx = pp - py * PW;
        const int iy = y0 - 1 + py, ix = x0 + pxx - 1;
        const bool ok = py < TH + 2 && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;
        const int qe = (pslot ^ ((pp >> 1) & 2)) * EPS;                  // slot swizzle of the 16 x 16 x 32 operand layout (see below)
        pbase[i] = ok ? px_ + (((int64_t)bimg * p.Hi + iy) * p.Wi + ix) * p.Ck + qe : zero_ + qe;
        okmask |= (ok ? 1u : 0u) << i;
        pdst[i] = j * 256;
    }
    const int wk = 9 * p.Ck;
    const int qeb = (pslot ^ ((prow >> 1) & 2)) * EPS;
    const T* const bbase = pw_ + (int64_t)(n0 + wave * 16 + prow) * wk + qeb;
    const T* const zsrc = zero_ + qeb;

    // patch piece i of chunk cch -> stage buffer dst (past the last chunk: the zero line)
#define PP_PATCH(i, dst, cch) if (SD_PP_ABL == 6 && (cch) != 0) { lds_dma16(zsrc, (dst) + pdst[i]); } else if (SD_PP_ABL != 2 || (cch) == 0) { lds_dma16((((okmask >> (i)) & 1u) && (cch) < nchunks) ? pbase[i] + (cch) * KC : (((okmask >> (i)) & 1u) ? zero_ + (lane & 3) * EPS : pbase[i]), (dst) + pdst[i]); }
    // weights of (chunk cci, tap ti) -> ring stage st
#define PP_ISSUE_B(cci, ti, st)                                                                    \
    {                                                                                              \
        float* bd = Bs + (st) * PP_B_FLOATS + wave * 256;                                          \
        if (SD_PP_ABL == 2 && pp_in_loop) {}                                                        \
        else if (SD_PP_ABL == 6 && pp_in_loop) lds_dma16(zsrc, bd);                                \
        else if ((cci) < nchunks) lds_dma16(bbase + ((p.pt_flip ? 8 - (ti) : (ti)) * p.Ck + (cci) * KC), bd); \
        else lds_dma16(zsrc, bd);                                                                  \
    }

    // ---- MFMA side: inside the group the four waves take 64 pixels x all 128 channels each = 4 x 8 tiles of v_mfma_f32_16x16x32_bf16.
    // Shape: the chip holds a higher clock on the 16x16x32 form than on 32x32x16 at the same flops and LDS traffic (random operands;
    // tools/micro/mfma_bf16_shape.hip, and this kernel with every 32x32x16 replaced by two 16x16x32 on the same registers: forward -4.7 %).
    // Operand layout: lane = (r16 = lane & 15: pixel row / channel column of the tile, kq = lane >> 4: 8-wide k group) -- ONE ds_read_b128
    // per operand tile covers the chunk's K = 32.  Conflict-free slot swizzle for that lane map at every tap offset: slot ^ ((row >> 1) & 2)
    // (brute-forced over all patch origins against the four ds_read_b128 lane groups).  The pixel side has the expensive addresses (tap
    // offset + swizzle per tile and tap), the weight side one base + immediates: hence 4 pixel tiles x 8 channel tiles per wave.
    const int wm0 = wl * 64;
    const int r16 = lane & 15, kq = lane >> 4;
    uint32_t bppk[MT];                                                      // byte offset of (tile pixel, k group) in an unswizzled patch
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int ml = wm0 + mi * 16 + r16;
        bppk[mi] = (uint32_t)((ml >> TWl) * PW + (ml & (TW - 1))) * 64u + (uint32_t)(kq << 4);
    }
    const uint32_t b_k0 = (uint32_t)(r16 * BKB * 4) + (uint32_t)((kq ^ ((r16 >> 1) & 2)) << 4);
    constexpr int TSTR = 16 * BKB * 4;                                      // 16 weight rows = one channel tile
    f32x4 acc[MT][NTW];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int ni = 0; ni < NTW; ++ni) acc[mi][ni] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- prologue: the patch of chunk 0 and the weights of taps 0 .. PP_D-1
    bool pp_in_loop = false; (void)pp_in_loop;
#pragma unroll
    for (int i = 0; i < NPP; ++i) PP_PATCH(i, Pt, 0)
#pragma unroll
    for (int t = 0; t < PP_D; ++t) PP_ISSUE_B(0, t, t)
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();               // group 1 runs half a tap behind

    const uint32_t pt_base = lds_addr(Pt), bs_base = lds_addr(Bs);
    int rp = 0, ip = PP_D;                                     // ring stage of the current tap / of the tap being issued
    pp_in_loop = true;
#if SD_PP_ABL == 1 || SD_PP_ABL == 7
#define PP_RD(OFF, addr) pp_fake_read(addr)
#else
#define PP_RD(OFF, addr) lds_read128_async<OFF>(addr)
#endif
#if SD_PP_ABL == 3 || SD_PP_ABL == 7
#define PP_MFMA(A, B, mi, ni) asm volatile("" :: "v"(A), "v"(B));
#else
#define PP_MFMA(A, B, mi, ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, B), acc[mi][ni], 0, 0, 0);
#endif
#define PP_TAP(t, NW)                                                                              \
    {                                                                                              \
        /* LOAD */                                                                                 \
        PP_T(t0_)                                                                                  \
        const int tapoff = ((t) / 3) * PW + ((t) % 3);                                             \
        /* four VALU per address, recomputed every tap: hoisted out of the chunk loop the 36 (tile, tap) addresses do not fit the    \
           register file beside 128 accumulators + 48 fragment registers (the empty asm keeps the compiler from hoisting them) */    \
        uint32_t aa[MT];                                                                           \
        _Pragma("unroll") for (int mi = 0; mi < MT; ++mi) {                                        \
            asm volatile("" : "+v"(bppk[mi]));                                                     \
            const uint32_t x0 = bppk[mi] + (uint32_t)tapoff * 64u;      /* = pixel * 64 + kq * 16: bit 8 = bit 2 of the patch pixel */ \
            aa[mi] = (x0 ^ ((x0 >> 3) & 32u)) + pt_cur;                                            \
        }                                                                                          \
        const uint32_t bb = bs_base + (uint32_t)rp * (PP_B_FLOATS * 4) + b_k0;                     \
        f32x4 a0 = PP_RD(0, aa[0]), a1 = PP_RD(0, aa[1]), a2 = PP_RD(0, aa[2]), a3 = PP_RD(0, aa[3]); \
        f32x4 b0 = PP_RD(0, bb), b1 = PP_RD(TSTR, bb), b2 = PP_RD(2 * TSTR, bb), b3 = PP_RD(3 * TSTR, bb); \
        f32x4 b4 = PP_RD(4 * TSTR, bb), b5 = PP_RD(5 * TSTR, bb), b6 = PP_RD(6 * TSTR, bb), b7 = PP_RD(7 * TSTR, bb); \
        /* the DMA goes out AFTER the fragment reads: the CU takes one 1 KB LDS-DMA instruction per ~32 cycles and a wave waits at \
           its DMA instruction until the queue takes it -- in front of the reads that wait was on the interval's critical path */ \
        if ((t) == 0) { PP_PATCH(0, pt_nxt, cc + 1) PP_PATCH(1, pt_nxt, cc + 1) }                  \
        if ((t) == 1) { PP_PATCH(2, pt_nxt, cc + 1) PP_PATCH(3, pt_nxt, cc + 1) }                  \
        if ((t) == 2) { PP_PATCH(4, pt_nxt, cc + 1) PP_PATCH(5, pt_nxt, cc + 1) }                  \
        if ((t) == 3) { PP_PATCH(6, pt_nxt, cc + 1) }                                              \
        PP_ISSUE_B(cc + ((t) + PP_D) / 9, ((t) + PP_D) % 9, ip)                                    \
        rp = rp + 1 == PP_NST ? 0 : rp + 1;                                                        \
        ip = ip + 1 == PP_NST ? 0 : ip + 1;                                                        \
        if (SD_PP_ABL != 5) wait_vmcnt<NW>();                                                      \
        __builtin_amdgcn_s_barrier();                                                              \
        /* MFMA */                                                                                 \
        SD_LDS_WAIT8(4, a0, a1, a2, a3, b0, b1, b2, b3);                                           \
        PP_T(t1_)                                                                                  \
        if (SD_PP_ABL != 4) __builtin_amdgcn_s_setprio(1);                                         \
        PP_MFMA(a0, b0, 0, 0) PP_MFMA(a1, b0, 1, 0) PP_MFMA(a2, b0, 2, 0) PP_MFMA(a3, b0, 3, 0)    \
        PP_MFMA(a0, b1, 0, 1) PP_MFMA(a1, b1, 1, 1) PP_MFMA(a2, b1, 2, 1) PP_MFMA(a3, b1, 3, 1)    \
        PP_MFMA(a0, b2, 0, 2) PP_MFMA(a1, b2, 1, 2) PP_MFMA(a2, b2, 2, 2) PP_MFMA(a3, b2, 3, 2)    \
        PP_MFMA(a0, b3, 0, 3) PP_MFMA(a1, b3, 1, 3) PP_MFMA(a2, b3, 2, 3) PP_MFMA(a3, b3, 3, 3)    \
        SD_LDS_WAIT4(0, b4, b5, b6, b7);                                                           \
        PP_MFMA(a0, b4, 0, 4) PP_MFMA(a1, b4, 1, 4) PP_MFMA(a2, b4, 2, 4) PP_MFMA(a3, b4, 3, 4)    \
        PP_MFMA(a0, b5, 0, 5) PP_MFMA(a1, b5, 1, 5) PP_MFMA(a2, b5, 2, 5) PP_MFMA(a3, b5, 3, 5)    \
        PP_MFMA(a0, b6, 0, 6) PP_MFMA(a1, b6, 1, 6) PP_MFMA(a2, b6, 2, 6) PP_MFMA(a3, b6, 3, 6)    \
        PP_MFMA(a0, b7, 0, 7) PP_MFMA(a1, b7, 1, 7) PP_MFMA(a2, b7, 2, 7) PP_MFMA(a3, b7, 3, 7)    \
        __builtin_amdgcn_s_setprio(0);                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        PP_T(t2_)                                                                                  \
        __builtin_amdgcn_s_barrier();                                                              \
        PP_T(t3_)                                                                                  \
        PP_ACC(0, t0_, t1_) PP_ACC(1, t1_, t2_) PP_ACC(2, t2_, t3_)                                \
    }
#ifdef SD_PP_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tr_begin = __builtin_readcyclecounter();
    const unsigned long long tr_rbegin = __builtin_amdgcn_s_memrealtime();
#endif
    for (int cc = 0; cc < nchunks; ++cc) {
        const uint32_t pt_cur = pt_base + (uint32_t)(cc & 1) * (PT_STAGE_FLOATS * 4);
        float* const pt_nxt = Pt + ((cc + 1) & 1) * PT_STAGE_FLOATS;
        PP_TAP(0, 6) PP_TAP(1, 8) PP_TAP(2, 10) PP_TAP(3, 11) PP_TAP(4, 9) PP_TAP(5, 7) PP_TAP(6, 5) PP_TAP(7, 4) PP_TAP(8, 4)
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    wait_vmcnt<0>();
#ifdef SD_PP_TRACE
    const unsigned long long tr_end = __builtin_readcyclecounter();
    tr[3] = tr_end - tr_begin; tr[4] = tr_begin - tr_start; tr[7] = __builtin_amdgcn_s_memrealtime() - tr_rbegin;
#endif
#undef PP_TAP
#undef PP_MFMA
#undef PP_RD
#undef PP_ISSUE_B
#undef PP_PATCH
    // 16-byte epilogue through the (now idle) LDS: 16 KB per wave; the barrier makes sure no past-the-end DMA is still landing.
    // The 512-pixel tile is one statistics row: wave rows 0, 1 = group 0, rows 2, 3 = group 1.
    __syncthreads();
    float* T0 = pp_lds + wave * 4096;
    pp_epilogue_bf16<HEAD>(p, acc, mbase, TWl, tid, wave, n0, tile_m, T0, T0 + 2048);        // (no half-size residual here: conv_pp_geometry)
#ifdef SD_PP_TRACE
    tr[5] = __builtin_readcyclecounter() - tr_end;
    tr[6] = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == 8 && lane == 0) { for (int k = 0; k < 8; ++k) g_pp_trace[wave][k] = tr[k]; }
    if (wave == 0 && lane == 0 && blockIdx.x < 4096) {
        g_pp_tl[blockIdx.x][0] = tl_start; g_pp_tl[blockIdx.x][1] = tr_rbegin; g_pp_tl[blockIdx.x][2] = tr_rbegin + tr[7]; g_pp_tl[blockIdx.x][3] = tr[6];
    }
#endif
}

__global__ __launch_bounds__(512, 1) void k_conv3x3_bf16_pp(ConvArgs p) { conv3x3_bf16_pp_body<false>(p); }
// the same convolution with the network's 1x1 head applied to its output tile in the epilogue (inference): the FPN output is never stored
__global__ __launch_bounds__(512, 1) void k_conv3x3_bf16_pp_head(ConvArgs p) { conv3x3_bf16_pp_body<true>(p); }

// ---------------------------------------------------------------------------------------------
// bf16 1x1 / stride 1 convolution onto 128 output channels (the FPN laterals: 64 -> 128 on the 128 x 128 map moves 470 MB for 17 GFLOP)
// as a STREAM: k_conv_igemm's 128 x 128 tile has a two-chunk loop there, all prologue and epilogue (172 us = 2.7 TB/s at bs = 64).
// Here every wave is its own pipeline with no LDS and no barrier:
//   weights : all 128 x CIN in registers as the A operands of v_mfma_f32_16x16x32_bf16 (8 channel tiles x CIN / 32 k-steps, loaded once)
//   pixels  : the B operand straight from global memory, 16 bytes per lane = 8 consecutive channels of the lane's pixel (the operand
//             is streamed once and shared with no other wave: no LDS round trip); the next tile's loads are issued before this tile's
//             epilogue, the residual's before the MFMAs
//   output  : D rows = channels, columns = pixels.  Tile t holds the channels 32 (t / 2) + 8 (row / 4) + 4 (t % 2) + row % 4, so that a
//             lane ends up with 8 CONSECUTIVE channels of one pixel per tile pair: bias + (upsampled) residual + ReLU in registers and
//             one 16-byte store, no transpose through LDS.
// Tiles of 16 P pixels, waves walk them with a grid stride.  Host side: launch_igemm (bf16, R = S = 1, unit stride, Nn = 128, Ck = 64 / 128,
// no scale, no statistics, res_up2 <= 1).
// ---------------------------------------------------------------------------------------------
template <int CIN, int P>
__global__ __launch_bounds__(256, 2) void k_conv1x1_stream_bf16(ConvArgs p, int ntiles) {
    constexpr int KS = CIN / 32;
    const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
    const uint16_t* const x = reinterpret_cast<const uint16_t*>(p.x);
    const uint16_t* const w = reinterpret_cast<const uint16_t*>(p.w);
    const uint16_t* const res = reinterpret_cast<const uint16_t*>(p.res);
    uint16_t* const y = reinterpret_cast<uint16_t*>(p.y);
    // A operands: tile t, row i = j -> channel 32 (t >> 1) + 8 (i >> 2) + 4 (t & 1) + (i & 3); this lane holds its k group g of every step
    bf16x8 wf[8][KS];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int c = 32 * (t >> 1) + 8 * (j >> 2) + 4 * (t & 1) + (j & 3);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) wf[t][ks] = *reinterpret_cast<const bf16x8*>(w + c * CIN + ks * 32 + g * 8);
    }
    const int nwaves = gridDim.x * 4;
    int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int HWo = p.Ho * p.Wo, Wh = p.Wo >> 1, HWh = (p.Ho >> 1) * Wh;
    bf16x8 xb[P][KS];
    if (tile < ntiles) {
#pragma unroll
        for (int pt = 0; pt < P; ++pt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xb[pt][ks] = *reinterpret_cast<const bf16x8*>(x + (int64_t)(tile * (16 * P) + pt * 16 + j) * CIN + ks * 32 + g * 8);
    }
    for (; tile < ntiles; tile += nwaves) {
        const int m0 = tile * (16 * P);
        // residual runs of this tile: 8 channels (32 u + 8 g ..) of the lane's pixel, or of its parent in the half-size map
        uint4 rr[P][4];
        if (res) {
#pragma unroll
            for (int pt = 0; pt < P; ++pt) {
                const int m = m0 + pt * 16 + j;
                int64_t rm = m;
                if (p.res_up2) {
                    const int b = m / HWo, r = m - b * HWo, oy = r / p.Wo, ox = r - oy * p.Wo;
                    rm = (int64_t)b * HWh + (oy >> 1) * Wh + (ox >> 1);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) rr[pt][u] = *reinterpret_cast<const uint4*>(res + rm * 128 + u * 32 + g * 8);
            }
        }
        f32x4 acc[P][8];
#pragma unroll
        for (int pt = 0; pt < P; ++pt)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[pt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int pt = 0; pt < P; ++pt) acc[pt][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[t][ks], xb[pt][ks], acc[pt][t], 0, 0, 0);
        // the next tile's pixels are requested before this tile's epilogue
        const int nxt = tile + nwaves;
        if (nxt < ntiles) {
#pragma unroll
            for (int pt = 0; pt < P; ++pt)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) xb[pt][ks] = *reinterpret_cast<const bf16x8*>(x + (int64_t)(nxt * (16 * P) + pt * 16 + j) * CIN + ks * 32 + g * 8);
        }
#pragma unroll
        for (int pt = 0; pt < P; ++pt) {
            const int64_t m = m0 + pt * 16 + j;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int cb = u * 32 + g * 8;
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = acc[pt][2 * u][e]; v[4 + e] = acc[pt][2 * u + 1][e]; }
                if (p.shift) {
                    const float4 s0 = *reinterpret_cast<const float4*>(p.shift + cb), s1 = *reinterpret_cast<const float4*>(p.shift + cb + 4);
                    v[0] += s0.x; v[1] += s0.y; v[2] += s0.z; v[3] += s0.w; v[4] += s1.x; v[5] += s1.y; v[6] += s1.z; v[7] += s1.w;
                }
                if (res) {
                    const uint4 r = rr[pt][u];
                    v[0] += __uint_as_float(r.x << 16); v[1] += __uint_as_float(r.x & 0xffff0000u);
                    v[2] += __uint_as_float(r.y << 16); v[3] += __uint_as_float(r.y & 0xffff0000u);
                    v[4] += __uint_as_float(r.z << 16); v[5] += __uint_as_float(r.z & 0xffff0000u);
                    v[6] += __uint_as_float(r.w << 16); v[7] += __uint_as_float(r.w & 0xffff0000u);
                }
                uint4 pk;
                pk.x = rs_pack2(v[0], v[1]); pk.y = rs_pack2(v[2], v[3]); pk.z = rs_pack2(v[4], v[5]); pk.w = rs_pack2(v[6], v[7]);
                if (p.relu) { pk.x = rs_relu2(pk.x); pk.y = rs_relu2(pk.y); pk.z = rs_relu2(pk.z); pk.w = rs_relu2(pk.w); }
                *reinterpret_cast<uint4*>(y + m * 128 + cb) = pk;
            }
        }
    }
}
// geometry / argument conditions of k_conv1x1_stream_bf16; `mode` as in launch_igemm.  sd_set_option("conv1x1_stream_min_pixels", n): output
// pixels from which the stream kernel replaces the tile kernel (tests: 32; off: 1 << 30).
static thread_local int g_conv1x1_stream_min_px = 32 * 2048;
static bool conv1x1_stream_geometry(const ConvArgs& a, int mode) {
    if (mode != 0 || a.R != 1 || a.S != 1 || a.mul != 1 || a.div != 1 || a.off != 0) return false;
    if (a.Nn != 128 || (a.Ck != 64 && a.Ck != 128) || a.Ho != a.Hi || a.Wo != a.Wi) return false;
    if (a.M % 32 || a.M < g_conv1x1_stream_min_px) return false;          // small maps: the tile kernel (split-K) fills the chip better
    return true;
}
static bool conv1x1_stream_args(const ConvArgs& a) {
    return !a.scale && !a.stat && !a.bn_x && a.splits <= 1 && a.res_up2 <= 1 && (!a.res_up2 || (a.Ho % 2 == 0 && a.Wo % 2 == 0));
}

// ---------------------------------------------------------------------------------------------
// bf16 3x3 / stride 1 / pad 1 convolution of the 64 -> 64 channel layers (layer1; forward and flipped-tap data-gradient) as a ROW
// STREAM with the weights in registers.  K is only 576 here: a 256-pixel tile is 18 taps long, so a tile kernel spends as long in its
// prologue / epilogue as in its loop (k_conv3x3_patch<64, true>: 130-170 us per launch = 450-590 TFLOP/s).  Here a persistent block
// (4 waves, one per SIMD) walks down a 128-pixel-wide strip of one image, one output row per step:
//   weights : every wave holds ALL 72 B fragments (9 taps x 4 k-steps x 2 column tiles = 288 registers, loaded once per block); no
//             weight traffic and no weight LDS reads in the loop
//   input   : ring of 5 input rows (136 pixels x 128 bytes, slot c of pixel x at c ^ ((x >> 1) & 7): conflict-free ds_read_b128) filled
//             by LDS-DMA three rows ahead; a wave multiplies 32 pixels x 64 channels per row: 36 A reads + 72 MFMAs (LDS traffic 25 %
//             of the array's rate -- with the weights in LDS a 32 x 64 wave tile would need 75 %)
//   output  : accumulators -> (scale, shift) -> wave-private LDS scratch; stored ONE ROW LATER as rows (16-byte stores) with the residual
//             (prefetched a row earlier), ReLU and the BatchNorm column sums of the rounded values, so that neither the stores nor the
//             residual loads are waited for; one s_barrier + vmcnt(0) per row (2304 MFMA cycles)
//   unit    : (image, strip, `rows` consecutive output rows); statistics: one partial row per unit.
// ---------------------------------------------------------------------------------------------
// (RowsArgs and the ring constants RS_PX / RS_ROW_BYTES / RS_NR: sd_conv_rows.h)
constexpr int RS_SCR = 72;                                                          // scratch row stride (floats)
constexpr int RS_LDS_BYTES = RS_NR * RS_ROW_BYTES + 4 * 32 * RS_SCR * 4 + 4 * 128 * 4;

// The MFMA of the row-stream kernel with explicit register classes: accumulators and the first RS_B_AGPR weight fragments live in AGPRs and
// are read from there (hipcc otherwise parks most of the 288 weight registers in AGPRs and copies four of them to VGPRs in front of EVERY
// MFMA: 288 v_accvgpr_read per row = half of the vector issue slots the MFMAs leave free).  256 AGPRs = 32 accumulator + 56 x 4 weights.
#if defined(SD_RS_ABL) && SD_RS_ABL == 3
#define SD_RS_NOWAIT 1          // timing experiment (WRONG RESULTS): the A fragments are not waited for
#else
#define SD_RS_NOWAIT 0
#endif
constexpr int RS_B_AGPR = 56;
template <bool B_IN_AGPR, bool ZERO>
__device__ __forceinline__ void rs_mfma(f32x16& acc, const f32x4& a, const bf16x8& b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (ZERO) {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
#else
    (void)acc; (void)a; (void)b;
#endif
}

__global__ __launch_bounds__(256) void k_conv3x3_c64_rows_bf16(RowsArgs p) {
    extern __shared__ __attribute__((aligned(16))) float rs_lds[];
    char* const ring = reinterpret_cast<char*>(rs_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* const scr = reinterpret_cast<float*>(ring + RS_NR * RS_ROW_BYTES) + wave * 32 * RS_SCR;
    float* const sred = reinterpret_cast<float*>(ring + RS_NR * RS_ROW_BYTES) + 4 * 32 * RS_SCR;      // [4][128]
    const int fr = lane & 31, fh = lane >> 5;
    const uint16_t* const zero_ = reinterpret_cast<const uint16_t*>(g_zero_line);

    // ---- every B fragment of the layer: lane (n = 32 ni + fr, k-half fh) holds w[n][tap][16 kc + 8 fh .. + 7]
    bf16x8 Bw[72];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int kc = 0; kc < 4; ++kc)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
                Bw[(t * 4 + kc) * 2 + ni] = *reinterpret_cast<const bf16x8*>(p.w + ((ni * 32 + fr) * 9 + (p.flip ? 8 - t : t)) * 64 + kc * 16 + fh * 8);
    float sc[2] = {1.f, 1.f}, sh[2] = {0.f, 0.f};
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        if (p.scale) sc[ni] = p.scale[ni * 32 + fr];
        if (p.shift) sh[ni] = p.shift[ni * 32 + fr];
    }
    // A fragment offsets inside a ring row: output pixel 32 wave + fr, tap column s -> ring pixel 32 wave + fr + s (ring pixel 0 = image column x0 - 1)
    uint32_t aoff[3][4];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int pxr = wave * 32 + fr + s;
#pragma unroll
        for (int kc = 0; kc < 4; ++kc) aoff[s][kc] = (uint32_t)pxr * 128u + (uint32_t)(((2 * kc + fh) ^ ((pxr >> 1) & 7)) << 4);
    }
    const uint32_t ring_base = lds_addr(ring);
    const int dpx = lane >> 3, dslot = lane & 7;          // LDS-DMA: lane -> (pixel within an 8-pixel piece, physical 16-byte slot)
    const int spx = lane >> 3, sc8 = lane & 7;            // row form of the output: lane -> (pixel within 8, 8-channel group)

    for (int unit = blockIdx.x; unit < p.nunits; unit += gridDim.x) {
        const int col = unit / p.units_per_col, yu = unit - col * p.units_per_col;
        const int b = col / p.segs, x0 = (col - b * p.segs) * 128;
        const int y0 = yu * p.rows, y1 = min(y0 + p.rows, p.H);
        const uint16_t* const img = p.x + (int64_t)b * p.H * p.W * 64;
        // input row iy -> ring slot (iy - y0 + 1) % 5; wave w loads pieces w, w + 4, ...
        auto issue_row = [&](int iy) {
            const int slot = (iy - y0 + 1) % RS_NR;
            float* const dst = reinterpret_cast<float*>(ring + slot * RS_ROW_BYTES);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int pc = wave + 4 * j;
                if (pc < 17) {
                    const int pxr = pc * 8 + dpx, ix = x0 - 1 + pxr;
                    const bool ok = pxr < 130 && (unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H;
                    const uint16_t* src = ok ? img + ((int64_t)iy * p.W + ix) * 64 + ((dslot ^ ((pxr >> 1) & 7)) << 3) : zero_ + (dslot << 3);
                    lds_dma16(src, dst + pc * 256);
                }
            }
        };
        float ssum[8], ssq[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { ssum[k] = 0.f; ssq[k] = 0.f; }
        uint4 resv[4] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        // the row form of output row yy: scratch (+ residual) -> ReLU -> bf16 -> statistics -> 16-byte stores
        auto store_row = [&](int yy) {
            const int64_t rowbase = (((int64_t)b * p.H + yy) * p.W + x0 + wave * 32) * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int pxl = spx + 8 * it;
                const float4 v0 = *reinterpret_cast<const float4*>(scr + pxl * RS_SCR + sc8 * 8);
                const float4 v1 = *reinterpret_cast<const float4*>(scr + pxl * RS_SCR + sc8 * 8 + 4);
                float f[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                if (p.res) {
                    const uint4 r = resv[it];
                    f[0] += bf2f((uint16_t)(r.x & 0xffff)); f[1] += bf2f((uint16_t)(r.x >> 16)); f[2] += bf2f((uint16_t)(r.y & 0xffff)); f[3] += bf2f((uint16_t)(r.y >> 16));
                    f[4] += bf2f((uint16_t)(r.z & 0xffff)); f[5] += bf2f((uint16_t)(r.z >> 16)); f[6] += bf2f((uint16_t)(r.w & 0xffff)); f[7] += bf2f((uint16_t)(r.w >> 16));
                }
                uint16_t h[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    if (p.relu) f[k] = fmaxf(f[k], 0.f);
                    h[k] = f2bf(f[k]);
                    const float a = bf2f(h[k]);
                    ssum[k] += a; ssq[k] += a * a;
                }
                uint4 o;
                o.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); o.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
                o.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16); o.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
                *reinterpret_cast<uint4*>(p.y + rowbase + pxl * 64 + sc8 * 8) = o;
            }
        };
        for (int iy = y0 - 1; iy <= min(y0 + 2, y1); ++iy) issue_row(iy);
        wait_vmcnt<0>();
        __syncthreads();
#ifdef SD_PP_TRACE
        unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        // One wave per SIMD: nothing else hides the row form of the previous output row (4 items of ~60 vector instructions + a store)
        // or the LDS-DMA issue (one instruction per 66 cycles and wave), so both are spread between the MFMAs of this row (an MFMA
        // holds the vector issue 8 of its 32 cycles).  LDS ordering: A fragment i + 6 is read at step i (rolling window of six), the
        // scratch reads S(it) of item `it` go out six steps before the item; every step waits `lgkmcnt(6)` = everything but the six
        // youngest operations is back (LDS operations return in order), so S(it) has long landed when item `it` starts.  The first row of a unit has no previous row: the item work is
        // skipped (wave-uniform branches).
        const uint32_t scr_base = lds_addr(scr);
        for (int y = y0; y < y1; ++y) {
            PP_T(r0_)
#if defined(SD_RS_ABL) && SD_RS_ABL != 3
            const bool prev = false;                                          // timing experiment (WRONG RESULTS): no row form of the previous row
#else
            const bool prev = y > y0;
#endif
            const int64_t srow = (((int64_t)b * p.H + y - 1) * p.W + x0 + wave * 32) * 64;      // the previous row (items are skipped on a unit's first row)
#if defined(SD_RS_ABL) && SD_RS_ABL >= 2
            const bool more = false;                                          // timing experiment (WRONG RESULTS): no LDS-DMA in the loop
#else
            const bool more = y + 3 <= y1;
#endif
            const int dslot_row = (y + 3 - y0 + 1) % RS_NR;
            float* const ddst = reinterpret_cast<float*>(ring + dslot_row * RS_ROW_BYTES);
            f32x16 acc[2];                                                    // (the first MFMA of the row starts them from 0)
            uint32_t sb[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) sb[r] = ring_base + (uint32_t)((y - 1 + r - y0 + 1) % RS_NR) * RS_ROW_BYTES;
            f32x4 A[8], S[2][2];                                              // A: rolling window, read i + 6 is issued at step i
#define RS_ADDR(i) (sb[(i) / 12] + aoff[((i) % 12) / 4][(i) % 4])
#define RS_SCR_RD(it) { const uint32_t sa_ = scr_base + (uint32_t)(((spx + 8 * (it)) * RS_SCR + sc8 * 8) * 4); \
                        S[(it) & 1][0] = lds_read128_async<0>(sa_); S[(it) & 1][1] = lds_read128_async<16>(sa_); }
            RS_SCR_RD(0)
#pragma unroll
            for (int u = 0; u < 6; ++u) A[u] = lds_read128_async<0>(RS_ADDR(u));
            uint4 resn[4];                                                    // residual of THIS row (used one row later)
            float f[8]; uint4 o;
#define RS_MFMA2(G, u)                                                                                                                \
            {                                                                                                                         \
                constexpr int i_ = 6 * (G) + (u), t_ = (i_ / 12) * 3 + (i_ % 12) / 4, kc_ = i_ % 4, b0_ = (t_ * 4 + kc_) * 2;           \
                /* read i + 6 goes out, then all but the six youngest LDS operations must be back: read i is (the scratch reads that sit \
                   between the A reads only make the wait stricter; LDS operations return in order) */                                   \
                if (i_ + 6 < 36) A[(i_ + 6) & 7] = lds_read128_async<0>(RS_ADDR(i_ + 6 < 36 ? i_ + 6 : 0));                            \
                /* the fragment the MFMAs just issued are still reading stays allocated: hipcc otherwise computes the next read address \
                   (a VALU write) and lands the read in those very registers, and the write waits for the MFMA's operand reads */       \
                if (i_ > 0) asm volatile("" :: "v"(A[(i_ - 1) & 7]));                                                                  \
                if (SD_RS_NOWAIT && i_ + 6 < 36) asm volatile("" : "+v"(A[i_ & 7]) :: "memory");                                        \
                else if (i_ + 6 < 36) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(A[i_ & 7]) :: "memory");                               \
                else if (i_ == 30) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(A[i_ & 7]) :: "memory");                                  \
                else if (i_ == 31) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(A[i_ & 7]) :: "memory");                                  \
                else if (i_ == 32) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[i_ & 7]) :: "memory");                                  \
                else if (i_ == 33) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[i_ & 7]) :: "memory");                                  \
                else if (i_ == 34) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(A[i_ & 7]) :: "memory");                                  \
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[i_ & 7]) :: "memory");                                                \
                rs_mfma<(b0_ < RS_B_AGPR), i_ == 0>(acc[0], A[i_ & 7], Bw[b0_]);                                                       \
                rs_mfma<(b0_ + 1 < RS_B_AGPR), i_ == 0>(acc[1], A[i_ & 7], Bw[b0_ + 1]);                                               \
                __builtin_amdgcn_sched_barrier(0);                                                                                    \
            }
#define RS_DMA_PIECE(pc_expr, chk)                                                                                                    \
            {                                                                                                                         \
                const int pc = (pc_expr), pxr = pc * 8 + dpx, ix = x0 - 1 + pxr;                                                       \
                const bool ok = (!(chk) || pxr < 130) && (unsigned)ix < (unsigned)p.W && (unsigned)(y + 3) < (unsigned)p.H;            \
                lds_dma16(ok ? img + ((int64_t)(y + 3) * p.W + ix) * 64 + ((dslot ^ ((pxr >> 1) & 7)) << 3) : zero_ + (dslot << 3), ddst + pc * 256); \
            }
            // group G: item G of the previous row (G < 4) and, in group 0, the LDS-DMA of row y + 3, cut into pieces between the MFMAs
#define RS_GROUP(G)                                                                                                                   \
            {                                                                                                                         \
                RS_MFMA2(G, 0)                                                                                                        \
                /* S(G) was issued six steps ago, in front of A reads that have been waited for since */                              \
                if ((G) < 4) asm volatile("" : "+v"(S[(G) & 1][0]), "+v"(S[(G) & 1][1]));                                              \
                if ((G) < 3) RS_SCR_RD((G) + 1)                                                                                       \
                if ((G) < 4 && prev) {                                                                                                \
                    const f32x4 v0 = S[(G) & 1][0], v1 = S[(G) & 1][1];                                                               \
                    f[0] = v0[0]; f[1] = v0[1]; f[2] = v0[2]; f[3] = v0[3]; f[4] = v1[0]; f[5] = v1[1]; f[6] = v1[2]; f[7] = v1[3];    \
                    if (p.res) {                                                                                                      \
                        const uint4 r = resv[(G) & 3];                                                                                \
                        f[0] += __uint_as_float(r.x << 16); f[1] += __uint_as_float(r.x & 0xffff0000u); f[2] += __uint_as_float(r.y << 16); f[3] += __uint_as_float(r.y & 0xffff0000u); \
                        f[4] += __uint_as_float(r.z << 16); f[5] += __uint_as_float(r.z & 0xffff0000u); f[6] += __uint_as_float(r.w << 16); f[7] += __uint_as_float(r.w & 0xffff0000u); \
                    }                                                                                                                 \
                }                                                                                                                     \
                if ((G) == 0 && more) { RS_DMA_PIECE(wave, 0) RS_DMA_PIECE(wave + 4, 0) }                                             \
                __builtin_amdgcn_sched_barrier(0);                                                                                    \
                RS_MFMA2(G, 1)                                                                                                        \
                if ((G) < 4 && prev) {   /* one v_cvt_pk_bf16_f32 per pair; ReLU on the packed pairs: max as signed 16-bit integers */ \
                    o.x = rs_pack2(f[0], f[1]); o.y = rs_pack2(f[2], f[3]); o.z = rs_pack2(f[4], f[5]); o.w = rs_pack2(f[6], f[7]);    \
                    if (p.relu) { o.x = rs_relu2(o.x); o.y = rs_relu2(o.y); o.z = rs_relu2(o.z); o.w = rs_relu2(o.w); }                \
                }                                                                                                                     \
                if ((G) == 0 && more) RS_DMA_PIECE(wave + 8, 0)                                                                       \
                __builtin_amdgcn_sched_barrier(0);                                                                                    \
                RS_MFMA2(G, 2)                                                                                                        \
                if ((G) < 4 && prev && p.stat) {                                                                                      \
                    const uint32_t od[4] = {o.x, o.y, o.z, o.w};                                                                      \
                    _Pragma("unroll") for (int k = 0; k < 4; ++k) {                                                                   \
                        const float a0 = __uint_as_float(od[k] << 16), a1 = __uint_as_float(od[k] & 0xffff0000u);                      \
                        ssum[2 * k] += a0; ssq[2 * k] += a0 * a0; ssum[2 * k + 1] += a1; ssq[2 * k + 1] += a1 * a1;                    \
                    }                                                                                                                 \
                }                                                                                                                     \
                if ((G) == 0 && more) RS_DMA_PIECE(wave + 12, 0)                                                                      \
                __builtin_amdgcn_sched_barrier(0);                                                                                    \
                RS_MFMA2(G, 3)                                                                                                        \
                if ((G) < 4 && prev) *reinterpret_cast<uint4*>(p.y + srow + (spx + 8 * ((G) & 3)) * 64 + sc8 * 8) = o;                 \
                if ((G) == 0 && more && wave == 0) RS_DMA_PIECE(16, 1)                                                                \
                __builtin_amdgcn_sched_barrier(0);                                                                                    \
                RS_MFMA2(G, 4)                                                                                                        \
                if ((G) < 4 && p.res) {   /* residual of this row, item G: consumed one row later */                                  \
                    resn[(G) & 3] = *reinterpret_cast<const uint4*>(p.res + (((int64_t)b * p.H + y) * p.W + x0 + wave * 32) * 64 + (spx + 8 * ((G) & 3)) * 64 + sc8 * 8); \
                }                                                                                                                     \
                __builtin_amdgcn_sched_barrier(0);                                                                                    \
                RS_MFMA2(G, 5)                                                                                                        \
            }
            RS_GROUP(0) RS_GROUP(1) RS_GROUP(2) RS_GROUP(3) RS_GROUP(4) RS_GROUP(5)
#undef RS_GROUP
#undef RS_DMA_PIECE
#undef RS_MFMA2
            // the asm MFMAs are invisible to the hazard recogniser: their results must not be read (v_accvgpr_read) for 18 wait states
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#undef RS_ADDR
#undef RS_SCR_RD
            if (p.res) {
#pragma unroll
                for (int it = 0; it < 4; ++it) resv[it] = resn[it];
            }
            PP_T(r3_)
            // the DMA pieces of row y + 3 went out in group 0 (~2000 cycles ago); what the next row needs (row y + 2) was issued a row ago.
            // vmcnt(0) also covers this row's four stores and residual loads (issued in groups 0 .. 3)
            wait_vmcnt<0>();
            PP_T(r4_)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    scr[((e & 3) + 8 * (e >> 2) + 4 * fh) * RS_SCR + ni * 32 + fr] = acc[ni][e] * sc[ni] + sh[ni];
            PP_T(r5_)
            __syncthreads();                 // every wave is done with input row y - 1 and has published its pieces of row y + 3
            PP_T(r6_)
            PP_ACC(2, r0_, r3_) PP_ACC(3, r3_, r4_) PP_ACC(4, r4_, r5_) PP_ACC(5, r5_, r6_)
#ifdef SD_PP_TRACE
            tr[7] += 1;
#endif
        }
#ifdef SD_PP_TRACE
        if (blockIdx.x == 8 && lane == 0) { for (int k = 0; k < 8; ++k) g_pp_trace[wave][k] = tr[k]; }
#endif
        store_row(y1 - 1);
        if (p.stat) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                float a = ssum[k], q = ssq[k];
                for (int o = 8; o < 64; o <<= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
                if (lane < 8) { sred[wave * 128 + lane * 8 + k] = a; sred[wave * 128 + 64 + lane * 8 + k] = q; }
            }
            __syncthreads();
            if (tid < 128) p.stat[(int64_t)unit * 128 + tid] = (sred[tid] + sred[128 + tid]) + (sred[256 + tid] + sred[384 + tid]);
        }
        wait_vmcnt<0>();
        __syncthreads();                     // the ring, the scratch and sred are reused by the next unit
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 3x3 / stride 1 / pad 1 convolution of the 64 -> 64 channel layers (layer1; forward and flipped-tap data-gradient) as a ROW
// STREAM with the weights in registers -- the fp32 twin of k_conv3x3_c64_rows_bf16.  K is only 576: a tile kernel re-stages the
// 147 KB of weights for every 128-pixel tile and spends 10 % of a tile in prologue + epilogue (k_conv_igemm<64, 0>: 115 TFLOP/s).
// Here a persistent block (4 waves, one per SIMD) walks down a 64-pixel-wide strip of one image, one output row per step:
//   weights : wave (pixel half ph, channel half ch) holds the B operands of its 32 output channels for all 9 taps x 64 input
//             channels = 288 registers (240 AGPRs + 48 VGPRs), loaded once per block;
//   input   : ring of 5 input rows (72 pixels x 256 bytes; 16-byte slot c of pixel x at c ^ (x & 7): two-way ds_read_b128 conflicts, far from binding)
//             filled by LDS-DMA three rows ahead; the reduction index inside a tap is permuted so that a lane's four k of four
//             consecutive 32x32x2 MFMAs are ONE 16-byte read: k-step (q, e) <-> input channel 8 q + 4 (lane / 32) + e;
//   a row   : 72 reads + 288 MFMAs (18432 MFMA cycles) per wave.  With ONE wave per SIMD every VALU instruction between two MFMAs costs
//             16 cycles of the matrix pipe (tools/micro/mfma_f32_mix.hip: 64.0 -> 68.0 cycles per MFMA for one v_add per four;
//             ds_read / s_waitcnt / SALU cost nothing), so the loop holds no VALU at all: the 36 read addresses of a row (3 rows x 3
//             tap columns x 4 slot groups; the upper slot half is an immediate offset) are formed before its first MFMA;
//             the epilogue (scale / shift, residual, ReLU, column sums, stores straight from the accumulator layout: a wave's element e is
//             two pixels x 32 channels = two full 128-byte lines) follows the row;
//   unit    : (image, strip, `rows` consecutive output rows); statistics: one partial row per unit.
// ---------------------------------------------------------------------------------------------
constexpr int RF_PX = 72, RF_ROW_BYTES = RF_PX * 256, RF_NR = 5, RF_B_AGPR = 240;
constexpr int RF_LDS_BYTES = RF_NR * RF_ROW_BYTES + 4 * 64 * 4;

struct RowsArgsF {
    const float* x;        // [B][H][W][64]
    const float* w;        // [64][9][64] (forward: as stored; data-gradient: the transposed copy, taps walked backwards)
    float* y;              // [B][H][W][64]
    const float* scale;    // per output channel, or null
    const float* shift;
    const float* res;      // same shape as y, or null
    float* stat;           // [nunits][2][64] column sums / sums of squares of the stored values, or null
    int relu, flip;
    int B, H, W, segs, rows, units_per_col, nunits;
};

template <bool B_IN_AGPR, bool ZERO>
__device__ __forceinline__ void rf_mfma(f32x16& acc, float a, float b) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (ZERO) {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=a"(acc) : "v"(a), "v"(b));
    } else {
        if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "a"(b));
        else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
    }
#else
    (void)acc; (void)a; (void)b;
#endif
}

__global__ __launch_bounds__(256) void k_conv3x3_c64_rows_f32(RowsArgsF p) {
    extern __shared__ __attribute__((aligned(16))) float rf_lds[];
    char* const ring = reinterpret_cast<char*>(rf_lds);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ph = wave & 1, chh = wave >> 1;
    float* const sred = reinterpret_cast<float*>(ring + RF_NR * RF_ROW_BYTES);                          // [4 waves][2][32]
    const int fr = lane & 31, fh = lane >> 5;
    const float* const zero_ = g_zero_line;

    // ---- every B operand of the wave: lane (n = 32 chh + fr, k-half fh), tap t, k-step 4 q + e <-> input channel 8 q + 4 fh + e
    float Bw[288];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(p.w + ((chh * 32 + fr) * 9 + (p.flip ? 8 - t : t)) * 64 + 8 * q + 4 * fh);
            Bw[t * 32 + 4 * q + 0] = v.x; Bw[t * 32 + 4 * q + 1] = v.y; Bw[t * 32 + 4 * q + 2] = v.z; Bw[t * 32 + 4 * q + 3] = v.w;
        }
    const float sc = p.scale ? p.scale[chh * 32 + fr] : 1.f, sh = p.shift ? p.shift[chh * 32 + fr] : 0.f;
    // A read offsets inside a ring row: output pixel 32 ph + fr, tap column s -> ring pixel 32 ph + fr + s (ring pixel 0 = image column x0 - 1)
    uint32_t aoff[3][4];                                   // slot group qq (k-steps q = qq and qq + 4: + 128 bytes) of tap column s3
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int pxr = ph * 32 + fr + s3;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) aoff[s3][qq] = (uint32_t)pxr * 256u + ((uint32_t)((2 * qq + fh) ^ (pxr & 7)) << 4);
    }
    const int dpx = lane >> 4, dslot = lane & 15;         // LDS-DMA: lane -> (pixel within a 4-pixel piece, physical 16-byte slot)

    for (int unit = blockIdx.x; unit < p.nunits; unit += gridDim.x) {
        const int col = unit / p.units_per_col, yu = unit - col * p.units_per_col;
        const int b = col / p.segs, x0 = (col - b * p.segs) * 64;
        const int y0 = yu * p.rows, y1 = min(y0 + p.rows, p.H);
        const float* const img = p.x + (int64_t)b * p.H * p.W * 64;
        // input row iy -> ring slot (iy - y0 + 1) % 5; wave w loads pieces w, w + 4, ... of the row's 18
        auto issue_row = [&](int iy) {
            float* const dst = reinterpret_cast<float*>(ring + ((iy - y0 + 1) % RF_NR) * RF_ROW_BYTES);
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int pc = wave + 4 * j;
                if (pc < 18) {
                    const int pxr = pc * 4 + dpx, ix = x0 - 1 + pxr;
                    const bool ok = pxr < 66 && (unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H;
                    lds_dma16(ok ? img + ((int64_t)iy * p.W + ix) * 64 + ((dslot ^ (pxr & 7)) << 2) : zero_ + (dslot << 2), dst + pc * 256);
                }
            }
        };
        float ssum = 0.f, ssq = 0.f;
        for (int iy = y0 - 1; iy <= min(y0 + 2, y1); ++iy) issue_row(iy);
        wait_vmcnt<0>();
        __syncthreads();
        const uint32_t ring_base = lds_addr(ring);
#ifdef SD_PP_TRACE
        unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
        for (int y = y0; y < y1; ++y) {
            PP_T(r0_)
            // the residual of this row: requested now, consumed after the MFMAs
            float resv[16];
            // accumulator layout: lane = (channel 32 chh + fr, pixels 4 fh + (e & 3) + 8 (e >> 2)); element e of a wave = two pixels x 32
            // channels = two full 128-byte lines, so the row is stored (and its residual read) straight from that layout
            const int64_t orow = (((int64_t)b * p.H + y) * p.W + x0 + ph * 32 + 4 * fh) * 64 + chh * 32 + fr;
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 16; ++e) resv[e] = p.res[orow + ((e & 3) + 8 * (e >> 2)) * 64];
            }
            // the LDS-DMA of input row y + 3 (this wave's pieces w, w + 4, ...): sources formed here, issued between the first MFMAs
            const bool more = y + 3 <= y1;
            float* const ddst = reinterpret_cast<float*>(ring + ((y + 3 - y0 + 1) % RF_NR) * RF_ROW_BYTES);
            const float* dsrc[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int pxr = (wave + 4 * j) * 4 + dpx, ix = x0 - 1 + pxr;
                const bool ok = pxr < 66 && (unsigned)ix < (unsigned)p.W && (unsigned)(y + 3) < (unsigned)p.H;
                dsrc[j] = ok ? img + ((int64_t)(y + 3) * p.W + ix) * 64 + ((dslot ^ (pxr & 7)) << 2) : zero_ + (dslot << 2);
                asm volatile("" : "+v"(dsrc[j]));            // formed HERE: hipcc otherwise sinks these VALU instructions between the MFMAs
            }
            uint32_t ad[3][3][4];                                             // every read address of the row: no VALU between the MFMAs
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const uint32_t sb = ring_base + (uint32_t)((y - 1 + r - y0 + 1) % RF_NR) * RF_ROW_BYTES;
#pragma unroll
                for (int s3 = 0; s3 < 3; ++s3)
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq) ad[r][s3][qq] = sb + aoff[s3][qq];
            }
            f32x16 acc;                                                       // (the first MFMA of the row starts it from 0)
            // read i (tap i / 8, k-steps 4 (i % 8) ..) is issued three groups before its four MFMAs
#define RF_RD(i) lds_read128_async<(((i) % 8) >> 2) * 128>(ad[((i) / 8) / 3][((i) / 8) % 3][(i) % 4])
            f32x4 A[4];
            PP_T(r1_)
            A[0] = RF_RD(0); A[1] = RF_RD(1); A[2] = RF_RD(2);
#define RF_STEP(i)                                                                                                     \
            {                                                                                                          \
                if ((i) + 3 < 72) A[((i) + 3) & 3] = RF_RD((i) + 3 < 72 ? (i) + 3 : 0);                                   \
                /* LDS operations return in order: all but the three youngest are back */                              \
                if ((i) + 3 < 72) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[(i) & 3]) :: "memory");                   \
                else if ((i) == 69) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(A[(i) & 3]) :: "memory");                 \
                else if ((i) == 70) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(A[(i) & 3]) :: "memory");                 \
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[(i) & 3]) :: "memory");                                \
                if ((i) < 5 && more && wave + 4 * (i) < 18) lds_dma16(dsrc[(i) < 5 ? (i) : 0], ddst + (wave + 4 * (i)) * 256);  \
                const f32x4 a_ = A[(i) & 3];                                                                           \
                rf_mfma<(4 * (i) + 0 < RF_B_AGPR), (i) == 0>(acc, a_[0], Bw[4 * (i) + 0]);                              \
                rf_mfma<(4 * (i) + 1 < RF_B_AGPR), false>(acc, a_[1], Bw[4 * (i) + 1]);                                 \
                rf_mfma<(4 * (i) + 2 < RF_B_AGPR), false>(acc, a_[2], Bw[4 * (i) + 2]);                                 \
                rf_mfma<(4 * (i) + 3 < RF_B_AGPR), false>(acc, a_[3], Bw[4 * (i) + 3]);                                 \
                __builtin_amdgcn_sched_barrier(0);                                                                     \
            }
#define RF_TAP(t) RF_STEP(8 * (t)) RF_STEP(8 * (t) + 1) RF_STEP(8 * (t) + 2) RF_STEP(8 * (t) + 3) RF_STEP(8 * (t) + 4) RF_STEP(8 * (t) + 5) RF_STEP(8 * (t) + 6) RF_STEP(8 * (t) + 7)
            RF_TAP(0) RF_TAP(1) RF_TAP(2) RF_TAP(3) RF_TAP(4) RF_TAP(5) RF_TAP(6) RF_TAP(7) RF_TAP(8)
#undef RF_TAP
#undef RF_STEP
#undef RF_RD
            // the asm MFMAs are invisible to the hazard recogniser: their results must not be read for 18 wait states
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            PP_T(r2_)
            // Everything this wave has in flight is a row old by now (its pieces of row y + 3 and the residual from the row's start, the
            // previous row's stores): this wait is free -- the stores below stay in flight across the barrier (waiting for THEM at the
            // row's end cost a quarter of the row: a store round trip is ~4800 cycles)
            wait_vmcnt<0>();
            PP_T(r3_)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = acc[e] * sc + sh;
                if (p.res) v += resv[e];
                if (p.relu) v = fmaxf(v, 0.f);
                ssum += v; ssq += v * v;
                p.y[orow + ((e & 3) + 8 * (e >> 2)) * 64] = v;
            }
            PP_T(r4_)
            __builtin_amdgcn_s_barrier();    // every wave is done with input row y - 1 and has published its pieces of row y + 3
            PP_T(r5_)
            PP_ACC(0, r0_, r1_) PP_ACC(1, r1_, r2_) PP_ACC(2, r2_, r3_) PP_ACC(3, r3_, r4_) PP_ACC(4, r4_, r5_)
#ifdef SD_PP_TRACE
            tr[7] += 1;
#endif
        }
#ifdef SD_PP_TRACE
        if (blockIdx.x == 8 && lane == 0) { for (int kk = 0; kk < 8; ++kk) g_pp_trace[wave][kk] = tr[kk]; }
#endif
        if (p.stat) {
            ssum += __shfl_xor(ssum, 32); ssq += __shfl_xor(ssq, 32);
            if (fh == 0) { sred[wave * 64 + fr] = ssum; sred[wave * 64 + 32 + fr] = ssq; }
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, n = tid & 63, cw = (n >> 5) * 2;          // the two pixel-half waves of the channel half
                p.stat[(int64_t)unit * 128 + tid] = sred[cw * 64 + which * 32 + (n & 31)] + sred[(cw + 1) * 64 + which * 32 + (n & 31)];
            }
        }
        __syncthreads();                     // the ring and sred are reused by the next unit
    }
}

#undef SD_BNRED_TERM

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[n][tap][c] = sum_m dY[m][n] * X[pix(m, tap)][c]  -- a GEMM whose reduction
// index is the pixel.  Output tile 128 (n) x 128 (c-chunk, one tap) per block, split over
// `splits` pixel ranges (grid.y); partial tiles are summed by k_wgrad_reduce (deterministic).
// Operands are read from LDS "transposed" (k = pixel is the slow LDS index), which for
// ds_read_b32 with 32 consecutive n / c per half-wave is conflict-free.
// ---------------------------------------------------------------------------------------------
struct WgradArgs {
    const float* dy;   // [M][Nn]
    const float* x;    // NHWC [B][Hi][Wi][Ck]
    float* part;       // [splits][Nn][R*S][Ck]
    int B, Hi, Wi, Ck, Ho, Wo, Nn, R, S, stride, pad;
    int M, splits, m_per_split;
    int strips, chunks_total, chunks_per_split;   // k_wgrad3x3_ring: 32-pixel column strips per map row; chunk id = (image * strips + strip) * Ho + row
};

template <int TN, int TC>   // tile sizes along n (Cout) and c (Cin): 128/64 and 128/64
__global__ __launch_bounds__(256, 2) void k_conv_wgrad(WgradArgs p) {
    constexpr int PK = 32;                        // pixels per chunk
    constexpr int LDN = TN + 4, LDC = TC + 4;     // row pads keep the ds_write_b128 rows 16-byte aligned
    constexpr int NT = TN / 64, CT = TC / 64;     // wave tile = TN/2 x TC/2 -> (TN/64) x (TC/64) MFMA tiles
    constexpr int DV = PK * TN / 4 / 256, XV = PK * TC / 4 / 256;   // float4 slots per thread: 4 or 2
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* Ds = lds;                              // [2][PK][LDN]   dY chunk  (k = pixel, n)
    float* Xs = lds + 2 * PK * LDN;               // [2][PK][LDC]   X chunk   (k = pixel, c)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c_tiles = p.Ck / TC, n_tiles = p.Nn / TN;
    int t = blockIdx.x;
    const int ct = t % c_tiles; t /= c_tiles;
    const int nt = t % n_tiles; t /= n_tiles;
    const int tap = t;
    const int r = tap / p.S, s = tap - r * p.S;
    const int n0 = nt * TN, c0 = ct * TC;
    const int split = blockIdx.y;
    const int m_beg = split * p.m_per_split, m_end = min(m_beg + p.m_per_split, p.M);

    // staging slots (named variables, see the note on k_conv_igemm): slot i -> flat float4 index tid + 256*i
    float4 rd0, rd1, rd2, rd3, rx0, rx1, rx2, rx3;
    rd2 = rd3 = rx2 = rx3 = make_float4(0.f, 0.f, 0.f, 0.f);
#define SD_DSLOT(i) const int drow##i = (tid + 256 * i) / (TN / 4), dcol##i = ((tid + 256 * i) % (TN / 4)) * 4;
#define SD_XSLOT(i)                                                                                   \
    const int xrow##i = (tid + 256 * i) / (TC / 4), xcol##i = ((tid + 256 * i) % (TC / 4)) * 4;      \
    int pxx##i, pxy##i, pxb##i;                                                                       \
    { const int m = m_beg + xrow##i; pxx##i = m % p.Wo; pxy##i = (m / p.Wo) % p.Ho; pxb##i = m / (p.Wo * p.Ho); }
    SD_DSLOT(0) SD_DSLOT(1) SD_DSLOT(2) SD_DSLOT(3)
    SD_XSLOT(0) SD_XSLOT(1) SD_XSLOT(2) SD_XSLOT(3)
#undef SD_DSLOT
#undef SD_XSLOT
    int ld_m = m_beg;                               // chunks are visited strictly in order
    // one chunk = PK pixels further, decomposed once into (images, rows, columns): branch-free stepping
    const int adv_b = PK / (p.Wo * p.Ho), adv_r = PK - adv_b * p.Wo * p.Ho;
    const int adv_y = adv_r / p.Wo, adv_x = adv_r - adv_y * p.Wo;
#define SD_LOAD_D(i)                                                                                  \
    {                                                                                                 \
        const int m = ld_m + drow##i;                                                                 \
        const float* src = m < m_end ? p.dy + (int64_t)m * p.Nn + n0 + dcol##i : g_zero_line;        \
        rd##i = *reinterpret_cast<const float4*>(src);                                                \
    }
#define SD_LOAD_X(i)                                                                                  \
    {                                                                                                 \
        const int m = ld_m + xrow##i;                                                                 \
        const int iy = pxy##i * p.stride - p.pad + r, ix = pxx##i * p.stride - p.pad + s;             \
        const bool ok = m < m_end && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;  \
        const float* src = ok ? p.x + ((((int64_t)pxb##i * p.Hi + iy) * p.Wi + ix) * p.Ck + c0 + xcol##i) : g_zero_line; \
        rx##i = *reinterpret_cast<const float4*>(src);                                                \
        pxx##i += adv_x; pxy##i += adv_y; pxb##i += adv_b;                                            \
        { const bool wx = pxx##i >= p.Wo; pxx##i -= wx ? p.Wo : 0; pxy##i += wx ? 1 : 0; }            \
        { const bool wy = pxy##i >= p.Ho; pxy##i -= wy ? p.Ho : 0; pxb##i += wy ? 1 : 0; }            \
    }
#define SD_LOAD_CHUNK()                                                                               \
    SD_LOAD_D(0) SD_LOAD_D(1) if (DV == 4) { SD_LOAD_D(2) SD_LOAD_D(3) }                              \
    SD_LOAD_X(0) SD_LOAD_X(1) if (XV == 4) { SD_LOAD_X(2) SD_LOAD_X(3) }                              \
    ld_m += PK;
#define SD_STORE_CHUNK(buf)                                                                           \
    {                                                                                                 \
        float* dd = Ds + (buf) * PK * LDN;                                                            \
        *reinterpret_cast<float4*>(dd + drow0 * LDN + dcol0) = rd0;                                   \
        *reinterpret_cast<float4*>(dd + drow1 * LDN + dcol1) = rd1;                                   \
        if (DV == 4) {                                                                                \
            *reinterpret_cast<float4*>(dd + drow2 * LDN + dcol2) = rd2;                               \
            *reinterpret_cast<float4*>(dd + drow3 * LDN + dcol3) = rd3;                               \
        }                                                                                             \
        float* xd = Xs + (buf) * PK * LDC;                                                            \
        *reinterpret_cast<float4*>(xd + xrow0 * LDC + xcol0) = rx0;                                   \
        *reinterpret_cast<float4*>(xd + xrow1 * LDC + xcol1) = rx1;                                   \
        if (XV == 4) {                                                                                \
            *reinterpret_cast<float4*>(xd + xrow2 * LDC + xcol2) = rx2;                               \
            *reinterpret_cast<float4*>(xd + xrow3 * LDC + xcol3) = rx3;                               \
        }                                                                                             \
    }

    const int wn0 = (wave >> 1) * (TN / 2), wc0 = (wave & 1) * (TC / 2);
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[NT][CT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < CT; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int nchunks = (m_end - m_beg + PK - 1) / PK;
    if (nchunks > 0) {
        SD_LOAD_CHUNK()
        SD_STORE_CHUNK(0)
    }
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const int cur = ch & 1;
        if (ch + 1 < nchunks) { SD_LOAD_CHUNK() }
        const float* Db = Ds + cur * PK * LDN + wn0 + fr;
        const float* Xb = Xs + cur * PK * LDC + wc0 + fr;
        // fragments of step kk+1 are read while the MFMAs of step kk run
        float na0 = Db[fh * LDN], na1 = (NT == 2) ? Db[fh * LDN + 32] : 0.f;
        float nb0 = Xb[fh * LDC], nb1 = (CT == 2) ? Xb[fh * LDC + 32] : 0.f;
#pragma unroll
        for (int kk = 0; kk < PK / 2; ++kk) {
            const float a0 = na0, a1 = na1, b0 = nb0, b1 = nb1;
            if (kk + 1 < PK / 2) {
                na0 = Db[(2 * kk + 2 + fh) * LDN];
                if (NT == 2) na1 = Db[(2 * kk + 2 + fh) * LDN + 32];
                nb0 = Xb[(2 * kk + 2 + fh) * LDC];
                if (CT == 2) nb1 = Xb[(2 * kk + 2 + fh) * LDC + 32];
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch reads ahead of this step's MFMAs (hipcc sinks them otherwise)
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            if (CT == 2) acc[0][CT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][CT - 1], 0, 0, 0);
            if (NT == 2) acc[NT - 1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[NT - 1][0], 0, 0, 0);
            if (NT == 2 && CT == 2) acc[NT - 1][CT - 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[NT - 1][CT - 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (ch + 1 < nchunks) { SD_STORE_CHUNK(cur ^ 1) }
        __syncthreads();
    }
#undef SD_LOAD_D
#undef SD_LOAD_X
#undef SD_LOAD_CHUNK
#undef SD_STORE_CHUNK
#undef SD_STORE_A
#undef SD_STORE_B
#undef SD_DMA_A
#undef SD_DMA_B
#undef SD_DMA_CHUNK
    // D[n][c]: row index (m of the MFMA) = n, column (lane&31) = c  -> contiguous c per half-wave
    float* out = p.part + (int64_t)split * p.Nn * p.R * p.S * p.Ck;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = n0 + wn0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const int c = c0 + wc0 + j * 32 + fr;
                out[((int64_t)n * p.R * p.S + tap) * p.Ck + c] = acc[i][j][e];
            }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of a 3x3 / stride 1 / pad 1 conv (channel counts multiples of 64, rows that are whole 32-pixel chunks):
// ALL NINE taps of a 64 (n) x 64 (c) tile in one block, grid = (pixel splits, (Cout/64) * (Cin/64) tiles).
// PMC showed the one-tap-per-block kernel fabric-bound on the 64-channel layers (4.6 GB of L2 fills per launch for 0.54 GB
// of algorithmic traffic: nine blocks re-read the same dY and X ranges).  Here a block stages one dY chunk (32 pixels of an
// image row, its 64 n) and the 3 x 34 pixel input patch around it (its 64 c) ONCE (LDS-DMA) and runs 9 MFMAs per dY
// operand: 60 staged bytes per MFMA instead of 128.  acc[tap] : 32 (n) x 32 (c) per wave -> 144 accumulator VGPRs.
// Blocks of one pixel split are `splits` apart in the linear block order, a multiple of 8: they share an XCD's L2.
// ---------------------------------------------------------------------------------------------
// ROWW = 32: a chunk is 32 pixels of one image row (Wo % 32 == 0), patch 3 x 34 pixels.
// ROWW = 16: a chunk is two whole rows of a 16-wide map (Wo == 16, Ho even), patch 4 x 18 pixels.
template <int ROWW>
__global__ __launch_bounds__(256, 2) void k_wgrad3x3(WgradArgs p) {
    constexpr int W9_PX = ROWW + 2, W9_ROWS = 32 / ROWW + 2;
    constexpr int W9_PIECES = (W9_ROWS * W9_PX * 16 + 255) / 256 * 4;      // 1 KB DMA pieces, padded to a multiple of 4 (28 / 20)
    constexpr int W9_XFLOATS = W9_PIECES * 256;
    __shared__ __attribute__((aligned(16))) float Ds0[32 * 64];
    __shared__ __attribute__((aligned(16))) float Ds1[32 * 64];
    __shared__ __attribute__((aligned(16))) float Xs0[W9_XFLOATS];
    __shared__ __attribute__((aligned(16))) float Xs1[W9_XFLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int c_tiles = p.Ck >> 6;
    const int tn0 = ((int)blockIdx.y / c_tiles) * 64, tc0 = ((int)blockIdx.y % c_tiles) * 64;   // this block's 64 x 64 (n, c) tile
    const int m_beg = split * p.m_per_split, m_end = min(m_beg + p.m_per_split, p.M);
    const int nchunks = (m_end - m_beg + 31) / 32;
    const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // Patch pieces of this wave (w, w + 4, ...): what depends on the lane only is formed ONCE -- the element offset of the lane's 16-byte
    // slot from the chunk's first pixel, and which image borders can invalidate it.  (Per chunk and piece the source then takes 6 VALU
    // instructions instead of 17: VALU work of either wave of a SIMD takes fp32-MFMA time, tools/micro/mfma_f32_mix.hip; staging was
    // 127 VALU instructions per 144 MFMAs.)
    int poff[W9_PIECES / 4];
    unsigned pbits[W9_PIECES / 4];          // 1: first patch row, 2: last patch row, 4: first patch column, 8: last patch column, 16: past the patch, 32: always
#pragma unroll
    for (int j = 0; j < W9_PIECES / 4; ++j) {
        const int sl = (wave + 4 * j) * 64 + lane;              // 16-byte slot inside the [rows][px][16] patch
        const int r = sl / (W9_PX * 16), rem = sl - r * (W9_PX * 16), px = rem >> 4, c4 = rem & 15;
        poff[j] = ((r - 1) * p.Wi + (px - 1)) * p.Ck + tc0 + c4 * 4;
        pbits[j] = (r == 0 ? 1u : 0u) | (r == W9_ROWS - 1 ? 2u : 0u) | (px == 0 ? 4u : 0u) | (px == W9_PX - 1 ? 8u : 0u) | (r >= W9_ROWS ? 16u : 0u) | 32u;
    }
    const int doff = (lane >> 4) * p.Nn + tn0 + (lane & 15) * 4;
    // stage chunk `ch` into (D, X): 8 + 26 pieces of 1 KB, wave w issues pieces w, w+4, ...
#define W9_STAGE(ch, D, X)                                                                                        \
    {                                                                                                             \
        const int m0 = m_beg + (ch) * 32;                                                                         \
        const int ox0 = m0 % p.Wo, t_ = m0 / p.Wo, oy = t_ % p.Ho, b = t_ / p.Ho;                                 \
        const bool live = m0 < m_end;                                                                             \
        const float* const dbase = p.dy + (int64_t)m0 * p.Nn;                      /* wave-uniform */             \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                           \
            const int q = wave + 4 * j;                                                                           \
            lds_dma16(live ? dbase + q * 4 * p.Nn + doff : g_zero_line, (D) + q * 256);                           \
        }                                                                                                         \
        /* borders this chunk touches (wave-uniform): a slot is zero when one of its bits is among them */       \
        const unsigned cm = 16u | (live ? 0u : 32u) | (oy == 0 ? 1u : 0u) | (oy + W9_ROWS - 2 >= p.Hi ? 2u : 0u) | \
                            (ox0 == 0 ? 4u : 0u) | (ox0 + ROWW >= p.Wi ? 8u : 0u);                                 \
        const float* const xbase = p.x + (((int64_t)b * p.Hi + oy) * p.Wi + ox0) * p.Ck;                          \
        _Pragma("unroll") for (int j = 0; j < W9_PIECES / 4; ++j)                                                 \
            lds_dma16((pbits[j] & cm) ? g_zero_line : xbase + poff[j], (X) + (wave + 4 * j) * 256);                \
    }
    // 16 pixel pairs x 9 taps in four batches of four pairs.  A lane's B operand for (pair kk, tap r, s) is X[r][2 kk + s] (+ its
    // pixel parity in the base address): neighbouring pairs overlap, so a batch needs 3 x 9 X values + 4 dY values (31 LDS
    // reads for 36 MFMAs instead of 40), and the NEXT batch's operands are read while the current batch's MFMAs run -- with a
    // one-pair prefetch distance hipcc merged the reads into ds_read2 pairs that the very next MFMA group consumed, and put
    // an `s_waitcnt lgkmcnt(0)` in front of every other group (LDS latency exposed once per 18 MFMAs).
#define W9_LOAD(AV, XV, bb)                                                                                       \
    {                                                                                                             \
        constexpr int kk0_ = 4 * (bb), prow_ = (2 * kk0_) / ROWW, pcol_ = (2 * kk0_) % ROWW;                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) AV[i] = da[(2 * (kk0_ + i)) * 64];                          \
        _Pragma("unroll") for (int r = 0; r < 3; ++r)                                                             \
            _Pragma("unroll") for (int q = 0; q < 9; ++q) XV[r][q] = xb[((r + prow_) * W9_PX + pcol_ + q) * 64];  \
    }
#define W9_MFMAS(AV, XV)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                             \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i], XV[t / 3][2 * i + t % 3], acc[t], 0, 0, 0);
#define W9_COMPUTE(D, X)                                                                                          \
    {                                                                                                             \
        const float* da = (D) + fh * 64 + wn0 + fr;                                                               \
        const float* xb = (X) + fh * 64 + wc0 + fr;                                                               \
        float a0[4], x0[3][9], a1[4], x1[3][9];                                                                   \
        W9_LOAD(a0, x0, 0)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        W9_LOAD(a1, x1, 1)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        W9_MFMAS(a0, x0)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        W9_LOAD(a0, x0, 2)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        W9_MFMAS(a1, x1)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        W9_LOAD(a1, x1, 3)                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        W9_MFMAS(a0, x0)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        W9_MFMAS(a1, x1)                                                                                          \
    }
#define W9_ITER(CUR)                                                                                              \
    {                                                                                                             \
        if (ch + 1 < nchunks) { W9_STAGE(ch + 1, (CUR) ? Ds0 : Ds1, (CUR) ? Xs0 : Xs1) }                          \
        W9_COMPUTE((CUR) ? Ds1 : Ds0, (CUR) ? Xs1 : Xs0)                                                          \
        __syncthreads();                                                                                          \
        ++ch;                                                                                                     \
    }
    if (nchunks > 0) { W9_STAGE(0, Ds0, Xs0) }
    __syncthreads();
    int ch = 0;
    while (ch < nchunks) {
        W9_ITER(0)
        if (ch < nchunks) W9_ITER(1)
    }
#undef W9_ITER
#undef W9_COMPUTE
#undef W9_MFMAS
#undef W9_LOAD
#undef W9_STAGE
    float* out = p.part + (int64_t)split * p.Nn * 9 * p.Ck;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = tn0 + wn0 + (e & 3) + 8 * (e >> 2) + 4 * fh, c = tc0 + wc0 + fr;
            out[((int64_t)n * 9 + t) * p.Ck + c] = acc[t][e];
        }
}

// ---------------------------------------------------------------------------------------------
// k_wgrad3x3_ring (round 4; maps whose width is a multiple of 32): k_wgrad3x3<32> with the chunks of a block walking DOWN a 32-pixel
// column strip and the 3 x 34-pixel input patch kept as a RING of four patch rows (one separate __shared__ object per row, so that
// the compiler's wait insertion does not hold the fragment reads back for the LDS-DMA in flight into another row): chunk oy needs rows
// oy - 1 .. oy + 1, two of which are already in LDS -- per chunk ONE new row (9 pieces of 1 KB) + the dY chunk (8 pieces) instead of
// 26 + 8: a wave issues 4-5 LDS-DMA instructions per chunk instead of 9 (each costs the wave ~100 cycles and 6 VALU instructions of
// address arithmetic that take fp32-MFMA time).  Same operand reads and MFMAs; a split = a contiguous range of chunk ids (image, strip,
// row), every (image, strip) segment inside it starts with its own prologue.  LDS: 2 x 8 KB (dY) + 4 x 9 KB (rows of 36 pixels) = 52 KB.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_wgrad3x3_ring(WgradArgs p) {
    constexpr int RPX = 36, RFL = RPX * 64;                          // ring row: 36 pixels x 64 floats = 9 pieces of 1 KB
    __shared__ __attribute__((aligned(16))) float Ds0[32 * 64];
    __shared__ __attribute__((aligned(16))) float Ds1[32 * 64];
    __shared__ __attribute__((aligned(16))) float R0[RFL];
    __shared__ __attribute__((aligned(16))) float R1[RFL];
    __shared__ __attribute__((aligned(16))) float R2[RFL];
    __shared__ __attribute__((aligned(16))) float R3[RFL];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int c_tiles = p.Ck >> 6;
    const int tn0 = ((int)blockIdx.y / c_tiles) * 64, tc0 = ((int)blockIdx.y % c_tiles) * 64;
    const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int dpx = lane >> 4, dsl = (lane & 15) * 4;                  // DMA: lane -> (pixel within a 4-pixel piece, first of its 4 floats)
    const int doff = dpx * p.Nn + tn0 + dsl;
    // patch row iy of (image b, strip at ox0) -> ring row RR: piece w and w + 4 by wave w, piece 8 (pixels 32 .. 35) by wave 0
#define G9_PIECE(pc, iy, RR)                                                                                      \
    {                                                                                                             \
        const int px = (pc) * 4 + dpx, ix = ox0 - 1 + px;                                                         \
        const bool ok = px < 34 && (unsigned)(iy) < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;              \
        lds_dma16(ok ? p.x + (((int64_t)b * p.Hi + (iy)) * p.Wi + ix) * p.Ck + tc0 + dsl : g_zero_line, (RR) + (pc) * 256); \
    }
#define G9_ROW(iy, RR) { G9_PIECE(wave, iy, RR) G9_PIECE(wave + 4, iy, RR) if (wave == 0) G9_PIECE(8, iy, RR) }
#define G9_DY(j, D)                                                                                               \
    {                                                                                                             \
        const float* const dbase = p.dy + ((int64_t)(b * p.Ho + oy0 + (j)) * p.Wo + ox0) * p.Nn;                  \
        lds_dma16(dbase + wave * 4 * p.Nn + doff, (D) + wave * 256);                                              \
        lds_dma16(dbase + (wave + 4) * 4 * p.Nn + doff, (D) + (wave + 4) * 256);                                  \
    }
    // the operand reads and MFMAs of k_wgrad3x3<32>, the three patch rows as three objects
#define G9_LOAD(AV, XV, bb, XT, XM, XB)                                                                           \
    {                                                                                                             \
        constexpr int kk0_ = 4 * (bb), pcol_ = 2 * kk0_;                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) AV[i] = da[(2 * (kk0_ + i)) * 64];                          \
        _Pragma("unroll") for (int q = 0; q < 9; ++q) {                                                           \
            XV[0][q] = (XT)[xo + (pcol_ + q) * 64]; XV[1][q] = (XM)[xo + (pcol_ + q) * 64]; XV[2][q] = (XB)[xo + (pcol_ + q) * 64]; \
        }                                                                                                         \
    }
#define G9_MFMAS(AV, XV)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                             \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i], XV[t / 3][2 * i + t % 3], acc[t], 0, 0, 0);
#define G9_COMPUTE(D, XT, XM, XB)                                                                                 \
    {                                                                                                             \
        const float* da = (D) + fh * 64 + wn0 + fr;                                                               \
        float a0[4], x0[3][9], a1[4], x1[3][9];                                                                   \
        G9_LOAD(a0, x0, 0, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_LOAD(a1, x1, 1, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a0, x0)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_LOAD(a0, x0, 2, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a1, x1)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_LOAD(a1, x1, 3, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a0, x0)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a1, x1)                                                                                          \
    }
    // chunk k of the segment: its dY in DC, its rows in XT / XM / XB; chunk k + 1's dY goes to DN and its new row oy0 + k + 2 to XN
#define G9_ITER(DC, DN, XT, XM, XB, XN)                                                                           \
    {                                                                                                             \
        if (k + 1 < nseg) { G9_DY(k + 1, DN) G9_ROW(oy0 + k + 2, XN) }                                            \
        G9_COMPUTE(DC, XT, XM, XB)                                                                                \
        __syncthreads();                                                                                          \
        ++k;                                                                                                      \
    }
    const int xo = fh * 64 + wc0 + fr;
    const int c_beg = split * p.chunks_per_split, c_end = min(c_beg + p.chunks_per_split, p.chunks_total);
    for (int c = c_beg; c < c_end;) {
        const int unit = c / p.Ho, oy0 = c - unit * p.Ho, nseg = min(c_end - c, p.Ho - oy0);
        const int b = unit / p.strips, ox0 = (unit - b * p.strips) * 32;
        c += nseg;
        G9_ROW(oy0 - 1, R0)
        G9_ROW(oy0, R1)
        G9_ROW(oy0 + 1, R2)
        G9_DY(0, Ds0)
        __syncthreads();
        int k = 0;
        while (k < nseg) {
            G9_ITER(Ds0, Ds1, R0, R1, R2, R3)
            if (k < nseg) G9_ITER(Ds1, Ds0, R1, R2, R3, R0)
            if (k < nseg) G9_ITER(Ds0, Ds1, R2, R3, R0, R1)
            if (k < nseg) G9_ITER(Ds1, Ds0, R3, R0, R1, R2)
        }
    }
#undef G9_ITER
#undef G9_COMPUTE
#undef G9_MFMAS
#undef G9_LOAD
#undef G9_DY
#undef G9_ROW
#undef G9_PIECE
    float* out = p.part + (int64_t)split * p.Nn * 9 * p.Ck;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = tn0 + wn0 + (e & 3) + 8 * (e >> 2) + 4 * fh, c = tc0 + wc0 + fr;
            out[((int64_t)n * 9 + t) * p.Ck + c] = acc[t][e];
        }
}

// k_wgrad3x3_ring2: ONE 512-thread block per CU, two groups of four waves on the SAME (n, c) tile, each on its own half of the block's chunks
// with its own dY stages and ring rows; at the end group 1's accumulators are added to group 0's through LDS (three passes of three taps)
// and ONE partial tile is stored: half the partial-sum traffic (75 -> 37 MB per launch) and half the reduce pass.  Both groups run the
// same chunk loop; the group with fewer barrier phases (1 per segment + 1 per chunk) pads with bare barriers.
__global__ __launch_bounds__(512, 1) void k_wgrad3x3_ring2(WgradArgs p) {
    constexpr int RPX = 36, RFL = RPX * 64;                          // ring row: 36 pixels x 64 floats = 9 pieces of 1 KB
    __shared__ __attribute__((aligned(16))) float Ds0a[32 * 64];
    __shared__ __attribute__((aligned(16))) float Ds1a[32 * 64];
    __shared__ __attribute__((aligned(16))) float R0a[RFL];
    __shared__ __attribute__((aligned(16))) float R1a[RFL];
    __shared__ __attribute__((aligned(16))) float R2a[RFL];
    __shared__ __attribute__((aligned(16))) float R3a[RFL];
    __shared__ __attribute__((aligned(16))) float Ds0b[32 * 64];
    __shared__ __attribute__((aligned(16))) float Ds1b[32 * 64];
    __shared__ __attribute__((aligned(16))) float R0b[RFL];
    __shared__ __attribute__((aligned(16))) float R1b[RFL];
    __shared__ __attribute__((aligned(16))) float R2b[RFL];
    __shared__ __attribute__((aligned(16))) float R3b[RFL];
    __shared__ float red_buf[3 * 16 * 256];              // 48 KB: group 1 -> group 0 (104 + 48 = 152 KB: one block per CU)
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    const int split = blockIdx.x;
    const int c_tiles = p.Ck >> 6;
    const int tn0 = ((int)blockIdx.y / c_tiles) * 64, tc0 = ((int)blockIdx.y % c_tiles) * 64;
    const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const int dpx = lane >> 4, dsl = (lane & 15) * 4;                  // DMA: lane -> (pixel within a 4-pixel piece, first of its 4 floats)
    const int doff = dpx * p.Nn + tn0 + dsl;
    // patch row iy of (image b, strip at ox0) -> ring row RR: piece w and w + 4 by wave w, piece 8 (pixels 32 .. 35) by wave 0
#define G9_PIECE(pc, iy, RR)                                                                                      \
    {                                                                                                             \
        const int px = (pc) * 4 + dpx, ix = ox0 - 1 + px;                                                         \
        const bool ok = px < 34 && (unsigned)(iy) < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;              \
        lds_dma16(ok ? p.x + (((int64_t)b * p.Hi + (iy)) * p.Wi + ix) * p.Ck + tc0 + dsl : g_zero_line, (RR) + (pc) * 256); \
    }
#define G9_ROW(iy, RR) { G9_PIECE(wave, iy, RR) G9_PIECE(wave + 4, iy, RR) if (wave == 0) G9_PIECE(8, iy, RR) }
#define G9_DY(j, D)                                                                                               \
    {                                                                                                             \
        const float* const dbase = p.dy + ((int64_t)(b * p.Ho + oy0 + (j)) * p.Wo + ox0) * p.Nn;                  \
        lds_dma16(dbase + wave * 4 * p.Nn + doff, (D) + wave * 256);                                              \
        lds_dma16(dbase + (wave + 4) * 4 * p.Nn + doff, (D) + (wave + 4) * 256);                                  \
    }
    // the operand reads and MFMAs of k_wgrad3x3<32>, the three patch rows as three objects
#define G9_LOAD(AV, XV, bb, XT, XM, XB)                                                                           \
    {                                                                                                             \
        constexpr int kk0_ = 4 * (bb), pcol_ = 2 * kk0_;                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) AV[i] = da[(2 * (kk0_ + i)) * 64];                          \
        _Pragma("unroll") for (int q = 0; q < 9; ++q) {                                                           \
            XV[0][q] = (XT)[xo + (pcol_ + q) * 64]; XV[1][q] = (XM)[xo + (pcol_ + q) * 64]; XV[2][q] = (XB)[xo + (pcol_ + q) * 64]; \
        }                                                                                                         \
    }
#define G9_MFMAS(AV, XV)                                                                                          \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
        _Pragma("unroll") for (int t = 0; t < 9; ++t)                                                             \
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i], XV[t / 3][2 * i + t % 3], acc[t], 0, 0, 0);
#define G9_COMPUTE(D, XT, XM, XB)                                                                                 \
    {                                                                                                             \
        const float* da = (D) + fh * 64 + wn0 + fr;                                                               \
        float a0[4], x0[3][9], a1[4], x1[3][9];                                                                   \
        G9_LOAD(a0, x0, 0, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_LOAD(a1, x1, 1, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a0, x0)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_LOAD(a0, x0, 2, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a1, x1)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_LOAD(a1, x1, 3, XT, XM, XB)                                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a0, x0)                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                        \
        G9_MFMAS(a1, x1)                                                                                          \
    }
    // chunk k of the segment: its dY in DC, its rows in XT / XM / XB; chunk k + 1's dY goes to DN and its new row oy0 + k + 2 to XN
#define G9_ITER(DC, DN, XT, XM, XB, XN)                                                                           \
    {                                                                                                             \
        if (k + 1 < nseg) { G9_DY(k + 1, DN) G9_ROW(oy0 + k + 2, XN) }                                            \
        G9_COMPUTE(DC, XT, XM, XB)                                                                                \
        __syncthreads();                                                                                          \
        ++k;                                                                                                      \
    }
    const int xo = fh * 64 + wc0 + fr;
    const int c_beg = split * p.chunks_per_split, c_end = min(c_beg + p.chunks_per_split, p.chunks_total);
    const int c_mid = c_beg + (c_end - c_beg + 1) / 2;
    auto phases = [&](int cb, int ce) { int n = 0; for (int c = cb; c < ce;) { const int oy = c % p.Ho, ns = min(ce - c, p.Ho - oy); n += 1 + ns; c += ns; } return n; };
    const int ph0 = phases(c_beg, c_mid), ph1 = phases(c_mid, c_end);
#define G9_WALK(CB, CE, D0, D1, RA, RB, RC, RD)                                                                   \
    for (int c = (CB); c < (CE);) {                                                                               \
        const int unit = c / p.Ho, oy0 = c - unit * p.Ho, nseg = min((CE) - c, p.Ho - oy0);                       \
        const int b = unit / p.strips, ox0 = (unit - b * p.strips) * 32;                                          \
        c += nseg;                                                                                                \
        G9_ROW(oy0 - 1, RA)                                                                                       \
        G9_ROW(oy0, RB)                                                                                           \
        G9_ROW(oy0 + 1, RC)                                                                                       \
        G9_DY(0, D0)                                                                                              \
        __syncthreads();                                                                                          \
        int k = 0;                                                                                                \
        while (k < nseg) {                                                                                        \
            G9_ITER(D0, D1, RA, RB, RC, RD)                                                                       \
            if (k < nseg) G9_ITER(D1, D0, RB, RC, RD, RA)                                                         \
            if (k < nseg) G9_ITER(D0, D1, RC, RD, RA, RB)                                                         \
            if (k < nseg) G9_ITER(D1, D0, RD, RA, RB, RC)                                                         \
        }                                                                                                         \
    }
    if (grp == 0) { G9_WALK(c_beg, c_mid, Ds0a, Ds1a, R0a, R1a, R2a, R3a) }
    else { G9_WALK(c_mid, c_end, Ds0b, Ds1b, R0b, R1b, R2b, R3b) }
    for (int i = grp ? ph1 : ph0; i < max(ph0, ph1); ++i) __syncthreads();          // equal barrier counts for any split of the range
#undef G9_WALK
    // ---- group 1's sums into group 0: three taps per pass through LDS ([tap][element][thread]: conflict-free), fixed order
    {
        const int t4 = tid & 255;
#pragma unroll
        for (int ps = 0; ps < 3; ++ps) {
            __syncthreads();
            if (grp == 1) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red_buf[(t * 16 + e) * 256 + t4] = acc[ps * 3 + t][e];
            }
            __syncthreads();
            if (grp == 0) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ps * 3 + t][e] += red_buf[(t * 16 + e) * 256 + t4];
            }
        }
    }
#undef G9_ITER
#undef G9_COMPUTE
#undef G9_MFMAS
#undef G9_LOAD
#undef G9_DY
#undef G9_ROW
#undef G9_PIECE
    if (grp != 0) return;
    float* out = p.part + (int64_t)split * p.Nn * 9 * p.Ck;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = tn0 + wn0 + (e & 3) + 8 * (e >> 2) + 4 * fh, c = tc0 + wc0 + fr;
            out[((int64_t)n * 9 + t) * p.Ck + c] = acc[t][e];
        }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient of a 3x3 / stride 1 / pad 1 conv (mixed-precision training): the all-taps tiling of k_wgrad3x3 on the
// bf16 MFMA (v_mfma_f32_32x32x16_bf16, fp32 accumulation, fp32 partial sums).  The reduction index is the PIXEL, but both
// operands are stored pixel-major ([pixel][n] and [pixel][c], channels contiguous), while a lane of the 32x32x16 MFMA holds
// EIGHT CONSECUTIVE reduction indices of one row / column: the operands have to be transposed on the way LDS -> registers.
// gfx950's ds_read_b64_tr_b16 does that for free: per 16-lane group it reads a 4 (pixels) x 16 (channels) block and hands lane i
// the four pixels of channel i -- two reads per operand and 16-pixel step.  The nine taps reuse the dY operand; the X operand of
// tap (r, s) is the same patch read at row offset r, pixel offset s (row addresses are per lane, so the one-pixel shifts need
// no aligned copies).  20 transposed reads per 9 MFMAs.
// LDS images (LDS-DMA, 1 KB pieces = 8 pixel rows of 64 bf16 = 128 B): the 32-byte column slot t of pixel row i lives at slot
// t ^ (i & 3), so the four rows of a transposed read (any four CONSECUTIVE rows: distinct i & 3, and rows two apart differ in
// bit 1) and the two column halves of a 32-lane half fall on 8 disjoint bank groups -- conflict-free at every tap offset.
// The swizzle is applied to the SOURCE address of the DMA (the LDS side of global_load_lds is lane-linear).
// ---------------------------------------------------------------------------------------------
typedef short v4i16 __attribute__((ext_vector_type(4)));
struct alignas(16) TrPair { v4i16 lo, hi; };

__device__ __forceinline__ v4i16 lds_tr16(const uint16_t* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)p);
#else
    (void)p; return v4i16{0, 0, 0, 0};
#endif
}

struct WgradArgs16 {
    const uint16_t* dy;   // [M][Nn] bf16
    const uint16_t* x;    // NHWC [B][Hi][Wi][Ck] bf16
    float* part;          // [splits][Nn][9][Ck] fp32
    int B, Hi, Wi, Ck, Ho, Wo, Nn;
    int M, splits, m_per_split;
    int strips, chunks_total, chunks_per_split;   // k_wgrad3x3_bf16_ring: 32-pixel column strips per map row; chunk id = (image * strips + strip) * Ho + row
};

// Round 4: THREE LDS stages and explicit waits.  The first form (two stages, a __syncthreads() per 32-pixel chunk) spent two thirds of
// its time waiting: the barrier's fence drains the LDS-DMA issued one chunk earlier (a ~1-2 us round trip for 0.27 us of MFMAs per wave).
// Now chunk ch + 2 is issued while chunk ch is multiplied; per chunk one counted vmcnt wait + a bare s_barrier; the transposed reads are
// inline asm (the compiler's wait insertion would drain the DMAs in flight in front of them) with the stage and the tap's row offset as
// immediates: address registers = one for dY and four for X (the slot swizzle depends on (row & 3), i.e. on the tap's row constant mod 4).
#ifndef SD_W16_ABL
#define SD_W16_ABL 0          // timing experiments (WRONG RESULTS): 1 no DMA in the loop, 2 no transposed reads, 3 no MFMA, 4 no partial stores, 5 no barrier
#else
#define SD_W16_ABL_BUILD 1
#endif
template <int OFF>
__device__ __forceinline__ v4i16 lds_tr16_async(uint32_t addr) {
    v4i16 v;
    if (SD_W16_ABL == 2) { v = __builtin_bit_cast(v4i16, uint2{addr, addr ^ 0x3f803f80u}); asm volatile("" : "+v"(v)); return v; }
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
    return v;
}
// patch row constant of (k-block kb, tap t): the lane adds its own pixel `khalf`
template <int ROWW>
constexpr int w16_row(int kb, int t) { return ROWW == 32 ? (t / 3) * 34 + kb * 16 + t % 3 : (kb + t / 3) * 18 + t % 3; }
template <int ROWW, int XB, int KB, int T>
__device__ __forceinline__ void w16_read_tap(const uint32_t (&xo)[4], TrPair& bp) {
    constexpr int C = w16_row<ROWW>(KB, T);
    bp.lo = lds_tr16_async<XB + C * 128>(xo[C & 3]);
    bp.hi = lds_tr16_async<XB + (C + 4) * 128>(xo[C & 3]);
}
#define SD_W16_WAIT10(N, a, b0, b1, b2, b3, b4, b5, b6, b7, b8)                                                   \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(a.lo), "+v"(a.hi), "+v"(b0.lo), "+v"(b0.hi), "+v"(b1.lo), "+v"(b1.hi), "+v"(b2.lo), "+v"(b2.hi), \
                 "+v"(b3.lo), "+v"(b3.hi), "+v"(b4.lo), "+v"(b4.hi), "+v"(b5.lo), "+v"(b5.hi), "+v"(b6.lo), "+v"(b6.hi), "+v"(b7.lo), "+v"(b7.hi),    \
                 "+v"(b8.lo), "+v"(b8.hi) :: "memory")

template <int ROWW>
__global__ __launch_bounds__(256, 2) void k_wgrad3x3_bf16(WgradArgs16 p) {
    constexpr int PX = ROWW + 2, ROWS = 32 / ROWW + 2;
    constexpr int XPIX = ROWS * PX;                                  // patch pixels: 102 / 72
    constexpr int XPIECES = ((XPIX + 7) / 8 + 3) / 4 * 4;            // 1 KB pieces of 8 pixel rows, a multiple of 4: 16 / 12
    constexpr int NP = 1 + XPIECES / 4;                              // LDS-DMA instructions per wave and stage
    constexpr int STG = 2048 + XPIECES * 512;                        // elements per stage: dY chunk (32 x 64) + patch
    constexpr int STGB = STG * 2;                                    // bytes: 20480 / 16384
    __shared__ __attribute__((aligned(16))) uint16_t W3[3 * STG];    // ONE object: the stage is an immediate offset of the reads
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int c_tiles = p.Ck >> 6;
    const int tn0 = ((int)blockIdx.y / c_tiles) * 64, tc0 = ((int)blockIdx.y % c_tiles) * 64;
    const int m_beg = split * p.m_per_split, m_end = min(m_beg + p.m_per_split, p.M);
    const int nchunks = (m_end - m_beg + 31) / 32;
    const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const uint16_t* const zero16 = reinterpret_cast<const uint16_t*>(g_zero_line);

    // DMA source of this lane inside an 8-row piece: row lane / 8, logical 16-byte chunk (lane % 8) ^ ((row & 3) << 1)
    const int srow = lane >> 3;
    const int schunk = ((lane & 7) ^ ((srow & 3) << 1)) * 8;          // element offset inside the 64-channel row
    // chunk ch -> stage ST (always NP instructions per wave: past-the-end chunks re-load the zero line, the vmcnt counts stay fixed)
#define W16_STAGE(ch, ST)                                                                                         \
    {                                                                                                             \
        uint16_t* const D_ = W3 + (ST) * STG;                                                                     \
        uint16_t* const X_ = D_ + 2048;                                                                           \
        const int m0 = m_beg + (ch) * 32;                                                                         \
        const bool live = m0 < m_end;                                                                             \
        const int ox0 = m0 % p.Wo, t_ = m0 / p.Wo, oy = t_ % p.Ho, b = t_ / p.Ho;                                 \
        {                                                                                                         \
            const int row = wave * 8 + srow;                    /* pixel of the chunk: 4 pieces, one per wave */   \
            const uint16_t* src = live ? p.dy + (int64_t)(m0 + row) * p.Nn + tn0 + schunk : zero16;               \
            lds_dma16(src, reinterpret_cast<float*>(D_ + wave * 512));                                            \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < XPIECES / 4; ++j) {                                                 \
            const int q = wave + 4 * j;                                                                           \
            const int idx = q * 8 + srow;                       /* patch pixel: row idx / PX, column idx % PX */   \
            const int r = idx / PX, px = idx - r * PX;                                                            \
            const int iy = oy - 1 + r, ix = ox0 - 1 + px;                                                         \
            const bool ok = live && idx < XPIX && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi; \
            const uint16_t* src = ok ? p.x + (((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.Ck + tc0 + schunk : zero16; \
            lds_dma16(src, reinterpret_cast<float*>(X_ + q * 512));                                               \
        }                                                                                                         \
    }
    // transposed-read geometry of this lane: 16-lane group g = lane / 16 -> column half g & 1, 8-pixel half g >> 1 of the 16-pixel
    // step; lane 4q + pp of the group addresses row q, columns 4 pp .. 4 pp + 3 of the group's 4 x 16 block.  Byte address of (row, slot):
    // row * 128 + ((slot ^ (row & 3)) << 5) + pp * 8 with row = constant + khalf: the constant * 128 is an immediate, the rest per lane
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const int a_slot = (wn0 >> 4) + (g & 1), b_slot = (wc0 >> 4) + (g & 1);
    const int khalf = (g >> 1) * 8 + q4;                             // pixel of the 16-pixel step this lane addresses (first read; +4 second)
    const uint32_t w3 = lds_addr(W3);
    const uint32_t ao0 = w3 + (uint32_t)(khalf * 128 + ((a_slot ^ (khalf & 3)) << 5) + pp * 8);       // dY rows: constants 0, 4, 16, 20
    uint32_t xo0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xo0[k] = w3 + (uint32_t)(khalf * 128 + ((b_slot ^ ((k + khalf) & 3)) << 5) + pp * 8);
#define W16_MFMA(A, BP, T) if (SD_W16_ABL == 3) { asm volatile("" :: "v"(A.lo), "v"(A.hi), "v"(BP.lo), "v"(BP.hi)); } else { acc[T] = SD_MFMA_BF16(4, __builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, BP), acc[T]); }
#define W16_COMPUTE(st)                                                                                           \
    {                                                                                                             \
        constexpr int DB = 0, XB = 4096;                                                                          \
        const uint32_t so_ = (uint32_t)(st) * STGB;             /* the stage: five adds per chunk */               \
        const uint32_t ao = ao0 + so_;                                                                            \
        const uint32_t xo[4] = {xo0[0] + so_, xo0[1] + so_, xo0[2] + so_, xo0[3] + so_};                          \
        TrPair a0, a1, b0, b1, b2, b3, b4, b5, b6, b7, b8;                                                        \
        a0.lo = lds_tr16_async<DB>(ao); a0.hi = lds_tr16_async<DB + 4 * 128>(ao);                                 \
        w16_read_tap<ROWW, XB, 0, 0>(xo, b0); w16_read_tap<ROWW, XB, 0, 1>(xo, b1); w16_read_tap<ROWW, XB, 0, 2>(xo, b2); \
        w16_read_tap<ROWW, XB, 0, 3>(xo, b3); w16_read_tap<ROWW, XB, 0, 4>(xo, b4); w16_read_tap<ROWW, XB, 0, 5>(xo, b5); \
        w16_read_tap<ROWW, XB, 0, 6>(xo, b6); w16_read_tap<ROWW, XB, 0, 7>(xo, b7); w16_read_tap<ROWW, XB, 0, 8>(xo, b8); \
        SD_W16_WAIT10(0, a0, b0, b1, b2, b3, b4, b5, b6, b7, b8);                                                 \
        a1.lo = lds_tr16_async<DB + 16 * 128>(ao); a1.hi = lds_tr16_async<DB + 20 * 128>(ao);                     \
        /* second k-block: a tap's operand registers are re-read as soon as its MFMA has been issued */            \
        W16_MFMA(a0, b0, 0) w16_read_tap<ROWW, XB, 1, 0>(xo, b0);                                                 \
        W16_MFMA(a0, b1, 1) w16_read_tap<ROWW, XB, 1, 1>(xo, b1);                                                 \
        W16_MFMA(a0, b2, 2) w16_read_tap<ROWW, XB, 1, 2>(xo, b2);                                                 \
        W16_MFMA(a0, b3, 3) w16_read_tap<ROWW, XB, 1, 3>(xo, b3);                                                 \
        W16_MFMA(a0, b4, 4) w16_read_tap<ROWW, XB, 1, 4>(xo, b4);                                                 \
        W16_MFMA(a0, b5, 5) w16_read_tap<ROWW, XB, 1, 5>(xo, b5);                                                 \
        W16_MFMA(a0, b6, 6) w16_read_tap<ROWW, XB, 1, 6>(xo, b6);                                                 \
        W16_MFMA(a0, b7, 7) w16_read_tap<ROWW, XB, 1, 7>(xo, b7);                                                 \
        W16_MFMA(a0, b8, 8) w16_read_tap<ROWW, XB, 1, 8>(xo, b8);                                                 \
        SD_W16_WAIT10(0, a1, b0, b1, b2, b3, b4, b5, b6, b7, b8);                                                 \
        W16_MFMA(a1, b0, 0) W16_MFMA(a1, b1, 1) W16_MFMA(a1, b2, 2) W16_MFMA(a1, b3, 3) W16_MFMA(a1, b4, 4)       \
        W16_MFMA(a1, b5, 5) W16_MFMA(a1, b6, 6) W16_MFMA(a1, b7, 7) W16_MFMA(a1, b8, 8)                           \
    }
    // iteration: chunk ch's pieces of this wave have landed (NP younger ones may be in flight), barrier (every wave's pieces are in, and
    // every wave is done reading chunk ch - 1: its stage takes chunk ch + 2), multiply
    W16_STAGE(0, 0)
    W16_STAGE(1, 1)
    int st = 0;                                                   // stage of chunk ch; chunk ch + 2 goes to stage (st + 2) % 3
    for (int ch = 0; ch < nchunks; ++ch) {
        if (SD_W16_ABL != 1) wait_vmcnt<NP>();
        if (SD_W16_ABL != 5) __builtin_amdgcn_s_barrier();
        const int st2 = st == 0 ? 2 : st - 1;
        if (SD_W16_ABL != 1) { W16_STAGE(ch + 2, st2) }
        W16_COMPUTE(st)
        st = st == 2 ? 0 : st + 1;
    }
    wait_vmcnt<0>();
#undef W16_COMPUTE
#undef W16_MFMA
#undef W16_STAGE
    float* out = p.part + (int64_t)split * p.Nn * 9 * p.Ck;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = tn0 + wn0 + (e & 3) + 8 * (e >> 2) + 4 * fh, c = tc0 + wc0 + fr;
            if (SD_W16_ABL != 4 || acc[t][e] == 123.456f) out[((int64_t)n * 9 + t) * p.Ck + c] = acc[t][e];
        }
}

// ---------------------------------------------------------------------------------------------
// k_wgrad3x3_bf16_ring (round 4; maps whose width is a multiple of 32): the same tile, the same MFMAs and transposed reads as
// k_wgrad3x3_bf16<32>, but the chunks of a block walk DOWN a 32-pixel column strip (chunk = row oy of the strip) and the 3 x 34-pixel
// input patch is a RING of five patch rows: chunk oy needs rows oy - 1 .. oy + 1, of which two are already in LDS -- ONE new row
// (5 pieces) + the dY chunk (4 pieces) per chunk instead of 16 + 4.  The ablations of the first form showed the LDS-DMA ISSUE in the compute
// waves as its largest single cost (no DMA in the loop: 125 -> 85 us per layer).  Pipeline as before: chunk k + 2 (its dY and its new row)
// is issued while chunk k is multiplied, one counted vmcnt + bare s_barrier per chunk; wave 0 issues three pieces per chunk, the others
// two (a wave-uniform branch picks the count).  A split = a contiguous range of chunk ids (image, strip, row); every (image, strip) segment
// inside it starts with its own prologue (rows oy0 - 1, oy0 and two stages).
// LDS: three dY stages of 4 KB + five ring rows of 40 pixels x 128 B = 37.9 KB.
// ---------------------------------------------------------------------------------------------
template <int D>          // prefetch distance in chunks: D + 1 dY stages, D + 3 ring rows
__global__ __launch_bounds__(256, 2) void k_wgrad3x3_bf16_ring(WgradArgs16 p) {
    constexpr int DSTG = 4096, RROW = 5120, ND = D + 1, NR = D + 3, RING0 = ND * DSTG;          // bytes
    __shared__ __attribute__((aligned(16))) uint16_t W3[(ND * DSTG + NR * RROW) / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int c_tiles = p.Ck >> 6;
    const int tn0 = ((int)blockIdx.y / c_tiles) * 64, tc0 = ((int)blockIdx.y % c_tiles) * 64;
    const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const uint16_t* const zero16 = reinterpret_cast<const uint16_t*>(g_zero_line);
    const int srow = lane >> 3;
    const int schunk = ((lane & 7) ^ ((srow & 3) << 1)) * 8;          // element offset inside the 64-channel row (source-side swizzle)
    char* const lds_b = reinterpret_cast<char*>(W3);

    // patch row iy of (image b, strip at ox0) -> ring slot: pieces of 8 pixels, piece w by wave w, piece 4 (pixels 32, 33) by wave 0
#define WR_ROW_PIECE(pc, iy, slot)                                                                                \
    {                                                                                                             \
        const int px = (pc) * 8 + srow, ix = ox0 - 1 + px;                                                        \
        const bool ok = px < 34 && (unsigned)(iy) < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;              \
        const uint16_t* src = ok ? p.x + (((int64_t)b * p.Hi + (iy)) * p.Wi + ix) * p.Ck + tc0 + schunk : zero16; \
        lds_dma16(src, reinterpret_cast<float*>(lds_b + RING0 + (slot) * RROW + (pc) * 1024));                    \
    }
#define WR_ROW(iy, slot) { WR_ROW_PIECE(wave, iy, slot) if (wave == 0) WR_ROW_PIECE(4, iy, slot) }
    // stage j of the segment: dY of row oy0 + j (stage j % 3) and patch row oy0 + j + 1 (ring slot (j + 2) % 5); past the segment: the
    // zero line (the counts stay fixed)
#define WR_STAGE(j, dst, slot)                                                                                    \
    {                                                                                                             \
        const bool live = (j) < nseg;                                                                             \
        const int oy = oy0 + (j);                                                                                 \
        const uint16_t* src = live ? p.dy + ((int64_t)(b * p.Ho + oy) * p.Wo + ox0 + wave * 8 + srow) * p.Nn + tn0 + schunk : zero16; \
        lds_dma16(src, reinterpret_cast<float*>(lds_b + (dst) * DSTG + wave * 1024));                             \
        const int iy = live ? oy + 1 : -1;                                                                        \
        WR_ROW(iy, slot)                                                                                          \
    }
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const int a_slot = (wn0 >> 4) + (g & 1), b_slot = (wc0 >> 4) + (g & 1);
    const int khalf = (g >> 1) * 8 + q4;
    const uint32_t w3 = lds_addr(W3);
    const uint32_t ao0 = w3 + (uint32_t)(khalf * 128 + ((a_slot ^ (khalf & 3)) << 5) + pp * 8);
    uint32_t xo0[3];                                                  // tap column s2: pixel s2 + khalf (+ 16 kb, + 4) of a ring row
#pragma unroll
    for (int k = 0; k < 3; ++k) xo0[k] = w3 + RING0 + (uint32_t)((k + khalf) * 128 + ((b_slot ^ ((k + khalf) & 3)) << 5) + pp * 8);
#define WR_MFMA(A, BP, T) if (SD_W16_ABL == 3) { asm volatile("" :: "v"(A.lo), "v"(A.hi), "v"(BP.lo), "v"(BP.hi)); } else { acc[T] = SD_MFMA_BF16(4, __builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, BP), acc[T]); }
#define WR_TAP(KB, T, BP) { BP.lo = lds_tr16_async<(KB) * 16 * 128>(xa[T]); BP.hi = lds_tr16_async<(KB) * 16 * 128 + 4 * 128>(xa[T]); }

    const int c_beg = split * p.chunks_per_split, c_end = min(c_beg + p.chunks_per_split, p.chunks_total);
    for (int c = c_beg; c < c_end;) {
        const int unit = c / p.Ho, oy0 = c - unit * p.Ho, nseg = min(c_end - c, p.Ho - oy0);
        const int b = unit / p.strips, ox0 = (unit - b * p.strips) * 32;
        c += nseg;
        // the previous segment's past-the-end pieces have landed and every wave is done with its rows
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        WR_ROW(oy0 - 1, 0)
        WR_ROW(oy0, 1)
#pragma unroll
        for (int j = 0; j < D; ++j) WR_STAGE(j, j, j + 2)
        int st = 0, s0 = 0;                                           // dY stage of chunk k (k % ND); ring slot of its top row (k % NR)
        for (int k = 0; k < nseg; ++k) {
            if (SD_W16_ABL != 1) { if (wave == 0) wait_vmcnt<3 * (D - 1)>(); else wait_vmcnt<2 * (D - 1)>(); }     // chunk k's pieces of this wave (stages k + 1 .. k + D - 1 may be in flight)
            if (SD_W16_ABL != 5) __builtin_amdgcn_s_barrier();
            const uint32_t ao = ao0 + (uint32_t)st * DSTG;
            const int s1 = s0 + 1 >= NR ? s0 + 1 - NR : s0 + 1, s2 = s1 + 1 >= NR ? s1 + 1 - NR : s1 + 1;
            const uint32_t r0 = (uint32_t)s0 * RROW, r1 = (uint32_t)s1 * RROW, r2 = (uint32_t)s2 * RROW;
            const uint32_t xa[9] = {xo0[0] + r0, xo0[1] + r0, xo0[2] + r0, xo0[0] + r1, xo0[1] + r1, xo0[2] + r1, xo0[0] + r2, xo0[1] + r2, xo0[2] + r2};
            TrPair a0, a1, b0, b1, b2, b3, b4, b5, b6, b7, b8;
            a0.lo = lds_tr16_async<0>(ao); a0.hi = lds_tr16_async<4 * 128>(ao);
            WR_TAP(0, 0, b0) WR_TAP(0, 1, b1) WR_TAP(0, 2, b2) WR_TAP(0, 3, b3) WR_TAP(0, 4, b4) WR_TAP(0, 5, b5) WR_TAP(0, 6, b6) WR_TAP(0, 7, b7) WR_TAP(0, 8, b8)
            if (SD_W16_ABL != 1) {            // (after the reads: a wave waits at an LDS-DMA instruction until the CU's queue takes it)
                const int st2 = st == 0 ? ND - 1 : st - 1;            // (k + D) % ND
                const int sl2 = s0 >= 1 ? s0 - 1 : NR - 1;            // (k + D + 2) % NR
                WR_STAGE(k + D, st2, sl2)
            }
            SD_W16_WAIT10(0, a0, b0, b1, b2, b3, b4, b5, b6, b7, b8);
            a1.lo = lds_tr16_async<16 * 128>(ao); a1.hi = lds_tr16_async<20 * 128>(ao);
            WR_MFMA(a0, b0, 0) WR_TAP(1, 0, b0)
            WR_MFMA(a0, b1, 1) WR_TAP(1, 1, b1)
            WR_MFMA(a0, b2, 2) WR_TAP(1, 2, b2)
            WR_MFMA(a0, b3, 3) WR_TAP(1, 3, b3)
            WR_MFMA(a0, b4, 4) WR_TAP(1, 4, b4)
            WR_MFMA(a0, b5, 5) WR_TAP(1, 5, b5)
            WR_MFMA(a0, b6, 6) WR_TAP(1, 6, b6)
            WR_MFMA(a0, b7, 7) WR_TAP(1, 7, b7)
            WR_MFMA(a0, b8, 8) WR_TAP(1, 8, b8)
            SD_W16_WAIT10(0, a1, b0, b1, b2, b3, b4, b5, b6, b7, b8);
            WR_MFMA(a1, b0, 0) WR_MFMA(a1, b1, 1) WR_MFMA(a1, b2, 2) WR_MFMA(a1, b3, 3) WR_MFMA(a1, b4, 4)
            WR_MFMA(a1, b5, 5) WR_MFMA(a1, b6, 6) WR_MFMA(a1, b7, 7) WR_MFMA(a1, b8, 8)
            st = st == ND - 1 ? 0 : st + 1;
            s0 = s1;
        }
    }
    wait_vmcnt<0>();
#undef WR_TAP
#undef WR_MFMA
#undef WR_STAGE
#undef WR_ROW
#undef WR_ROW_PIECE
    float* out = p.part + (int64_t)split * p.Nn * 9 * p.Ck;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = tn0 + wn0 + (e & 3) + 8 * (e >> 2) + 4 * fh, c = tc0 + wc0 + fr;
            if (SD_W16_ABL != 4 || acc[t][e] == 123.456f) out[((int64_t)n * 9 + t) * p.Ck + c] = acc[t][e];
        }
}

// ---------------------------------------------------------------------------------------------
// k_wgrad3x3_bf16_ring2 (round 4): the row-ring kernel as ONE 512-thread block per CU whose two groups of four waves work on the SAME
// (n, c) tile, each on its own half of the block's chunks with its own dY stages and patch ring, HALF A CHUNK APART -- the choreography of
// k_conv3x3_bf16_pp.  The ablations of k_wgrad3x3_bf16_ring showed that a wave's load phase (LDS-DMA issue, transposed reads) and its MFMA
// phase do not overlap (time = no-MFMA time + MFMA time): two independent blocks per CU drift into the same phase.  Here every chunk is
//   LOAD [the DMA of chunk k + 2, 20 transposed reads, lgkmcnt] - s_barrier - MFMA [18 MFMAs + the second k-block's 18 reads, vmcnt] - s_barrier
// and the block-wide barrier forces one group's LOAD under the other's MFMAs on the same SIMDs.  At the end group 1's accumulators are added
// to group 0's through LDS (three passes of three taps) and ONE partial tile is stored: half the partial-sum traffic of two blocks.
// Every phase ends with exactly one barrier; a group executes (segment prologue + 2 phases per chunk) per segment, the group with fewer
// phases pads with bare barriers -- the two barrier counts are equal for ANY split of the chunk range (no wave can be left waiting).
// Prefetch distance 2: past-the-end stages are not issued (nothing is in flight when a segment ends), the vmcnt count of a wait is picked
// by wave-uniform branches.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 1) void k_wgrad3x3_bf16_ring2(WgradArgs16 p) {
    constexpr int DSTG = 4096, RROW = 5120, ND = 3, NR = 5, RING0 = ND * DSTG, GRPB = ND * DSTG + NR * RROW;          // bytes: 37888 per group
    __shared__ __attribute__((aligned(16))) uint16_t W3[2 * GRPB / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave8 >> 2, wave = wave8 & 3;
    const int split = blockIdx.x;
    const int c_tiles = p.Ck >> 6;
    const int tn0 = ((int)blockIdx.y / c_tiles) * 64, tc0 = ((int)blockIdx.y % c_tiles) * 64;
    const int wn0 = (wave >> 1) * 32, wc0 = (wave & 1) * 32;
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    const uint16_t* const zero16 = reinterpret_cast<const uint16_t*>(g_zero_line);
    const int srow = lane >> 3;
    const int schunk = ((lane & 7) ^ ((srow & 3) << 1)) * 8;
    char* const lds_b = reinterpret_cast<char*>(W3) + grp * GRPB;

#define W2_ROW_PIECE(pc, iy, slot)                                                                                \
    {                                                                                                             \
        const int px = (pc) * 8 + srow, ix = ox0 - 1 + px;                                                        \
        const bool ok = px < 34 && (unsigned)(iy) < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;              \
        const uint16_t* src = ok ? p.x + (((int64_t)b * p.Hi + (iy)) * p.Wi + ix) * p.Ck + tc0 + schunk : zero16; \
        lds_dma16(src, reinterpret_cast<float*>(lds_b + RING0 + (slot) * RROW + (pc) * 1024));                    \
    }
#define W2_ROW(iy, slot) { W2_ROW_PIECE(wave, iy, slot) if (wave == 0) W2_ROW_PIECE(4, iy, slot) }
#define W2_STAGE(j, dst, slot)                                                                                    \
    {                                                                                                             \
        const int oy = oy0 + (j);                                                                                 \
        const uint16_t* src = p.dy + ((int64_t)(b * p.Ho + oy) * p.Wo + ox0 + wave * 8 + srow) * p.Nn + tn0 + schunk; \
        lds_dma16(src, reinterpret_cast<float*>(lds_b + (dst) * DSTG + wave * 1024));                             \
        W2_ROW(oy + 1, slot)                                                                                      \
    }
    // this wave's pieces are in, except those of the `n` youngest stages (n = 0 or 1)
#define W2_VMWAIT(n) { if ((n) == 0) wait_vmcnt<0>(); else if (wave == 0) wait_vmcnt<3>(); else wait_vmcnt<2>(); }
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const int a_slot = (wn0 >> 4) + (g & 1), b_slot = (wc0 >> 4) + (g & 1);
    const int khalf = (g >> 1) * 8 + q4;
    const uint32_t w3 = lds_addr(W3) + (uint32_t)grp * GRPB;
    const uint32_t ao0 = w3 + (uint32_t)(khalf * 128 + ((a_slot ^ (khalf & 3)) << 5) + pp * 8);
    uint32_t xo0[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) xo0[k] = w3 + RING0 + (uint32_t)((k + khalf) * 128 + ((b_slot ^ ((k + khalf) & 3)) << 5) + pp * 8);
#define W2_MFMA(A, BP, T) acc[T] = SD_MFMA_BF16(4, __builtin_bit_cast(bf16x8, A), __builtin_bit_cast(bf16x8, BP), acc[T]);
#define W2_TAP(KB, T, BP) { BP.lo = lds_tr16_async<(KB) * 16 * 128>(xa[T]); BP.hi = lds_tr16_async<(KB) * 16 * 128 + 4 * 128>(xa[T]); }

    // the block's chunk range, cut in two; barrier phases of a range = sum over its (image, strip) segments of (1 + 2 * chunks)
    const int c_beg = split * p.chunks_per_split, c_end = min(c_beg + p.chunks_per_split, p.chunks_total);
    const int c_mid = c_beg + (c_end - c_beg + 1) / 2;
    auto phases = [&](int cb, int ce) { int n = 0; for (int c = cb; c < ce;) { const int oy = c % p.Ho, ns = min(ce - c, p.Ho - oy); n += 1 + 2 * ns; c += ns; } return n; };
    const int ph0 = phases(c_beg, c_mid), ph1 = phases(c_mid, c_end);
    const int my_beg = grp ? c_mid : c_beg, my_end = grp ? c_end : c_mid;
    if (grp == 1) __builtin_amdgcn_s_barrier();                       // group 1 runs one phase behind
    for (int c = my_beg; c < my_end;) {
        const int unit = c / p.Ho, oy0 = c - unit * p.Ho, nseg = min(my_end - c, p.Ho - oy0);
        const int b = unit / p.strips, ox0 = (unit - b * p.strips) * 32;
        c += nseg;
        // ---- prologue phase (nothing of this group is in flight, every wave of the group is past its last reads)
        W2_ROW(oy0 - 1, 0)
        W2_ROW(oy0, 1)
        W2_STAGE(0, 0, 2)
        if (nseg > 1) W2_STAGE(1, 1, 3)
        W2_VMWAIT(nseg > 1 ? 1 : 0)
        __builtin_amdgcn_s_barrier();
        int st = 0, s0 = 0;
        for (int k = 0; k < nseg; ++k) {
            // ---- LOAD phase (the reads first: a wave waits at an LDS-DMA instruction until the CU's queue takes it)
            const uint32_t ao = ao0 + (uint32_t)st * DSTG;
            const int s1 = s0 + 1 >= NR ? s0 + 1 - NR : s0 + 1, s2 = s1 + 1 >= NR ? s1 + 1 - NR : s1 + 1;
            const uint32_t r0 = (uint32_t)s0 * RROW, r1 = (uint32_t)s1 * RROW, r2 = (uint32_t)s2 * RROW;
            const uint32_t xa[9] = {xo0[0] + r0, xo0[1] + r0, xo0[2] + r0, xo0[0] + r1, xo0[1] + r1, xo0[2] + r1, xo0[0] + r2, xo0[1] + r2, xo0[2] + r2};
            TrPair a0, a1, b0, b1, b2, b3, b4, b5, b6, b7, b8;
            a0.lo = lds_tr16_async<0>(ao); a0.hi = lds_tr16_async<4 * 128>(ao);
            W2_TAP(0, 0, b0) W2_TAP(0, 1, b1) W2_TAP(0, 2, b2) W2_TAP(0, 3, b3) W2_TAP(0, 4, b4) W2_TAP(0, 5, b5) W2_TAP(0, 6, b6) W2_TAP(0, 7, b7) W2_TAP(0, 8, b8)
            if (k + 2 < nseg) {
                const int st2 = st == 0 ? ND - 1 : st - 1, sl2 = s0 >= 1 ? s0 - 1 : NR - 1;
                W2_STAGE(k + 2, st2, sl2)
            }
            SD_W16_WAIT10(0, a0, b0, b1, b2, b3, b4, b5, b6, b7, b8);
            __builtin_amdgcn_s_barrier();
            // ---- MFMA phase
            __builtin_amdgcn_s_setprio(1);
            a1.lo = lds_tr16_async<16 * 128>(ao); a1.hi = lds_tr16_async<20 * 128>(ao);
            W2_MFMA(a0, b0, 0) W2_TAP(1, 0, b0)
            W2_MFMA(a0, b1, 1) W2_TAP(1, 1, b1)
            W2_MFMA(a0, b2, 2) W2_TAP(1, 2, b2)
            W2_MFMA(a0, b3, 3) W2_TAP(1, 3, b3)
            W2_MFMA(a0, b4, 4) W2_TAP(1, 4, b4)
            W2_MFMA(a0, b5, 5) W2_TAP(1, 5, b5)
            W2_MFMA(a0, b6, 6) W2_TAP(1, 6, b6)
            W2_MFMA(a0, b7, 7) W2_TAP(1, 7, b7)
            W2_MFMA(a0, b8, 8) W2_TAP(1, 8, b8)
            SD_W16_WAIT10(0, a1, b0, b1, b2, b3, b4, b5, b6, b7, b8);
            W2_MFMA(a1, b0, 0) W2_MFMA(a1, b1, 1) W2_MFMA(a1, b2, 2) W2_MFMA(a1, b3, 3) W2_MFMA(a1, b4, 4)
            W2_MFMA(a1, b5, 5) W2_MFMA(a1, b6, 6) W2_MFMA(a1, b7, 7) W2_MFMA(a1, b8, 8)
            __builtin_amdgcn_s_setprio(0);
            if (k + 1 < nseg) W2_VMWAIT(k + 2 < nseg ? 1 : 0)           // chunk k + 1's pieces of this wave
            __builtin_amdgcn_s_barrier();
            st = st == ND - 1 ? 0 : st + 1;
            s0 = s1;
        }
    }
    {   // equal barrier counts: pad to the longer group, then group 0 takes the barrier group 1 started with
        const int mine = grp ? ph1 : ph0, most = max(ph0, ph1);
        for (int i = mine; i < most; ++i) __builtin_amdgcn_s_barrier();
        if (grp == 0) __builtin_amdgcn_s_barrier();
    }
#undef W2_TAP
#undef W2_MFMA
#undef W2_VMWAIT
#undef W2_STAGE
#undef W2_ROW
#undef W2_ROW_PIECE
    // ---- group 1's sums into group 0: three taps per pass through LDS ([tap][element][thread]: conflict-free), fixed order
    float* const red = reinterpret_cast<float*>(W3);
    const int t4 = tid & 255;
#pragma unroll
    for (int ps = 0; ps < 3; ++ps) {
        __syncthreads();
        if (grp == 1) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) red[(t * 16 + e) * 256 + t4] = acc[ps * 3 + t][e];
        }
        __syncthreads();
        if (grp == 0) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[ps * 3 + t][e] += red[(t * 16 + e) * 256 + t4];
        }
    }
    if (grp == 0) {
        float* out = p.part + (int64_t)split * p.Nn * 9 * p.Ck;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = tn0 + wn0 + (e & 3) + 8 * (e >> 2) + 4 * fh, c = tc0 + wc0 + fr;
                out[((int64_t)n * 9 + t) * p.Ck + c] = acc[t][e];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16 weight gradient of the remaining geometries (strided 3x3, 1x1; mixed-precision training): ONE filter tap x 128 (n) x TC (c)
// tile per block, split over pixel ranges, on the bf16 MFMA with the transposed operand reads of k_wgrad3x3_bf16.  A chunk is 32
// consecutive output pixels: their dY rows (128 n = 256 B) and the input pixels the tap pairs them with (TC c; gathered per row by
// the DMA's per-lane source address, padding reads the zero line).
// LDS images: 256-byte rows keep their 32-byte slot t of row i at slot t ^ ((i & 3) << 1), 128-byte rows (TC = 64) at t ^ (i & 3):
// the four rows x two column halves of a transposed read then cover all 64 banks once.
// ---------------------------------------------------------------------------------------------
struct WgradArgs16t {
    const uint16_t* dy;   // [M][Nn] bf16
    const uint16_t* x;    // NHWC [B][Hi][Wi][Ck] bf16
    float* part;          // [splits][Nn][R*S][Ck] fp32
    int B, Hi, Wi, Ck, Ho, Wo, Nn, R, S, stride, pad;
    int M, splits, m_per_split;
};

template <int TC>
__global__ __launch_bounds__(256, 2) void k_wgrad_tap_bf16(WgradArgs16t p) {
    constexpr int TN = 128;
    constexpr int NTC = TC / 64;                                     // 32-column MFMA tiles per wave along c (wave tile 64 x TC/2)
    constexpr int XROWB = TC * 2;                                    // bytes per X row: 256 or 128
    constexpr int XPIECES = 32 * XROWB / 1024;                       // 8 or 4
    __shared__ __attribute__((aligned(16))) uint16_t Ds0[32 * TN];
    __shared__ __attribute__((aligned(16))) uint16_t Ds1[32 * TN];
    __shared__ __attribute__((aligned(16))) uint16_t Xs0[32 * TC];
    __shared__ __attribute__((aligned(16))) uint16_t Xs1[32 * TC];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.x;
    const int c_tiles = p.Ck / TC, n_tiles = p.Nn / TN;
    int t = blockIdx.y;
    const int ct = t % c_tiles; t /= c_tiles;
    const int nt = t % n_tiles; t /= n_tiles;
    const int tap = t, r = tap / p.S, s = tap - r * p.S;
    const int tn0 = nt * TN, tc0 = ct * TC;
    const int m_beg = split * p.m_per_split, m_end = min(m_beg + p.m_per_split, p.M);
    const int nchunks = (m_end - m_beg + 31) / 32;
    const int wn0 = (wave >> 1) * 64, wc0 = (wave & 1) * (TC / 2);
    const int fr = lane & 31, fh = lane >> 5;
    f32x16 acc[2][NTC];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const uint16_t* const zero16 = reinterpret_cast<const uint16_t*>(g_zero_line);

    // DMA lane geometry: 256-byte rows -> 4 rows per 1 KB piece (row lane / 16, chunk lane % 16, swizzle (row & 3) << 2 on the chunk);
    //                    128-byte rows -> 8 rows per piece (row lane / 8, chunk lane % 8, swizzle (row & 3) << 1)
    const int d_row = lane >> 4, d_chunk = ((lane & 15) ^ ((d_row & 3) << 2)) * 8;
    const int x_row = TC == 128 ? (lane >> 4) : (lane >> 3);
    const int x_chunk = TC == 128 ? ((lane & 15) ^ ((x_row & 3) << 2)) * 8 : ((lane & 7) ^ ((x_row & 3) << 1)) * 8;
    constexpr int XRPP = TC == 128 ? 4 : 8;                          // X rows per piece
#define WT_STAGE(ch, D, X)                                                                                        \
    {                                                                                                             \
        const int m0 = m_beg + (ch) * 32;                                                                         \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                 /* dY: 8 pieces of 4 rows, two per wave */ \
            const int q = wave + 4 * j, row = q * 4 + d_row;                                                      \
            const uint16_t* src = (m0 + row < m_end) ? p.dy + (int64_t)(m0 + row) * p.Nn + tn0 + d_chunk : zero16; \
            lds_dma16(src, reinterpret_cast<float*>((D) + q * 512));                                              \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < XPIECES / 4; ++j) {                                                 \
            const int q = wave + 4 * j, row = q * XRPP + x_row;                                                   \
            const int m = m0 + row;                                                                               \
            const int ox = m % p.Wo, t_ = m / p.Wo, oy = t_ % p.Ho, b = t_ / p.Ho;                                \
            const int iy = oy * p.stride - p.pad + r, ix = ox * p.stride - p.pad + s;                             \
            const bool ok = m < m_end && (unsigned)iy < (unsigned)p.Hi && (unsigned)ix < (unsigned)p.Wi;          \
            const uint16_t* src = ok ? p.x + (((int64_t)b * p.Hi + iy) * p.Wi + ix) * p.Ck + tc0 + x_chunk : zero16; \
            lds_dma16(src, reinterpret_cast<float*>((X) + q * 512));                                              \
        }                                                                                                         \
    }
    const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
    const int khalf = (g >> 1) * 8 + q4;
    auto rd_d = [&](const uint16_t* base, int row, int slot) -> v4i16 {          // 256-byte rows
        return lds_tr16(base + row * 128 + ((slot ^ ((row & 3) << 1)) << 4) + pp * 4);
    };
    auto rd_x = [&](const uint16_t* base, int row, int slot) -> v4i16 {
        if (TC == 128) return lds_tr16(base + row * 128 + ((slot ^ ((row & 3) << 1)) << 4) + pp * 4);
        return lds_tr16(base + row * 64 + ((slot ^ (row & 3)) << 4) + pp * 4);
    };
#define WT_COMPUTE(D, X)                                                                                          \
    {                                                                                                             \
        _Pragma("unroll") for (int kb = 0; kb < 2; ++kb) {                                                        \
            const int row = kb * 16 + khalf;                                                                      \
            bf16x8 av[2], bv[NTC];                                                                                \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                       \
                TrPair ap;                                                                                        \
                const int slot = (wn0 >> 4) + 2 * i + (g & 1);                                                    \
                ap.lo = rd_d((D), row, slot); ap.hi = rd_d((D), row + 4, slot);                                   \
                av[i] = __builtin_bit_cast(bf16x8, ap);                                                           \
            }                                                                                                     \
            _Pragma("unroll") for (int j = 0; j < NTC; ++j) {                                                     \
                TrPair bp;                                                                                        \
                const int slot = (wc0 >> 4) + 2 * j + (g & 1);                                                    \
                bp.lo = rd_x((X), row, slot); bp.hi = rd_x((X), row + 4, slot);                                   \
                bv[j] = __builtin_bit_cast(bf16x8, bp);                                                           \
            }                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                         \
                _Pragma("unroll") for (int j = 0; j < NTC; ++j)                                                   \
                    acc[i][j] = SD_MFMA_BF16(8, av[i], bv[j], acc[i][j]);                                       \
        }                                                                                                         \
    }
#define WT_ITER(CUR)                                                                                              \
    {                                                                                                             \
        if (ch + 1 < nchunks) { WT_STAGE(ch + 1, (CUR) ? Ds0 : Ds1, (CUR) ? Xs0 : Xs1) }                          \
        WT_COMPUTE((CUR) ? Ds1 : Ds0, (CUR) ? Xs1 : Xs0)                                                          \
        __syncthreads();                                                                                          \
        ++ch;                                                                                                     \
    }
    if (nchunks > 0) { WT_STAGE(0, Ds0, Xs0) }
    __syncthreads();
    int ch = 0;
    while (ch < nchunks) {
        WT_ITER(0)
        if (ch < nchunks) WT_ITER(1)
    }
#undef WT_ITER
#undef WT_COMPUTE
#undef WT_STAGE
    float* out = p.part + (int64_t)split * p.Nn * p.R * p.S * p.Ck;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTC; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = tn0 + wn0 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
                const int c = tc0 + wc0 + j * 32 + fr;
                out[((int64_t)n * p.R * p.S + tap) * p.Ck + c] = acc[i][j][e];
            }
}

constexpr int STEM_K = 147;   // 7 * 7 * 3

// ---------------------------------------------------------------------------------------------
// Stem (7x7 / stride 2 / pad 3, 3 -> 64 channels, NCHW image) with the input patch staged in LDS.
// One block = 128 consecutive output pixels of one output row: the 7 input rows x 261 columns x 3 channels
// it touches (22 KB) are loaded ONCE, coalesced, and every MFMA A-operand is read from that patch
// (address = row(ci, r) * 264 + 2 * px + s) -- no per-element global gathers.  k = (r*7 + s)*3 + ci.
// ---------------------------------------------------------------------------------------------
constexpr int SP_PITCH = 264, SP_ROWS = 21, SP_USED = 261;
constexpr int STEM_KPAD = 148;
constexpr size_t STEM_FWD_LDS_BYTES = (size_t)(21 * 264 + 148 * 64 + 4 * 1024 + 512) * sizeof(float);   // patch + weights + T + red = 78.5 KB
constexpr int STEM_KB = 160, STEM_WROW = 168;    // bf16 MFMA path: K padded to 10 steps of 16; LDS weight row of 168 bf16 (336 B, bank-spread)

struct StemArgs {
    const float* x;       // NCHW image
    const float* wt;      // [147][64] transposed weights (forward)
    const float* w;       // [64][147] weights as stored (forward on the bf16 MFMA)
    const float* dy;      // [M][64] (weight gradient)
    const uint16_t* dy16; // [M][64] bf16 (k_stem_wgrad_bf16_ring)
    float* y;             // [M][64] forward output / partial dW [blocks][64][147]
    const float* scale;
    const float* shift;
    int relu;
    int out_bf16;         // forward: store the NHWC output as bf16 (bf16 backbone)
    float* stat;          // forward: per-tile partial column sums [ntiles][2][64] of the raw output for the BatchNorm statistics (nullable)
    int B, H, W, Ho, Wo, tiles_x, ntiles;
    int rg;               // k_stem_wgrad_bf16_ring: output rows per unit
};

__device__ __forceinline__ int stem_koff(int k) {      // patch offset of reduction index k (without the pixel term)
    const int ci = k % 3, tap = k / 3, r = tap / 7, s = tap - r * 7;
    return (ci * 7 + r) * SP_PITCH + s;
}

// All global loads of a staging pass are issued before the first LDS store (fully unrolled, values in registers):
// a load-store-load-store loop would expose one global round trip per element.  Split in two halves so that a persistent block
// can have the NEXT tile's patch in flight (stem_fetch_patch) while the MFMAs of the current one run, and only store it
// (stem_commit_patch) once every wave is done with the current patch.
constexpr int SP_NLD = SP_ROWS + 1;
// Thread t fetches patch column t of all 21 (ci, r) rows (row addresses are wave-uniform: scalar arithmetic, one vector offset) and,
// for t < 105, one element of columns 256 .. 260.  Out-of-image rows / columns are fetched from the clamped coordinate and zeroed at
// commit time (no branch around a load, no per-element address registers carried across the tile loop).
__device__ __forceinline__ void stem_fetch_patch(const StemArgs& p, float (&v)[SP_NLD], int b, int oy, int ox0) {
    const int iy0 = 2 * oy - 3, ix0 = 2 * ox0 - 3, tid = threadIdx.x;
    const float* img = p.x + (int64_t)b * 3 * p.H * p.W;
    const int ixc = min(max(ix0 + tid, 0), p.W - 1);
#pragma unroll
    for (int row = 0; row < SP_ROWS; ++row) {
        const int ci = row / 7, r = row - ci * 7;
        const int iyc = min(max(iy0 + r, 0), p.H - 1);
        v[row] = img[(ci * p.H + iyc) * p.W + ixc];
    }
    const int e = min(tid, SP_ROWS * 5 - 1), row = e / 5, c = 256 + e - row * 5, ci = row / 7, r = row - ci * 7;
    v[SP_ROWS] = img[(ci * p.H + min(max(iy0 + r, 0), p.H - 1)) * p.W + min(max(ix0 + c, 0), p.W - 1)];
}
__device__ __forceinline__ void stem_commit_patch(const StemArgs& p, float* patch, const float (&v)[SP_NLD], int oy, int ox0) {
    const int iy0 = 2 * oy - 3, ix0 = 2 * ox0 - 3, tid = threadIdx.x;
    const bool cok = (unsigned)(ix0 + tid) < (unsigned)p.W;
#pragma unroll
    for (int row = 0; row < SP_ROWS; ++row) {
        const int r = row % 7;
        patch[row * SP_PITCH + tid] = (cok && (unsigned)(iy0 + r) < (unsigned)p.H) ? v[row] : 0.f;
    }
    if (tid < SP_ROWS * 5) {
        const int row = tid / 5, c = 256 + tid - row * 5, r = row % 7;
        patch[row * SP_PITCH + c] = ((unsigned)(ix0 + c) < (unsigned)p.W && (unsigned)(iy0 + r) < (unsigned)p.H) ? v[SP_ROWS] : 0.f;
    }
}
__device__ __forceinline__ void stem_fetch_dy(const StemArgs& p, f32x4 (&dv)[8], int64_t row0, int ox0) {     // 128 x 64 tile of dy
    // (f32x4, not HIP's float4: an array of that struct type carried across the tile loop stays in scratch memory)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int i = threadIdx.x + 256 * j, px = i >> 4, c4 = (i & 15) * 4;
        const float* src = (ox0 + px < p.Wo) ? p.dy + (row0 + px) * 64 + c4 : g_zero_line;
        dv[j] = *reinterpret_cast<const f32x4*>(src);
    }
}
// Tile `lin` of the persistent walk (lin = block + k * grid: its XCD is lin & 7) -> image tile.  Every XCD gets a contiguous range of
// tiles and, at any time, works on 64 consecutive ones (32 output rows): the five input rows that vertically neighbouring tiles share
// come from that XCD's L2 instead of HBM (PMC: 791 MB read per launch for a 201 MB image with the plain order).
__device__ __forceinline__ void stem_tile_coords(const StemArgs& p, int lin, int& b, int& oy, int& ox0) {
    const int tile = xcd_remap(lin, p.ntiles);
    const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x;
    oy = t2 % p.Ho; b = t2 / p.Ho; ox0 = tx * 128;
}

// BF16MM: the product runs on the bf16 MFMA (inference with the bf16 backbone): image patch and weights are rounded to bf16 on the
// way into the MFMA operands (eight reduction indices per lane and step), accumulation stays fp32 -- 20 MFMAs per wave instead of 148.
#ifdef SD_PP_TRACE
__device__ unsigned long long g_stem_trace[8][8];
#define ST_T(v) const unsigned long long v = __builtin_readcyclecounter();
#define ST_ACC(k, a, b) str[k] += (b) - (a);
#else
#define ST_T(v)
#define ST_ACC(k, a, b)
#endif
template <bool BF16MM>
__global__ __launch_bounds__(256, 2) void k_stem_fwd(StemArgs p) {
    // Persistent blocks (two per CU): the 7x7 weights are staged ONCE per block, then the block walks its 128-pixel tiles --
    // re-staging 37 KB of weights for every 32 KB of output was as expensive as the MFMAs.
    // LDS: [patch 21 x 264][weights][T: 4 waves x 16 rows x 64][red 4 x 2 x 64]
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;                              // [21][264]
    float* wl = lds + SP_ROWS * SP_PITCH;            // fp32: [148][64], row 147 = 0;  bf16: [64][STEM_WROW] bf16, k >= 147 zero
    float* Tall = wl + STEM_KPAD * 64;               // (the bf16 weight image is smaller than the fp32 one)
    float* red = Tall + 4 * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (BF16MM) {
        // weights [64][147] fp32 (original layout, k = (r*7 + s)*3 + ci) -> LDS [64][STEM_WROW] bf16
        uint16_t* wlb = reinterpret_cast<uint16_t*>(wl);
        constexpr int NE = 64 * STEM_KB / 256;                    // 40 elements per thread
        float wv[NE];
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int i = tid + 256 * j, n = i / STEM_KB, k = i - n * STEM_KB;
            wv[j] = (k < STEM_K) ? p.w[n * STEM_K + k] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < NE; ++j) {
            const int i = tid + 256 * j, n = i / STEM_KB, k = i - n * STEM_KB;
            wlb[n * STEM_WROW + k] = f2bf(wv[j]);
        }
    } else {
        constexpr int NW = (STEM_KPAD * 64 / 4 + 255) / 256;      // 10 float4 per thread
        float4 wv[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + 256 * j;
            const float* src = (i * 4 < STEM_K * 64) ? p.wt + i * 4 : g_zero_line;
            wv[j] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + 256 * j;
            if (i < STEM_KPAD * 64 / 4) reinterpret_cast<float4*>(wl)[i] = wv[j];
        }
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int px = wave * 32 + fr;
    const float* pa = patch + 2 * px;
    const float* pb = wl + fh * 64 + fr;
    const int c4 = (lane & 15) * 4;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + c4);
    if (p.shift) sh4 = *reinterpret_cast<const float4*>(p.shift + c4);
    float* T = Tall + wave * 1024;

    // the patch of tile t+1 is fetched into registers while tile t is multiplied: a tile never waits for a global round trip
    float pv[SP_NLD];
    if ((int)blockIdx.x < p.ntiles) {
        int b, oy, ox0;
        stem_tile_coords(p, blockIdx.x, b, oy, ox0);
        stem_fetch_patch(p, pv, b, oy, ox0);
    }
#ifdef SD_PP_TRACE
    unsigned long long str[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        ST_T(t0_)
        int b, oy, ox0;
        stem_tile_coords(p, tile, b, oy, ox0);
        stem_commit_patch(p, patch, pv, oy, ox0);
        ST_T(t1_)
        __syncthreads();                                 // patch (and, the first time, the weights) staged
        ST_T(t2_)
        if (tile + (int)gridDim.x < p.ntiles) {
            int nb, noy, nox0;
            stem_tile_coords(p, tile + gridDim.x, nb, noy, nox0);
            stem_fetch_patch(p, pv, nb, noy, nox0);
        }
        ST_T(t3_)
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        if (BF16MM) {
            const uint16_t* wb = reinterpret_cast<const uint16_t*>(wl) + fr * STEM_WROW + 8 * fh;
#pragma unroll
            for (int step = 0; step < STEM_KB / 16; ++step) {
                uint16_t ah[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {       // lane half fh supplies k = 16 step + 8 fh + j (indices past 146 meet zero weights)
                    const int k0 = 16 * step + j < STEM_K ? 16 * step + j : STEM_K - 1;
                    const int k1 = 16 * step + 8 + j < STEM_K ? 16 * step + 8 + j : STEM_K - 1;
                    ah[j] = f2bf(pa[fh ? stem_koff(k1) : stem_koff(k0)]);
                }
                const bf16x8 a = __builtin_bit_cast(bf16x8, ah);
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(wb + 16 * step);
                const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(wb + 32 * STEM_WROW + 16 * step);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc1, 0, 0, 0);
            }
        } else {
            // operands of step kk+1 are read while the MFMAs of step kk run
            float na = pa[fh ? stem_koff(1) : stem_koff(0)], nb0 = pb[0], nb1 = pb[32];
#pragma unroll
            for (int kk = 0; kk < STEM_KPAD / 2; ++kk) {
                const float a = na, b0 = nb0, b1 = nb1;
                if (kk + 1 < STEM_KPAD / 2) {
                    const int k0 = 2 * kk + 2, k1 = (2 * kk + 3 < STEM_K) ? 2 * kk + 3 : STEM_K - 1;   // k = 147 multiplies a zero weight row
                    na = pa[fh ? stem_koff(k1) : stem_koff(k0)];
                    nb0 = pb[(2 * kk + 2) * 64];
                    nb1 = pb[(2 * kk + 2) * 64 + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // 16-byte epilogue (see k_conv_igemm): the wave's 32 x 64 tile goes through its LDS region 16 rows at a time and comes back as rows
        ST_T(t4_)
        const int64_t row0 = ((int64_t)b * p.Ho + oy) * p.Wo;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * h + e8;                         // (e >> 2) in {2h, 2h+1}: rows 16h .. 16h+15
                const int rl = (e & 3) + 8 * ((e >> 2) - 2 * h) + 4 * fh;
                T[rl * 64 + fr] = acc0[e];
                T[rl * 64 + 32 + fr] = acc1[e];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 4 + (lane >> 4);
                const int ox = ox0 + wave * 32 + 16 * h + rl;
                if (ox >= p.Wo) continue;
                float4 v = *reinterpret_cast<const float4*>(T + rl * 64 + c4);
                v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
                if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                if (p.out_bf16) {
                    uint2 pk;
                    pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
                    pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
                    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (row0 + ox) * 64 + c4) = pk;
                } else {
                    *reinterpret_cast<float4*>(p.y + (row0 + ox) * 64 + c4) = v;
                }
            }
        }
        ST_T(t5_)
        if (p.stat) {
            // BatchNorm statistics of the raw conv output (tile pixels past the row end still see real image columns through the
            // 7-wide window, so they are masked): column sums per wave (32 pixels), the four waves combined through LDS in a fixed
            // order, one partial row per tile
            float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (ox0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh < p.Wo) {
                    // (bf16 output: the statistics of the ROUNDED values, i.e. of the tensor the BatchNorm normalises.  Compile-time for the
                    // fp32 instantiation: its instruction stream -- and with it the last bit of the statistics -- stays what it was)
                    const float a0 = (BF16MM && p.out_bf16) ? bf2f(f2bf(acc0[e])) : acc0[e], a1 = (BF16MM && p.out_bf16) ? bf2f(f2bf(acc1[e])) : acc1[e];
                    s0 += a0; q0 += a0 * a0; s1 += a1; q1 += a1 * a1;
                }
            }
            s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32); s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
            if (fh == 0) {
                red[(wave * 2 + 0) * 64 + fr] = s0; red[(wave * 2 + 0) * 64 + 32 + fr] = s1;
                red[(wave * 2 + 1) * 64 + fr] = q0; red[(wave * 2 + 1) * 64 + 32 + fr] = q1;
            }
            __syncthreads();
            if (tid < 128) {
                const int which = tid >> 6, n = tid & 63;
                const float v = (red[(0 * 2 + which) * 64 + n] + red[(1 * 2 + which) * 64 + n]) + (red[(2 * 2 + which) * 64 + n] + red[(3 * 2 + which) * 64 + n]);
                p.stat[(int64_t)tile * 128 + which * 64 + n] = v;
            }
        }
        ST_T(t6_)
        __syncthreads();                                 // every wave is done with the patch (and `red`) before the next tile's load
        ST_T(t7_)
        ST_ACC(0, t0_, t1_) ST_ACC(1, t1_, t2_) ST_ACC(2, t2_, t3_) ST_ACC(3, t3_, t4_) ST_ACC(4, t4_, t5_) ST_ACC(5, t5_, t6_) ST_ACC(6, t6_, t7_)
#ifdef SD_PP_TRACE
        str[7] += 1;
#endif
    }
#ifdef SD_PP_TRACE
    if (blockIdx.x == 8 && lane == 0) { for (int k = 0; k < 8; ++k) g_stem_trace[wave][k] = str[k]; }
#endif
}

// ---------------------------------------------------------------------------------------------
// fp32 stem forward, second form (round 3).  In-kernel clocks (tools/stem_trace_f32.py; `sd_set_option("stem_fwd_blocks", 256)` for
// one block per CU) of the two forms:
//   k_stem_fwd<false>, two blocks per CU: a tile = 25.1 k cycles, 10.7 k of them the MFMA loop (148 MFMAs = 9.47 k), the rest staging
//     the patch through registers (3.4 k + 3.1 k), epilogue 3.6 k, per-tile statistics 4.1 k;
//   this kernel, ONE block per CU: MFMA loop 9.54 k, patch wait 2.3 k, epilogue 1.4 k, statistics + DMA issue 1.5 k = 14.9 k;
//   this kernel, two blocks per CU: MFMA loop 9.54 k -- and the SAME epilogue 11.0 k: a wave streaming fp32 MFMAs leaves the other wave of
//     its SIMD next to no issue slots, so two resident blocks take turns (2 x (9.5 + 2.8) k per pair of tiles) instead of overlapping;
//     the second block only hides the patch round trip.  What counts is therefore the SOLO time of everything that is not an MFMA.
// Hence:
//   * the patch comes in by LDS-DMA: 22 pieces of 1 KB (patch column 0 = image column 2 ox0 - 4, so a group of four columns is 16-byte
//     aligned in the image and lies inside the row or outside), issued at the end of the tile (every wave is done with the patch: barrier
//     after the MFMA loop); per piece one compare pair and a select -- no staging registers, no LDS stores;
//   * the reduction runs in row-pair order (see the weight staging): the A address of every MFMA step is an immediate offset from one of
//     two per-lane bases.  In k = (r*7 + s)*3 + ci order hipcc hoisted the 74 per-lane offsets into registers, and what it spilled
//     instead were the DMA source addresses: a scratch reload + `s_waitcnt vmcnt(0)` in front of every DMA, 2 k cycles each;
//   * the BatchNorm column sums stay in registers for the whole block (fp32 sums of <= 64 tiles x 16 rows per lane, combined across
//     lanes / waves once): one partial row per BLOCK instead of per tile -- no per-tile masks (full tiles), barrier or store;
//   * patch, epilogue scratch and the weights are separate LDS objects: the compiler's wait insertion does not tie the epilogue's LDS
//     accesses to the DMA in flight.
// 840 -> 800 us per launch at bs = 64 (99 TFLOP/s).  Same 147 products per output as k_stem_fwd<false>, summed in another order
// (fp32 rounding only; the statistics likewise).
// ---------------------------------------------------------------------------------------------
constexpr int SPD_PIECES = 22, SPD_FLOATS = SPD_PIECES * 256, SPD_GROUPS = 66;     // [21][264] floats inside 22 KB

__device__ __forceinline__ int stem_koff1(int k) { return stem_koff(k) + 1; }      // (patch column 0 is one left of the 7-wide window)

template <bool STATS>
__global__ __launch_bounds__(256, 2) void k_stem_fwd_dma(StemArgs p) {
    __shared__ __attribute__((aligned(16))) float patch[SPD_FLOATS];
    __shared__ __attribute__((aligned(16))) float Tall[4 * 1024];
    __shared__ float red[4 * 2 * 64];
    extern __shared__ __attribute__((aligned(16))) float sd_stem_wl[];     // [148][64] transposed weights, row 147 = 0
    float* const wl = sd_stem_wl;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        // weights in MFMA step order: reduction index k' = 2 step + h (h = lane half).  Steps 0 .. 69: patch rows (2 rp + h) of row pair
        // rp = step / 7, column s = step % 7; steps 70 .. 73: the odd row 20, columns 2 j + h (column 7 is a phantom: zero weights).  The
        // two halves' operands are then a CONSTANT distance apart (one patch row, or one column): the A address of every step is an
        // immediate offset from one of two per-lane bases -- with k = (r*7 + s)*3 + ci order the distance changed from step to step, and
        // hipcc kept the 74 per-lane offsets in registers (spilling the DMA addresses instead).
        constexpr int NW = (STEM_KPAD * 64 / 4 + 255) / 256;      // 10 float4 per thread
        float4 wv[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + 256 * j, kp = i >> 4, n4 = (i & 15) * 4;
            const int step = kp >> 1, h = kp & 1;
            const int row = step < 70 ? 2 * (step / 7) + h : 20, sx = step < 70 ? step % 7 : 2 * (step - 70) + h;
            const int ci = row / 7, r = row - ci * 7;
            const float* src = (kp < STEM_KPAD && sx < 7) ? p.wt + ((r * 7 + sx) * 3 + ci) * 64 + n4 : g_zero_line;
            wv[j] = *reinterpret_cast<const float4*>(src);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int i = tid + 256 * j;
            if (i < STEM_KPAD * 64 / 4) reinterpret_cast<float4*>(wl)[i] = wv[j];
        }
    }
    // DMA pieces of this wave: q = wave + 4 j (j < 6; q < 22).  Lane constants: image offset of its 4-column group relative to the
    // patch origin, filter row and first column (a slot past row 20 never passes the row test).
    int poff[6], pr[6], pc[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int g = (wave + 4 * j) * 64 + lane, row = g / SPD_GROUPS, c4 = g - row * SPD_GROUPS, ci = row / 7, r = row - ci * 7;
        poff[j] = (ci * p.H + r) * p.W + 4 * c4;
        pr[j] = row < SP_ROWS ? r : (1 << 20);
        pc[j] = 4 * c4;
    }
#define SPD_ISSUE(tile_)                                                                                           \
    {                                                                                                              \
        int b_, oy_, ox0_;                                                                                         \
        stem_tile_coords(p, (tile_), b_, oy_, ox0_);                                                               \
        const int iy0_ = 2 * oy_ - 3, ix0_ = 2 * ox0_ - 4;                                                         \
        const float* const base_ = p.x + (int64_t)b_ * 3 * p.H * p.W + (int64_t)iy0_ * p.W + ix0_;                 \
        _Pragma("unroll") for (int j = 0; j < 6; ++j) {                                                            \
            if (wave + 4 * j < SPD_PIECES) {                                                                       \
                const bool ok_ = (unsigned)(iy0_ + pr[j]) < (unsigned)p.H && (unsigned)(ix0_ + pc[j]) < (unsigned)p.W; \
                lds_dma16(ok_ ? base_ + poff[j] : g_zero_line, patch + (wave + 4 * j) * 256);                      \
            }                                                                                                      \
        }                                                                                                          \
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int px = wave * 32 + fr;
    const float* pa = patch + 2 * px;
    const float* pb = wl + fh * 64 + fr;
    const int c4 = (lane & 15) * 4;
    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool affine = p.scale || p.shift || p.relu;
    if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + c4);
    if (p.shift) sh4 = *reinterpret_cast<const float4*>(p.shift + c4);
    float* T = Tall + wave * 1024;
    float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;            // BatchNorm column sums of this block's tiles (columns fr, 32 + fr)

    if ((int)blockIdx.x < p.ntiles) SPD_ISSUE(blockIdx.x)
#ifdef SD_PP_TRACE
    unsigned long long str[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        ST_T(t0_)
        int b, oy, ox0;
        stem_tile_coords(p, tile, b, oy, ox0);
        wait_vmcnt<0>();                                 // this wave's patch pieces (its youngest vector-memory operations) have landed
        ST_T(t1_)
        __syncthreads();                                 // every wave's pieces (and, the first time, the weights) are in LDS
        ST_T(t2_)
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        {
            // operands of step kk+1 are read while the MFMAs of step kk run (patch column 0 is one left of the 7-wide window: + 1)
            const float* pah = pa + fh * SP_PITCH + 1, * pac = pa + fh + 1;
            float na = pah[0], nb0 = pb[0], nb1 = pb[32];
#pragma unroll
            for (int kk = 0; kk < STEM_KPAD / 2; ++kk) {
                const float a = na, b0 = nb0, b1 = nb1;
                if (kk + 1 < STEM_KPAD / 2) {
                    constexpr int dummy = 0; (void)dummy;
                    const int k1 = kk + 1;
                    na = k1 < 70 ? pah[2 * (k1 / 7) * SP_PITCH + k1 % 7] : pac[20 * SP_PITCH + 2 * (k1 - 70)];
                    nb0 = pb[(2 * kk + 2) * 64];
                    nb1 = pb[(2 * kk + 2) * 64 + 32];
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        ST_T(t3_)
        __syncthreads();                                 // every wave is done with the patch (the next tile's pieces are issued after the epilogue)
        ST_T(t4_)
        ST_T(t5_)
        const bool full = ox0 + 128 <= p.Wo;
        // 16-byte epilogue (see k_conv_igemm): the wave's 32 x 64 tile goes through its LDS region 16 rows at a time and comes back as rows
        float* const ytile = p.y + ((((int64_t)b * p.Ho + oy) * p.Wo + ox0 + wave * 32 + (lane >> 4)) * 64 + c4);      // row (lane >> 4) of the wave's 32
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int e8 = 0; e8 < 8; ++e8) {
                const int e = 8 * h + e8;                         // (e >> 2) in {2h, 2h+1}: rows 16h .. 16h+15
                const int rl = (e & 3) + 8 * ((e >> 2) - 2 * h) + 4 * fh;
                T[rl * 64 + fr] = acc0[e];
                T[rl * 64 + 32 + fr] = acc1[e];
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rl = it * 4 + (lane >> 4);
                const int ox = ox0 + wave * 32 + 16 * h + rl;
                if (!full && ox >= p.Wo) continue;
                float4 v = *reinterpret_cast<const float4*>(T + rl * 64 + c4);
                if (affine) {
                    v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                }
                *reinterpret_cast<float4*>(ytile + (16 * h + it * 4) * 64) = v;
            }
        }
        ST_T(t6_)
        if (STATS) {
            if (full) {
#pragma unroll
                for (int e = 0; e < 16; ++e) { s0 += acc0[e]; q0 += acc0[e] * acc0[e]; s1 += acc1[e]; q1 += acc1[e] * acc1[e]; }
            } else {                                     // tile pixels past the row end still see real image columns through the window: masked
                const int lim = p.Wo - (ox0 + wave * 32 + 4 * fh);
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if ((e & 3) + 8 * (e >> 2) < lim) { s0 += acc0[e]; q0 += acc0[e] * acc0[e]; s1 += acc1[e]; q1 += acc1[e] * acc1[e]; }
            }
        }
        // The next tile's pieces go out BEHIND this tile's output stores: issued in front of them (right after the barrier) they kept the
        // stores waiting in the memory queue -- same tile time, measured both ways.  The round trip is covered by the other block's MFMAs.
        if (tile + (int)gridDim.x < p.ntiles) SPD_ISSUE(tile + gridDim.x)
        ST_T(t7_)
        ST_ACC(0, t0_, t1_) ST_ACC(1, t1_, t2_) ST_ACC(2, t2_, t3_) ST_ACC(3, t3_, t4_) ST_ACC(4, t4_, t5_) ST_ACC(5, t5_, t6_) ST_ACC(6, t6_, t7_)
#ifdef SD_PP_TRACE
        str[7] += 1;
#endif
    }
#ifdef SD_PP_TRACE
    if (blockIdx.x == 8 && lane == 0) { for (int kk = 0; kk < 8; ++kk) g_stem_trace[wave][kk] = str[kk]; }
#endif
#undef SPD_ISSUE
    if (STATS) {
        // one partial row per block: lane halves, then the four waves through LDS in a fixed order
        s0 += __shfl_xor(s0, 32); q0 += __shfl_xor(q0, 32); s1 += __shfl_xor(s1, 32); q1 += __shfl_xor(q1, 32);
        if (fh == 0) {
            red[(wave * 2 + 0) * 64 + fr] = s0; red[(wave * 2 + 0) * 64 + 32 + fr] = s1;
            red[(wave * 2 + 1) * 64 + fr] = q0; red[(wave * 2 + 1) * 64 + 32 + fr] = q1;
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, n = tid & 63;
            const float v = (red[(0 * 2 + which) * 64 + n] + red[(1 * 2 + which) * 64 + n]) + (red[(2 * 2 + which) * 64 + n] + red[(3 * 2 + which) * 64 + n]);
            p.stat[(int64_t)blockIdx.x * 128 + which * 64 + n] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Stem conv + folded BatchNorm + ReLU + 3x3 / stride 2 / pad 1 max-pool in ONE kernel (inference with the bf16 backbone;
// network.py:59-63 `adpater` = conv1, bn1, relu, maxpool): the full-resolution activation (B x H/2 x W/2 x 64: 537 MB in bf16 at
// bs=64 512x512, written by the conv and read back by the pool) never exists.  The pooling itself is exact with respect to the
// two-kernel form (sd_conv2d_stem_fwd with bf16 output, then sd_maxpool3x3s2_fwd_bf16): rounding to bf16 is monotonic, so the max
// of rounded values is the rounded max, and after the ReLU zero padding equals the pool's -inf padding; the conv sums its 147
// products in another order (k' below), i.e. differs from k_stem_fwd<true> by fp32 summation rounding (<= 1 bf16 ulp after rounding).
//   work unit : `rows` consecutive pooled rows q of one image; a persistent block walks its units.  Pooled row q needs conv rows
//               2q-1, 2q, 2q+1: the block keeps the horizontally pooled row 2q+1 as the "carry" of row q+1 (a unit's first row
//               recomputes conv row 2q-1 once).
//   conv row  : tiles of 128 pixels, left to right; patch rows are (ci, r) x 261 columns as bf16 (converted ONCE at staging),
//               reduction index k' = (ci*7 + r)*8 + 1 + s (slot 0 and row 21 meet zero weights; patch column 0 = image column
//               2 ox0 - 4, so that four-column groups are 16-byte aligned in the image): a lane's eight k' of a 32x32x16 MFMA
//               step are eight consecutive patch columns = 4 ds_read_b32, no conversion in the loop (k_stem_fwd<true>: 8 + 8 cvt)
//   pooling   : the wave tile goes to `rowbuf` (129 pixels x 64 channels bf16; pixel 0 = the previous tile's last pixel) after
//               scale / shift / ReLU / rounding; 16-byte items max over 3 pixels (v_pk_max_u16: non-negative bf16 order as
//               integers), then against the carry row; odd conv rows store the pooled row (16-byte coalesced stores).
// ---------------------------------------------------------------------------------------------
constexpr int SF_ROWS = 22, SF_PITCH = 320;        // bf16 patch rows of 640 bytes: the two lane halves (rows 2 step, 2 step + 1) hit disjoint banks
constexpr int SF_K = 176, SF_WROW = 184;           // weight image [64][184] bf16 (368-byte rows: conflict-free 16-byte reads)
typedef uint16_t u16x8 __attribute__((ext_vector_type(8)));
// LDS: [patch 22 x 320 bf16 = 14080 B][weights 64 x 184 bf16][rowbuf 129 x 64 bf16 = 16512 B][carry Wp x 64 bf16].  With the carry row of
// a 1024-wide input (Wp = 256: 32 KB) that is 86.9 KB: one block per CU (294 us at bs=16 1024x1024 against 211 us for the same pixels at
// 512x512).  alias = 1 puts the row tile ON the patch (dead once the tile's MFMAs have read it; one more barrier per tile, the left
// neighbour pixel kept in a 128-byte side buffer): [max(patch, rowbuf)][weights][halo][carry] = 72.9 KB at Wp = 256 -- two blocks per CU.
static size_t stem_pool_lds_bytes(int Wp, int alias = 0) {
    const size_t patch = (size_t)SF_ROWS * SF_PITCH * 2, rowbuf = 129 * 128;
    return (alias ? std::max(patch, rowbuf) + 128 : patch + rowbuf) + (size_t)64 * SF_WROW * 2 + (size_t)Wp * 128;
}

struct StemPoolArgs {
    const float* x;        // NCHW image
    const float* w;        // [64][147] stem weights as stored, k = (r*7 + s)*3 + ci
    const float* scale;    // folded BatchNorm
    const float* shift;
    uint16_t* y;           // [B][Hp][Wp][64] bf16
    int B, H, W, Ho, Wo, Hp, Wp, tiles_x, rows, units_per_img, nunits;
    int alias_rowbuf;      // wide inputs: the activated row tile reuses the patch's LDS (see stem_pool_lds_bytes)
};

__global__ __launch_bounds__(256, 2) void k_stem_pool_bf16(StemPoolArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const bool alias = p.alias_rowbuf != 0;
    uint16_t* patch = reinterpret_cast<uint16_t*>(lds);                 // [22][320]
    uint16_t* wl = patch + (alias ? 129 * 64 : SF_ROWS * SF_PITCH);     // [64][184]   (alias: behind max(patch, rowbuf) = 129 x 64)
    uint16_t* rowbuf = alias ? patch : wl + 64 * SF_WROW;               // [129][64]
    uint16_t* halo = wl + 64 * SF_WROW;                                 // alias: [64] left neighbour pixel of the next tile
    uint16_t* carry = alias ? halo + 64 : rowbuf + 129 * 64;            // [Wp][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < SF_ROWS * SF_PITCH / 2; i += 256) reinterpret_cast<uint32_t*>(patch)[i] = 0;   // pad columns / row 21 stay zero
    for (int i = tid; i < 64 * SF_WROW; i += 256) {
        const int n = i / SF_WROW, kk = i - n * SF_WROW, row = kk >> 3, sx = kk & 7, ci = row / 7, r = row - ci * 7;
        wl[i] = (kk < 168 && sx > 0) ? f2bf(p.w[n * STEM_K + (r * 7 + sx - 1) * 3 + ci]) : (uint16_t)0;     // slot 0 = the column left of the window
    }
    const int fr = lane & 31, fh = lane >> 5;
    const int px = wave * 32 + fr;
    const float sc0 = p.scale[fr], sc1 = p.scale[32 + fr], sh0 = p.shift[fr], sh1 = p.shift[32 + fr];
    const uint16_t* wb = wl + fr * SF_WROW + 8 * fh;

    // The tiles of a block form one sequence (unit -> conv row -> 128-pixel tile); the image patch of tile t+1 is fetched into
    // registers while tile t is multiplied and pooled, so that a tile never waits for a global round trip.
    struct Tile { int unit, t, nt, b, q0, oy0; bool valid; };
    auto open_unit = [&](int unit) {
        Tile it; it.unit = unit; it.t = 0; it.valid = unit < p.nunits;
        it.b = unit / p.units_per_img; it.q0 = (unit - it.b * p.units_per_img) * p.rows;
        const int q1 = min(it.q0 + p.rows, p.Hp);
        it.oy0 = max(2 * it.q0 - 1, 0);                    // (row -1 is padding: the carry starts as zeros instead)
        it.nt = (2 * q1 - it.oy0) * p.tiles_x;
        return it;
    };
    auto next_tile = [&](const Tile& c) { Tile n = c; if (++n.t >= n.nt) n = open_unit(c.unit + gridDim.x); return n; };
    // 21 rows x 66 groups of four columns; aligned 16-byte loads, 8-byte LDS stores
    constexpr int TOTAL4 = 21 * 66, NLD = (TOTAL4 + 255) / 256;
    float4 v[NLD];
    int g_r[NLD], g_c4[NLD], g_off[NLD], g_lds[NLD];       // per-thread constants: filter row, first column, image / patch offsets
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int i = tid + 256 * j;
        const int row = i / 66, c4 = (i - row * 66) * 4, ci = row / 7;
        g_r[j] = i < TOTAL4 ? row - ci * 7 : -(1 << 20);    // (past the patch: the row test below fails)
        g_c4[j] = c4; g_off[j] = (ci * p.H + g_r[j]) * p.W + c4; g_lds[j] = row * SF_PITCH + c4;
    }
    auto fetch = [&](const Tile& it) {
        const int oy = it.oy0 + it.t / p.tiles_x, ox0 = (it.t % p.tiles_x) * 128;
        const int iy0 = 2 * oy - 3, ix0 = 2 * ox0 - 4;
        const float* img = p.x + (int64_t)it.b * 3 * p.H * p.W + (int64_t)iy0 * p.W + ix0;
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int iy = iy0 + g_r[j], ix = ix0 + g_c4[j];
            // no branch around a load (hipcc would wait for every conditional load: one global round trip each): a group of four
            // columns is 16-byte aligned in the image (patch column 0 = image column 2 ox0 - 4), so it lies inside the row or outside
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            v[j] = *reinterpret_cast<const float4*>(ok ? img + g_off[j] : g_zero_line);
        }
    };
    Tile it = open_unit(blockIdx.x);
    if (it.valid) fetch(it);
#ifdef SD_PP_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    while (it.valid) {
        PP_T(s0_)
        const int row_i = it.t / p.tiles_x, tx = it.t - row_i * p.tiles_x, oy = it.oy0 + row_i, b = it.b;
        // conv row 2 q0 - 1: mode 0 (carry only); even rows: mode 1 (max into the carry); odd rows: mode 2 (store + new carry)
        const int mode = oy < 2 * it.q0 ? 0 : ((oy & 1) ? 2 : 1);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {                     // the patch of this tile: rounded to bf16 once
            uint2 pk;
            pk.x = (uint32_t)f2bf(v[j].x) | ((uint32_t)f2bf(v[j].y) << 16);
            pk.y = (uint32_t)f2bf(v[j].z) | ((uint32_t)f2bf(v[j].w) << 16);
            if (tid + 256 * j < TOTAL4) *reinterpret_cast<uint2*>(patch + g_lds[j]) = pk;
        }
        if (tx == 0 && tid < 8) reinterpret_cast<uint4*>(alias ? halo : rowbuf)[tid] = make_uint4(0, 0, 0, 0);  // left padding pixel
        if (it.t == 0 && it.q0 == 0) {                      // above the image: the carry is the padding row
            for (int i = tid; i < p.Wp * 8; i += 256) reinterpret_cast<uint4*>(carry)[i] = make_uint4(0, 0, 0, 0);
        }
        PP_T(s1_)
        const Tile nxt = next_tile(it);
        if (nxt.valid) fetch(nxt);
        PP_T(s2_)
        __syncthreads();                             // (1) patch staged (the first time: weights too)
        PP_T(s3_)
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        {   // operands of steps s+1 and s+2 are in flight while the MFMAs of step s run (an LDS round trip is ~2 steps long)
            constexpr int NS = SF_K / 16;
            const uint32_t* q = reinterpret_cast<const uint32_t*>(patch + fh * SF_PITCH + 2 * px);
            uint4 fa[NS]; bf16x8 fb0[NS], fb1[NS];
#pragma unroll
            for (int st = 0; st < 2; ++st) {
                const uint32_t* qn = q + st * SF_PITCH;
                fa[st] = make_uint4(qn[0], qn[1], qn[2], qn[3]);
                fb0[st] = *reinterpret_cast<const bf16x8*>(wb + 16 * st);
                fb1[st] = *reinterpret_cast<const bf16x8*>(wb + 32 * SF_WROW + 16 * st);
            }
#pragma unroll
            for (int step = 0; step < NS; ++step) {
                if (step + 2 < NS) {
                    const uint32_t* qn = q + (step + 2) * SF_PITCH;          // (uint32 units: two patch rows per step)
                    fa[step + 2] = make_uint4(qn[0], qn[1], qn[2], qn[3]);
                    fb0[step + 2] = *reinterpret_cast<const bf16x8*>(wb + 16 * (step + 2));
                    fb1[step + 2] = *reinterpret_cast<const bf16x8*>(wb + 32 * SF_WROW + 16 * (step + 2));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[step]), fb0[step], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[step]), fb1[step], acc1, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (alias) {
            __syncthreads();                         // (1b) every wave has read its operands: the row tile may overwrite the patch
            if (tid < 8) reinterpret_cast<uint4*>(rowbuf)[tid] = reinterpret_cast<const uint4*>(halo)[tid];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int pl = wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh;
            rowbuf[(1 + pl) * 64 + fr] = f2bf(fmaxf(acc0[e] * sc0 + sh0, 0.f));
            rowbuf[(1 + pl) * 64 + 32 + fr] = f2bf(fmaxf(acc1[e] * sc1 + sh1, 0.f));
        }
        PP_T(s4_)
        __syncthreads();                             // (2) the activated row tile is in rowbuf
        PP_T(s5_)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 256 * k, jl = item >> 3, cg = item & 7, j = 64 * tx + jl;
            if (j < p.Wp) {
                const u16x8* rb = reinterpret_cast<const u16x8*>(rowbuf) + (2 * jl) * 8 + cg;
                u16x8 hp = __builtin_elementwise_max(__builtin_elementwise_max(rb[0], rb[8]), rb[16]);
                u16x8* cp = reinterpret_cast<u16x8*>(carry) + j * 8 + cg;
                if (mode == 1) hp = __builtin_elementwise_max(hp, *cp);
                if (mode == 2) {
                    const u16x8 o = __builtin_elementwise_max(hp, *cp);
                    *reinterpret_cast<u16x8*>(p.y + ((((int64_t)b * p.Hp + (oy >> 1)) * p.Wp + j) * 64 + cg * 8)) = o;
                }
                *cp = hp;
            }
        }
        PP_T(s6_)
        __syncthreads();                             // (3) rowbuf read: keep its last pixel as the next tile's left neighbour
        if (tid < 8) reinterpret_cast<uint4*>(alias ? halo : rowbuf)[tid] = reinterpret_cast<const uint4*>(rowbuf)[128 * 8 + tid];
        it = nxt;
        PP_T(s7_)
        PP_ACC(0, s0_, s1_) PP_ACC(1, s1_, s2_) PP_ACC(2, s2_, s3_) PP_ACC(3, s3_, s4_) PP_ACC(4, s4_, s5_) PP_ACC(5, s5_, s6_) PP_ACC(6, s6_, s7_)
#ifdef SD_PP_TRACE
        tr[7] += 1;
#endif
    }
#ifdef SD_PP_TRACE
    if (blockIdx.x == 8 && lane == 0) { for (int k = 0; k < 8; ++k) g_pp_trace[wave][k] = tr[k]; }
#endif
}

// Weight gradient of the stem: persistent blocks walk the 128-pixel tiles, accumulating the whole 64 x 160 dW tile in
// registers (each wave takes 32 of the 128 pixels: 10 accumulators = 160 VGPRs), then the four waves are summed
// through LDS and the block writes ONE partial dW.
__global__ __launch_bounds__(256, 1) void k_stem_wgrad2(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;                              // [21][264]
    float* dys = lds + SP_ROWS * SP_PITCH;           // [128][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    int koff[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) {
        const int kc = kt * 32 + fr;
        koff[kt] = stem_koff(kc < STEM_K ? kc : STEM_K - 1);     // columns >= 147 are computed on valid data and dropped
    }
    f32x16 acc[2][5];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // patch and dy tile of tile t+1 are fetched into registers while tile t is multiplied (one block per CU: nothing else would
    // cover the global round trip of a staging pass)
    float pv[SP_NLD];
    f32x4 dv[8];
#define SW2_FETCH(tile_)                                                                                         \
    {                                                                                                            \
        int b_, oy_, ox0_;                                                                                       \
        stem_tile_coords(p, (tile_), b_, oy_, ox0_);                                                             \
        stem_fetch_patch(p, pv, b_, oy_, ox0_);                                                                  \
        stem_fetch_dy(p, dv, ((int64_t)b_ * p.Ho + oy_) * p.Wo + ox0_, ox0_);                                    \
    }
    if ((int)blockIdx.x < p.ntiles) SW2_FETCH(blockIdx.x)
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        __syncthreads();                                 // every wave is done with the previous tile's operands
        {
            int b, oy, ox0;
            stem_tile_coords(p, tile, b, oy, ox0);
            stem_commit_patch(p, patch, pv, oy, ox0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int i = tid + 256 * j, px = i >> 4, c4 = (i & 15) * 4;
            *reinterpret_cast<f32x4*>(dys + px * 64 + c4) = dv[j];
        }
        __syncthreads();
        if (tile + (int)gridDim.x < p.ntiles) SW2_FETCH(tile + gridDim.x)
#undef SW2_FETCH
        const float* da = dys + (wave * 32 + fh) * 64 + fr;
        const float* xb = patch + 2 * (wave * 32 + fh);
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            const float a0 = da[kk * 128], a1 = da[kk * 128 + 32];
            float bv[5];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) bv[kt] = xb[koff[kt] + 4 * kk];
#pragma unroll
            for (int kt = 0; kt < 5; ++kt) {
                acc[0][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[kt], acc[0][kt], 0, 0, 0);
                acc[1][kt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv[kt], acc[1][kt], 0, 0, 0);
            }
        }
    }
    // sum the four waves through LDS: R[n][160]
    __syncthreads();
    float* R = lds;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int n = i * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh, kc = j * 32 + fr;
                        float* dst = R + n * 160 + kc;
                        *dst = (w == 0 ? 0.f : *dst) + acc[i][j][e];
                    }
        }
        __syncthreads();
    }
    float* out = p.y + (int64_t)blockIdx.x * 64 * STEM_K;
    for (int i = tid; i < 64 * STEM_K; i += 256) {
        const int n = i / STEM_K, k = i - n * STEM_K;
        out[i] = R[n * 160 + k];
    }
}

// ---------------------------------------------------------------------------------------------
// Stem weight gradient on the bf16 MFMA (mixed-precision step; under autocast the reference's conv1 backward runs in bf16 too):
// dW[n][k] = sum over pixels of dy[px][n] * patch[px][k] -- the reduction index of the MFMA is the PIXEL, so each operand must present
// 8 consecutive pixels per lane:
//   dy    : the tile's 128 x 64 block is transposed on its way into LDS, DT[n][px] (bf16)
//   patch : element (px, k = (r, s, ci)) is image column 2 px + s of patch row (ci, r): for fixed k the pixels walk every second column.
//           The patch is split into its even / odd column planes, plane_p[row][i] = patch[row][2 i + p]; then (px, s) is
//           plane_{s & 1}[row][px + (s >> 1)] -- consecutive in px -- and FOUR copies of every plane, shifted by 0 .. 3 elements, make
//           each such 8-pixel run a 16-byte aligned ds_read_b128: copy sh holds plane[i + sh] at position i.
// 16-24 MFMAs (32x32x16) per wave and 128-pixel tile instead of 160 (32x32x2); persistent blocks keep the 64 x 160 dW tile in registers
// (2-3 of its ten 32 x 32 tiles per wave) and write one partial dW per block.
// ---------------------------------------------------------------------------------------------
constexpr int SW_RL = 136;                                    // row length (elements) of the plane copies and of DT: 272-byte rows
constexpr int SW_ROWS = 22;                                   // 21 (ci, r) rows + one zero row (k >= 147)
constexpr int SW_PLANES = 2 * 4 * SW_ROWS * SW_RL;            // elements
constexpr size_t SW_LDS_BYTES = (size_t)(SW_PLANES + 64 * SW_RL) * 2 > (size_t)64 * 160 * 4 ? (size_t)(SW_PLANES + 64 * SW_RL) * 2 : (size_t)64 * 160 * 4;

__global__ __launch_bounds__(256, 2) void k_stem_wgrad_bf16(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    uint16_t* planes = reinterpret_cast<uint16_t*>(lds);      // [p][sh][row][SW_RL]
    uint16_t* DT = planes + SW_PLANES;                        // [64][SW_RL]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    for (int i = tid; i < (SW_PLANES + 64 * SW_RL) / 2; i += 256) reinterpret_cast<uint32_t*>(planes)[i] = 0;     // zero row, pads
    // The ten 32 x 32 tiles of dW (2 along n, 5 along k) are dealt to the waves: wave w owns tiles w, w + 4, w + 8 (three for waves 0 / 1)
    // over ALL pixels -- 48 accumulator registers, no cross-wave sum at the end (a pixel split needs 160 and spilled).
    // B operand of k-tile kt: lane fr <-> k = 32 kt + fr = (r * 7 + s) * 3 + ci -> copy (s & 1, s >> 1), row ci * 7 + r
    int boff[3], aoff[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = wave + 4 * u, ni = (t < 10 ? t : 0) / 5, kt = (t < 10 ? t : 0) % 5;
        const int k = kt * 32 + fr;
        aoff[u] = (ni * 32 + fr) * SW_RL;
        if (k < STEM_K) {
            const int ci = k % 3, tap = k / 3, r = tap / 7, sx = tap - r * 7;
            boff[u] = ((((sx & 1) * 4 + (sx >> 1)) * SW_ROWS) + ci * 7 + r) * SW_RL;
        } else {
            boff[u] = (SW_ROWS - 1) * SW_RL;                   // the zero row of copy (0, 0)
        }
    }
    f32x16 acc[3];
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;

    // the operands of a tile: global -> registers -> LDS, the image patch first, then dy (two resident blocks per CU overlap each other's
    // round trips; fetching tile t + 1 under tile t's MFMAs was measured: 22 spilled registers, 20 % slower)
    constexpr int TOTAL = SP_ROWS * SP_PITCH, NLD = (TOTAL + 255) / 256;
    const int pg = tid >> 4, c4 = (tid & 15) * 4;              // dy: pixel group (8 pixels), first of 4 channels
    for (int tile = blockIdx.x; tile < p.ntiles; tile += gridDim.x) {
        const int tx = tile % p.tiles_x, t2 = tile / p.tiles_x, oy = t2 % p.Ho, b = t2 / p.Ho;
        const int ox0 = tx * 128;
        __syncthreads();                                       // the previous tile's operands have been read
        {   // image patch: 21 rows x 261 columns, coalesced along the columns; every element goes to its plane's four shifted copies
            const int iy0 = 2 * oy - 3, ix0 = 2 * ox0 - 3;
            const float* img = p.x + (int64_t)b * 3 * p.H * p.W;
            float v[NLD];
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int i = tid + 256 * j;
                const int row = i / SP_PITCH, col = i - row * SP_PITCH;
                const int ci = row / 7, r = row - ci * 7;
                const int iy = iy0 + r, ix = ix0 + col;
                const bool ok = i < TOTAL && col < SP_USED && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                v[j] = *(ok ? img + ((int64_t)ci * p.H + iy) * p.W + ix : g_zero_line);
            }
#pragma unroll
            for (int j = 0; j < NLD; ++j) {
                const int i = tid + 256 * j;
                const int row = i / SP_PITCH, col = i - row * SP_PITCH;
                if (i < TOTAL && col < SP_USED) {
                    const uint16_t h = f2bf(v[j]);
                    uint16_t* dst = planes + ((col & 1) * 4 * SW_ROWS + row) * SW_RL + (col >> 1);
#pragma unroll
                    for (int sh = 0; sh < 4; ++sh)
                        if ((col >> 1) >= sh) dst[sh * SW_ROWS * SW_RL - sh] = h;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        {   // dy tile, transposed: DT[n][px].  A thread takes 4 channels x 8 consecutive pixels (eight 16-byte loads, a wave-instruction reads
            // four 256-byte pixel rows) and stores one 16-byte run of 8 pixels per channel
            const int64_t row0 = ((int64_t)b * p.Ho + oy) * p.Wo + ox0;
            float4 dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int px = pg * 8 + j;
                const float* src = (ox0 + px < p.Wo) ? p.dy + (row0 + px) * 64 + c4 : g_zero_line;
                dv[j] = *reinterpret_cast<const float4*>(src);
            }
            uint4 o[4];
#define SW_PK(a, b) ((uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16))
            o[0] = make_uint4(SW_PK(dv[0].x, dv[1].x), SW_PK(dv[2].x, dv[3].x), SW_PK(dv[4].x, dv[5].x), SW_PK(dv[6].x, dv[7].x));
            o[1] = make_uint4(SW_PK(dv[0].y, dv[1].y), SW_PK(dv[2].y, dv[3].y), SW_PK(dv[4].y, dv[5].y), SW_PK(dv[6].y, dv[7].y));
            o[2] = make_uint4(SW_PK(dv[0].z, dv[1].z), SW_PK(dv[2].z, dv[3].z), SW_PK(dv[4].z, dv[5].z), SW_PK(dv[6].z, dv[7].z));
            o[3] = make_uint4(SW_PK(dv[0].w, dv[1].w), SW_PK(dv[2].w, dv[3].w), SW_PK(dv[4].w, dv[5].w), SW_PK(dv[6].w, dv[7].w));
#undef SW_PK
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<uint4*>(DT + (c4 + k) * SW_RL + pg * 8) = o[k];
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const int px0 = kk * 16 + fh * 8;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (u < 2 || wave < 2) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(DT + aoff[u] + px0);
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(planes + boff[u] + px0);
                    acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[u], 0, 0, 0);
                }
            }
        }
    }
    // every wave writes its tiles: R[n][160]
    __syncthreads();
    float* R = lds;
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int t = wave + 4 * u;
        if (t < 10) {
            const int ni = t / 5, kt = t % 5;
#pragma unroll
            for (int e = 0; e < 16; ++e) R[(ni * 32 + (e & 3) + 8 * (e >> 2) + 4 * fh) * 160 + kt * 32 + fr] = acc[u][e];
        }
    }
    __syncthreads();
    float* out = p.y + (int64_t)blockIdx.x * 64 * STEM_K;
    for (int i = tid; i < 64 * STEM_K; i += 256) {
        const int n = i / STEM_K, k = i - n * STEM_K;
        out[i] = R[n * 160 + k];
    }
}

// ---------------------------------------------------------------------------------------------
// The same gradient from a bf16 dy (mixed-precision step: the stem's BatchNorm backward stores bf16, as autocast does), row ring form.
// k_stem_wgrad_bf16 spent a 128-pixel tile's time (~9500 cycles per tile and CU for 640 cycles of MFMA) fetching and spreading its whole
// 21-row image patch (22 scalar loads and 88 two-byte LDS stores per thread) and transposing an fp32 dy tile through registers, with one
// exposed global round trip per tile.  Here
//   * a block walks DOWN a 128-pixel column strip (units of SR_RG output rows): consecutive tiles share five of their seven image rows
//     per channel, so a step brings only two new rows (x 3 channels); the plane copies are a ring of 8 image rows per channel (row iy
//     lives in slot iy & 7; the B operand offsets are recomputed per tile: 10 adds).  The three row pairs a unit's first tile needs
//     besides its own are three steps without a dy tile and without MFMAs;
//   * everything arrives by LDS-DMA, SR_D steps ahead, with counted waits (six instructions per wave and step, dummies included):
//     dy bf16 as stored ([pixel][64], transposed by ds_read_b64_tr_b16 on its way into the A operand) and the raw fp32 image rows
//     (aligned groups of four columns; converted and spread into the plane copies LDS -> LDS, 7 reads / 25 two-byte stores per thread).
//     One block per CU: 72 KB in flight, which is what ~2.5 us of latency needs at 5 TB/s (two blocks with two stages each: 231 us);
//   * v_mfma_f32_16x16x32_bf16: wave w owns channels 16 w .. 16 w + 15 and all ten 16-wide k tiles (40 accumulator registers, every wave
//     the same 40 MFMAs per tile; the 32x32 shape dealt ten tiles to four waves as 3 / 3 / 2 / 2).
// dy tile in LDS: pixel row i (128 B) keeps its 32-byte channel slot t at slot t ^ (i & 3) ^ ((i >> 3) & 1): the two 16-lane groups of a
// half-wave read the same slot of rows 8 apart, and the four rows of a transposed read are consecutive -- all on disjoint banks.
// ---------------------------------------------------------------------------------------------
#ifndef SD_SR_ABL
#define SD_SR_ABL 0            // timing experiments (WRONG RESULTS): 1 no ring commit, 2 no MFMA phase, 3 no DMA in the loop
#else
#define SD_SR_ABL_BUILD 1
#endif
constexpr int SR_RLB = 272;                                   // bytes per plane-copy row: 136 bf16 (131 used + shift pads), 17 16-byte chunks
constexpr int SR_CSTRIDE = (25 * 17 + 2) * 16;                // bytes between the copies of s and s + 1: 25 rows + 2 chunks, = 3 chunks mod 8 (see below)
constexpr int SR_PLANES0 = 16;                                // the first copy starts one chunk into LDS (stores of the first columns land up to 6 B in front of a row)
constexpr int SR_PLANES_B = SR_PLANES0 + 8 * SR_CSTRIDE;      // one copy per tap column s = 0 .. 6 (plane s & 1 shifted by s >> 1 elements) + a dummy eighth: 54672 B
constexpr int SR_NST = 4, SR_D = SR_NST - 1;                  // LDS stages; steps in flight
constexpr int SR_GPR = 66, SR_IMGROW = 4 * SR_GPR;            // 16-byte groups / floats per staged image row: columns 2 ox0 - 4 .. 2 ox0 + 259
constexpr int SR_STAGE = 128 * 64 * 2 + 8 * 64 * 16;          // bytes per stage: dy tile (16 KB) + 8 image DMA instructions (6 x 66 groups used)
constexpr size_t SR_LDS_BYTES = (size_t)SR_PLANES_B + (size_t)SR_NST * SR_STAGE;        // 54672 + 4 x 24576 = 152976 B: one block per CU
constexpr int SR_NDMA = 6;                                    // LDS-DMA instructions per wave and step

// Where the strip's steps are: (unit, step of the unit) -> image, strip, row pair; advanced incrementally (block-uniform scalars, the
// divisions once per unit).  A unit = p.rg output rows of one strip = p.rg + 3 steps (three row pairs without a tile first).
struct SrCursor {
    int sv, unit, b, ox0, oy_a, oy_b;
    __device__ __forceinline__ void decode(const StemArgs& p, int groups, int units) {
        if (unit < units) {
            const int rg = unit % groups, t2 = unit / groups, tx = t2 % p.tiles_x;
            b = t2 / p.tiles_x; ox0 = tx * 128; oy_a = rg * p.rg; oy_b = min(oy_a + p.rg, p.Ho);
        } else {
            b = 0; ox0 = 0; oy_a = 0; oy_b = -1000;            // past the end: neither rows nor a tile (dummy DMAs keep the counts)
        }
    }
    __device__ __forceinline__ void advance(const StemArgs& p, int groups, int units) {
        if (++sv == p.rg + 3) { sv = 0; unit += (int)gridDim.x; decode(p, groups, units); }
    }
    __device__ __forceinline__ int v() const { return oy_a - 3 + sv; }
    __device__ __forceinline__ bool has_rows() const { return v() < oy_b; }
    __device__ __forceinline__ bool has_tile() const { return v() < oy_b && v() >= oy_a; }
};

// Plane copies, bank layout: the B operand read of k tile kt is a ds_read_b128 whose 8-lane groups hold 8 CONSECUTIVE k = 3 (7 r + s) + ci
// (one 16-byte chunk each, same pixel offset).  Copy s starts 3 s chunks (mod 8) into the banks, and inside a copy the row of
// (channel ci, ring slot sigma) is 3 ((8 - sigma) & 7) + ci (17 chunks per row = 1 mod 8); with sigma = (iy0 + r) & 7 the chunk class
// of lane k is 3 (s - r) + ci + const = k + const (mod 8): eight consecutive classes, no bank conflict.  (First layout: rows ci * 8 + sigma,
// copies 25 rows apart -- the three channels of a tap fell on the same banks: the MFMA phase took 2740 cycles per tile for 640 of MFMA.)
// Waves split the tile's PIXELS: wave w multiplies pixels 32 w .. 32 w + 31 against all 64 x 160 outputs (8 transposed reads + 10 B reads =
// 14 KB of LDS per wave and tile for 40 MFMAs; a split of the outputs -- 2 x 5 tiles per wave -- read 28 KB and was LDS-bound), 160
// accumulator registers, the four partial dW tiles are summed once, at the end.
__global__ __launch_bounds__(256, 1) void k_stem_wgrad_bf16_ring(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* const stages = reinterpret_cast<char*>(lds) + SR_PLANES_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < (int)(SR_LDS_BYTES / 4); i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0;               // zero row, pads
    const uint16_t* const zero16 = reinterpret_cast<const uint16_t*>(g_zero_line);
    const int kcol = lane & 15, kq = lane >> 4;
    int bbase[10], brow[10];                                  // bytes (without the ring row); tap row r, or -1: the zero row (k >= 147)
#pragma unroll
    for (int kt = 0; kt < 10; ++kt) {
        const int k = kt * 16 + kcol;
        if (k < STEM_K) {
            const int ci = k % 3, tap = k / 3, r = tap / 7, sx = tap - r * 7;
            bbase[kt] = SR_PLANES0 + sx * SR_CSTRIDE + ci * SR_RLB + 16 * kq + 64 * wave;
            brow[kt] = r;
        } else {
            bbase[kt] = SR_PLANES0 + 24 * SR_RLB + 16 * kq;   // row 24 of copy 0 is never written
            brow[kt] = -1;
        }
    }
    // A operand (transposed read): 16-lane group kq addresses pixel rows 8 kq + q4 (+ 4 for the second read) of the wave's 32 pixels,
    // lane (q4, pp) the columns 4 pp .. 4 pp + 3 of 16-channel slot t (slot t of row i lives at t ^ (i & 3) ^ ((i >> 3) & 1))
    const int q4 = (lane >> 2) & 3, pp = lane & 3;
    int aoff[4];                                              // bytes
#pragma unroll
    for (int t = 0; t < 4; ++t) aoff[t] = (32 * wave + 8 * kq + q4) * 128 + ((t ^ q4 ^ (kq & 1)) << 5) + pp * 8;
    const int srow = lane >> 3;                               // dy DMA: this lane's row inside an 8-row piece
    // image DMA: instruction q = wave + 4 j covers the 16-byte groups g = 64 q + lane of the step's six rows (row6 = g / 66: channel row6 >> 1,
    // image row 2 v + 2 + (row6 & 1); column group g % 66)
    int irow[2], icol[2], ichan[2], dyoff[4];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int g = 64 * (wave + 4 * j) + lane;
        irow[j] = g < 6 * SR_GPR ? g / SR_GPR : 0;             // (the last instruction's spare lanes re-read row 0: their LDS slots are never read)
        icol[j] = 4 * (g % SR_GPR);
        ichan[j] = (irow[j] >> 1) * p.H * p.W;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {                              // element offset of this lane's 16 bytes inside the dy tile (source side of the slot swizzle)
        const int q = wave + 4 * j;
        dyoff[j] = (8 * q + srow) * 64 + ((lane & 7) ^ (((srow & 3) ^ (q & 1)) << 1)) * 8;
    }
    // commit: thread t converts column t of the six rows (+ one of the 30 elements of columns 256 .. 260 for t < 30)
    const int xrow = min(tid, 29) / 5, xcol = 256 + min(tid, 29) % 5;
    f32x4 acc[4][10];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int kt = 0; kt < 10; ++kt) acc[t][kt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int groups = (p.Ho + p.rg - 1) / p.rg;
    const int units = p.B * p.tiles_x * groups;
    const int my_units = (int)blockIdx.x < units ? (units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int nsteps = my_units * (p.rg + 3);
    // the step at cursor C_ -> stage ST_: exactly SR_NDMA instructions per wave, dummies where a step has no tile / no rows.  The image rows are
    // fetched from CLAMPED coordinates (always a valid address: scalar base + one 32-bit lane offset) and zeroed when they are committed.
#define SR_PREP(C_, ST_)                                                                                          \
    const int iv_ = C_.v(), ib_ = C_.b, iox0_ = C_.ox0;                                                           \
    const bool itile_ = C_.has_tile();                                                                            \
    char* const ist_ = stages + (ST_) * SR_STAGE;                                                                 \
    const uint16_t* const idy_ = p.dy16 + (((int64_t)ib_ * p.Ho + iv_) * p.Wo + iox0_) * 64;                      \
    const float* const iimg_ = p.x + (int64_t)ib_ * 3 * p.H * p.W;                                                \
    const int iy0_ = min(max(2 * iv_ + 2, 0), p.H - 1) * p.W, iy1_ = min(max(2 * iv_ + 3, 0), p.H - 1) * p.W;
#define SR_DMA_DY(j)                                                                                              \
    {                                                                                                             \
        const int q = wave + 4 * (j), px = 8 * q + srow;                                                          \
        const uint16_t* src = (itile_ && iox0_ + px < p.Wo) ? idy_ + dyoff[j] : zero16;                           \
        lds_dma16(src, reinterpret_cast<float*>(ist_ + q * 1024));                                                \
    }
#define SR_DMA_IMG(j)                                                                                             \
    {                                                                                                             \
        const int off = ichan[j] + ((irow[j] & 1) ? iy1_ : iy0_) + min(max(2 * iox0_ - 4 + icol[j], 0), p.W - 4); \
        lds_dma16(iimg_ + off, reinterpret_cast<float*>(ist_ + 16384 + (wave + 4 * (j)) * 1024));                 \
    }
#define SR_MFMA(T, KT) acc[T][KT] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, at[T]), __builtin_bit_cast(bf16x8, bb[KT]), acc[T][KT], 0, 0, 0);
#define SR_MFMA_ROW(KT) SR_MFMA(0, KT) SR_MFMA(1, KT) SR_MFMA(2, KT) SR_MFMA(3, KT)

    const uint32_t lds_a = lds_addr(lds), stages_a = lds_addr(stages);
    SrCursor ci_{0, (int)blockIdx.x, 0, 0, 0, 0}, cc_{0, (int)blockIdx.x, 0, 0, 0, 0};
    ci_.decode(p, groups, units); cc_.decode(p, groups, units);
    __syncthreads();                                           // LDS zeroed
    for (int s = 0; s < SR_D; ++s) { SR_PREP(ci_, s) SR_DMA_DY(0) SR_DMA_DY(1) SR_DMA_DY(2) SR_DMA_DY(3) SR_DMA_IMG(0) SR_DMA_IMG(1) ci_.advance(p, groups, units); }
    int stc = 0;                                               // stage of step s
#ifdef SD_PP_TRACE
    unsigned long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int s = 0; s < nsteps; ++s) {
        PP_T(t0_)
        // Every LDS access of the loop is inline asm: the compiler's wait insertion takes the LDS-DMAs in flight for possible aliases of
        // any LDS load or store it knows about and would drain them (s_waitcnt vmcnt(0)) once per step.
        wait_vmcnt_and_lds<(SR_D - 1) * SR_NDMA>();            // this wave's pieces of step s have landed (steps s + 1 .. s + SR_D - 1 stay in flight)
        __builtin_amdgcn_s_barrier();                          // (A) step s is complete; every wave is done with tile s - 1 (its stage, its two oldest ring rows)
        PP_T(t1_)
        SR_PREP(ci_, (stc + SR_D) & (SR_NST - 1))              // step s + SR_D goes into the stage of step s - 1 (between MFMAs it was slower: 240 vs 227 us)
        ci_.advance(p, groups, units);
        const int v = cc_.v(), cox0 = cc_.ox0;
        const bool has_rows = cc_.has_rows();
        cc_.advance(p, groups, units);
        const uint32_t st_a = stages_a + (uint32_t)stc * SR_STAGE;
        stc = (stc + 1) & (SR_NST - 1);
        // the raw image rows of this step (their LDS round trip runs under the DMA issue that follows)
        const uint32_t ra = st_a + 16384 + (uint32_t)(tid + 1) * 4, rx = st_a + 16384 + (uint32_t)(xrow * SR_IMGROW + xcol + 1) * 4;
        uint32_t raw[7];
        if (has_rows && SD_SR_ABL != 1) {
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(raw[0]) : "v"(ra), "n"(0 * SR_IMGROW * 4) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(raw[1]) : "v"(ra), "n"(1 * SR_IMGROW * 4) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(raw[2]) : "v"(ra), "n"(2 * SR_IMGROW * 4) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(raw[3]) : "v"(ra), "n"(3 * SR_IMGROW * 4) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(raw[4]) : "v"(ra), "n"(4 * SR_IMGROW * 4) : "memory");
            asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(raw[5]) : "v"(ra), "n"(5 * SR_IMGROW * 4) : "memory");
            asm volatile("ds_read_b32 %0, %1" : "=v"(raw[6]) : "v"(rx) : "memory");
        }
        if (SD_SR_ABL != 3) { SR_DMA_DY(0) SR_DMA_DY(1) SR_DMA_DY(2) SR_DMA_DY(3) SR_DMA_IMG(0) SR_DMA_IMG(1) }
        PP_T(t2_)
        if (has_rows && SD_SR_ABL != 1) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(raw[0]), "+v"(raw[1]), "+v"(raw[2]), "+v"(raw[3]), "+v"(raw[4]), "+v"(raw[5]), "+v"(raw[6]) :: "memory");
            // ring rows of this step: image rows 2 v + 2 (slot sg0) and 2 v + 3; element (row, col) -> plane o = col & 1, index col >> 1, written to
            // the copies s = o, o + 2, o + 4, o + 6 at position index - (s >> 1): immediates n * (2 * SR_CSTRIDE - 2) from the copy of s = o.  No
            // predicates: the first columns' shifted stores fall into the pad at the end of the row in front, odd columns' fourth into the dummy copy
            const int sg0 = (2 * v + 2 + 8) & 7;
#pragma unroll
            for (int r6 = 0; r6 < 7; ++r6) {
                const int row6 = r6 < 6 ? r6 : xrow, col = r6 < 6 ? tid : xcol;
                if (r6 < 6 || tid < 30) {
                    const bool in_img = (unsigned)(2 * v + 2 + (row6 & 1)) < (unsigned)p.H && (unsigned)(2 * cox0 - 3 + col) < (unsigned)p.W;
                    const uint32_t h = in_img ? f2bf(__builtin_bit_cast(float, raw[r6])) : 0u;
                    const int sg = (sg0 + (row6 & 1)) & 7, o = col & 1, ix = col >> 1;
                    const uint32_t dst = lds_a + (uint32_t)(SR_PLANES0 + o * SR_CSTRIDE + (3 * ((8 - sg) & 7) + (row6 >> 1)) * SR_RLB + ix * 2);
                    asm volatile("ds_write_b16 %0, %1" :: "v"(dst), "v"(h) : "memory");
                    asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(dst), "v"(h), "n"(1 * (2 * SR_CSTRIDE - 2)) : "memory");
                    asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(dst), "v"(h), "n"(2 * (2 * SR_CSTRIDE - 2)) : "memory");
                    asm volatile("ds_write_b16 %0, %1 offset:%2" :: "v"(dst), "v"(h), "n"(3 * (2 * SR_CSTRIDE - 2)) : "memory");
                }
            }
        }
        PP_T(t3_)
        wait_vmcnt_and_lds<SR_D * SR_NDMA>();                  // this wave's ring stores are done (nothing of the DMA queue is waited for)
        __builtin_amdgcn_s_barrier();                          // (B) the ring holds rows 2 v - 3 .. 2 v + 3
        PP_T(t4_)
        // every step multiplies: a step without a tile has the zero line in its dy stage (0 x finite ring rows), and accumulators that are
        // touched on every path stay in the accumulation registers (inside `if (has_tile)` the compiler kept 48 of them in VGPRs and copied
        // them in and out around the MFMAs: 340 moves per step)
        if (SD_SR_ABL != 2) {
            TrPair at[4];
            f32x4 bb[10];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t aa = st_a + (uint32_t)aoff[t];
                at[t].lo = lds_tr16_async<0>(aa); at[t].hi = lds_tr16_async<512>(aa);
            }
#pragma unroll
            for (int kt = 0; kt < 10; ++kt)
                bb[kt] = lds_read128_async<0>(lds_a + (uint32_t)(bbase[kt] + (brow[kt] >= 0 ? 3 * ((3 - 2 * v - brow[kt]) & 7) * SR_RLB : 0)));
            PP_T(t5_)
            asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(at[0].lo), "+v"(at[0].hi), "+v"(at[1].lo), "+v"(at[1].hi), "+v"(at[2].lo), "+v"(at[2].hi),
                         "+v"(at[3].lo), "+v"(at[3].hi), "+v"(bb[0]), "+v"(bb[1]), "+v"(bb[2]), "+v"(bb[3]) :: "memory");
            SR_MFMA_ROW(0) SR_MFMA_ROW(1) SR_MFMA_ROW(2) SR_MFMA_ROW(3)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bb[4]), "+v"(bb[5]), "+v"(bb[6]), "+v"(bb[7]), "+v"(bb[8]), "+v"(bb[9]) :: "memory");
            SR_MFMA_ROW(4) SR_MFMA_ROW(5) SR_MFMA_ROW(6) SR_MFMA_ROW(7) SR_MFMA_ROW(8) SR_MFMA_ROW(9)
            PP_T(t6_)
            PP_ACC(0, t0_, t1_) PP_ACC(1, t1_, t2_) PP_ACC(2, t2_, t3_) PP_ACC(3, t3_, t4_) PP_ACC(4, t4_, t5_) PP_ACC(5, t5_, t6_)
#ifdef SD_PP_TRACE
            tr[7] += 1;
#endif
        }
    }
#ifdef SD_PP_TRACE
    if (blockIdx.x == 8 && lane == 0) { for (int k = 0; k < 8; ++k) g_pp_trace[wave][k] = tr[k]; }
#endif
#undef SR_MFMA_ROW
#undef SR_MFMA
#undef SR_DMA_IMG
#undef SR_DMA_DY
#undef SR_PREP
    // the four waves' partial tiles are summed in wave order into R[n][160]
    wait_vmcnt<0>();
    __syncthreads();
    float* R = lds;
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int kt = 0; kt < 10; ++kt)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* r = R + (16 * t + 4 * kq + e) * 160 + kt * 16 + kcol;
                        *r = (w == 0 ? 0.f : *r) + acc[t][kt][e];
                    }
        }
        __syncthreads();
    }
    float* out = p.y + (int64_t)blockIdx.x * 64 * STEM_K;
    for (int i = tid; i < 64 * STEM_K; i += 256) {
        const int n = i / STEM_K, k = i - n * STEM_K;
        out[i] = R[n * 160 + k];
    }
}

// ---------------------------------------------------------------------------------------------
// Stem FORWARD of the mixed-precision step (bf16 NHWC output + BatchNorm statistics) on the same row ring: k_stem_fwd<true> re-staged the
// whole 21-row patch of every 128-pixel tile and built each MFMA operand from eight 4-byte LDS reads + conversions (376 us at bs = 64 for
// 79 us of MFMA work and a 140 us HBM floor).  Here the reduction index is ordered (ci, r, j) with j = tap column + 1 in 0 .. 7 (j = 0: zero
// weight), so the eight k of a lane are EIGHT CONSECUTIVE IMAGE COLUMNS 2 px .. 2 px + 7 of one (ci, r) row: one aligned ds_read_b128 from
// the copy px & 3 of the row, where copy c holds the bf16 row shifted by 2 c columns (copies start 2 c chunks (mod 8) into the banks: the
// eight lanes of a read group -- eight consecutive pixels -- hit eight chunk classes).  K = 24 rows x 8 (21 real rows) = 6 MFMA steps.
//   * D^T = W x patch^T: A = the weights (all in registers: 6 steps x 4 channel tiles), B = the patch; a lane then holds 16 consecutive
//     channels of ONE pixel (rows of channel tile t are channels 16 (n >> 2) + 4 t + (n & 3)): two 16-byte stores per lane and tile;
//   * steps, units, the image DMA and its staging as in k_stem_wgrad_bf16_ring (three stages, two steps in flight); the rows are committed
//     as packed pairs (even shifts: four ds_write_b32 per pair of columns);
//   * TWO WAVE GROUPS (512 threads, one block per CU): loads and stores share vmcnt and may retire out of order with each other, so a wave
//     that both prefetches by LDS-DMA with counted waits and stores its outputs has to wait for the stores' acknowledgement every step
//     (measured: 334 us; 659 with two blocks per CU, whose 256 registers spill).  Waves 0-3 issue the DMA, multiply and leave the tile's
//     packed bf16 rows in LDS (chunks swizzled by pixel & 7); waves 4-7 store the previous tile from there (64 contiguous bytes per thread),
//     sum the statistics of the rounded values (one partial row per block) and never wait for a store; both groups commit the new rows.
// ---------------------------------------------------------------------------------------------
#ifndef SD_SF_ABL
#define SD_SF_ABL 0            // timing experiment (WRONG RESULTS): 1 no global stores of the outputs
#else
#define SD_SF_ABL_BUILD 1
#endif
constexpr int SF_ROWB = 528;                                  // bytes per row copy: 256 bf16 + 8 (overflow of the shifted stores)
constexpr int SF_RING = 9;                                    // image rows per channel: the seven of a tile + the two the next step brings
constexpr int SF_CSTRIDE = 3 * SF_RING * SF_ROWB + 112;       // 3 channels x ring + 7 chunks: = 2 chunks mod 8
constexpr int SF_P0 = 16;
constexpr int SF_PLANES_B = SF_P0 + 4 * SF_CSTRIDE;           // 57488 B
constexpr int SF_NST = 3;                                    // stages of 8 KB (seven stages = six steps in flight: the same 232 us)
constexpr int SF_STAGE = 8 * 1024;                            // 8 image DMA instructions (6 rows x 66 groups of four columns used)
constexpr int SF_OBUF = 128 * 128;                            // a tile's bf16 outputs: 128 pixels x 64 channels
constexpr size_t SF_LDS_BYTES = (size_t)SF_PLANES_B + (size_t)SF_NST * SF_STAGE + 2 * SF_OBUF;       // 114832 B: one block of 8 waves per CU
constexpr int SF_NDMA = 2;
constexpr int SF_PAIRS = 6 * 132, SF_NPAIR = (SF_PAIRS + 255) / 256;      // column pairs of a step's six rows; per thread of the second group
static_assert((SF_CSTRIDE / 16) % 8 == 2, "copy c starts 2 c chunks (mod 8) into the banks");

// Schedule (one barrier per step; step s = tile row v of a strip, or one of the three row pairs in front of a unit's first tile):
//   waves 0-3 : DMA of step s + 3 -> stage (s mod 3) | operand reads + 48 MFMAs of tile s | packed bf16 rows -> obuf[s & 1]
//   waves 4-7 : commit of step s + 1 (stage -> ring rows 2 v + 4, 2 v + 5: with NINE ring rows they are not among the seven tile s reads)
//               | tile s - 1: obuf[(s - 1) & 1] -> global (64 contiguous bytes per thread) + statistics
// (with eight ring rows the commit had to sit between two barriers of its own step: 232 us)
__global__ __launch_bounds__(512, 1) void k_stem_fwd_bf16_ring(StemArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    char* const stages = reinterpret_cast<char*>(lds) + SF_PLANES_B;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2, cw = wave & 3;                 // group 0: DMA + MFMA waves; group 1: commit / output / statistics waves
    for (int i = tid; i < (int)(SF_LDS_BYTES / 4); i += 512) reinterpret_cast<uint32_t*>(lds)[i] = 0;
    const int n16 = lane & 15, kq = lane >> 4;
    const uint32_t lds_a = lds_addr(lds), stages_a = lds_addr(stages), obuf_a = stages_a + SF_NST * SF_STAGE;
    const int groups = (p.Ho + p.rg - 1) / p.rg;
    const int units = p.B * p.tiles_x * groups;
    const int my_units = (int)blockIdx.x < units ? (units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    const int nsteps = my_units * (p.rg + 3);
    __syncthreads();                                           // LDS zeroed
    if (grp == 0) {
        // ================= DMA + MFMA waves =================
        // A operand: weights of (k step ks, channel tile t): row n16 <-> channel 16 (n16 >> 2) + 4 t + (n16 & 3); k = row (4 ks + kq) = (ci, r), j = s + 1
        bf16x8 wreg[6][4];
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const int krow = 4 * ks + kq, ci = krow / 7, r = krow - ci * 7;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int ch = 16 * (n16 >> 2) + 4 * t + (n16 & 3);
                uint16_t h[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) h[j] = (krow < 21 && j >= 1) ? f2bf(p.w[((ch * 7 + r) * 7 + (j - 1)) * 3 + ci]) : (uint16_t)0;
                const uint4 pk = make_uint4(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16), h[4] | ((uint32_t)h[5] << 16), h[6] | ((uint32_t)h[7] << 16));
                wreg[ks][t] = __builtin_bit_cast(bf16x8, pk);
            }
        }
        // B operand: pixel px = 32 cw + 16 u + n16 of the tile -> copy px & 3 at byte 16 (px >> 2); row of k step ks = (ci, ring slot of r)
        uint32_t bpix[2], opix[2], opix1[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int px = 32 * cw + 16 * u + n16;
            bpix[u] = lds_a + SF_P0 + (px & 3) * SF_CSTRIDE + 16 * (px >> 2);
            opix[u] = obuf_a + px * 128 + (((2 * kq) ^ (px & 7)) << 4);        // this lane's 32 bytes: chunks 2 kq, 2 kq + 1, swizzled by px & 7
            opix1[u] = obuf_a + px * 128 + (((2 * kq + 1) ^ (px & 7)) << 4);
        }
        int kci[6], kr[6];
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            const int krow = min(4 * ks + kq, 20);             // rows 21 .. 23: zero weights, any row
            kci[ks] = (krow / 7) * SF_RING * SF_ROWB; kr[ks] = krow % 7;
        }
        int irow[2], icol[2], ichan[2];                        // image DMA (as k_stem_wgrad_bf16_ring)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int g = 64 * (cw + 4 * j) + lane;
            irow[j] = g < 6 * SR_GPR ? g / SR_GPR : 0;
            icol[j] = 4 * (g % SR_GPR);
            ichan[j] = (irow[j] >> 1) * p.H * p.W;
        }
        SrCursor ci_{0, (int)blockIdx.x, 0, 0, 0, 0}, cc_{0, (int)blockIdx.x, 0, 0, 0, 0};
        ci_.decode(p, groups, units); cc_.decode(p, groups, units);
#define SF_ISSUE(C_, ST_)                                                                                         \
        {                                                                                                         \
            const int iv_ = C_.v();                                                                               \
            const float* const iimg_ = p.x + (int64_t)C_.b * 3 * p.H * p.W;                                       \
            const int iy0_ = min(max(2 * iv_ + 2, 0), p.H - 1) * p.W, iy1_ = min(max(2 * iv_ + 3, 0), p.H - 1) * p.W; \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                       \
                const int off = ichan[j] + ((irow[j] & 1) ? iy1_ : iy0_) + min(max(2 * C_.ox0 - 4 + icol[j], 0), p.W - 4); \
                lds_dma16(iimg_ + off, reinterpret_cast<float*>(stages + (ST_) * SF_STAGE + (cw + 4 * j) * 1024)); \
            }                                                                                                     \
        }
        // steps 0, 1, 2 are in flight before the first barrier; at the barrier of iteration s (s = -1 .. nsteps) step s + 1 must have landed
        for (int s = 0; s < SF_NST; ++s) { SF_ISSUE(ci_, s) ci_.advance(p, groups, units); }
        for (int s = -1; s <= nsteps; ++s) {
            wait_vmcnt_and_lds<(SF_NST - 2) * SF_NDMA>();      // steps .. s + 1 have landed (s + 2 may be in flight); this wave's obuf stores are done
            __builtin_amdgcn_s_barrier();
            if (s < 0 || s >= nsteps) continue;
            SF_ISSUE(ci_, s % SF_NST)                          // step s + 3 into the stage of step s (committed during iteration s - 1)
            ci_.advance(p, groups, units);
            const int v = cc_.v();
            const bool has_tile = cc_.has_tile();
            cc_.advance(p, groups, units);
            if (has_tile) {
                f32x4 acc[2][4];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[u][t] = f32x4{0.f, 0.f, 0.f, 0.f};
                const int slot0 = (2 * v - 3 + 2 * SF_RING) % SF_RING;         // ring slot of image row 2 v - 3
                // operand reads two k steps ahead of their MFMAs, three rotating register pairs
                f32x4 b0[3], b1[3];
#define SF_READ(ks)                                                                                               \
                {                                                                                                 \
                    int sl_ = slot0 + kr[ks];                                                                     \
                    sl_ = sl_ >= SF_RING ? sl_ - SF_RING : sl_;                                                   \
                    const uint32_t ro = (uint32_t)(kci[ks] + sl_ * SF_ROWB);                                      \
                    b0[(ks) % 3] = lds_read128_async<0>(bpix[0] + ro);                                            \
                    b1[(ks) % 3] = lds_read128_async<0>(bpix[1] + ro);                                            \
                }
#define SF_KSTEP(ks, N)                                                                                           \
                asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(b0[(ks) % 3]), "+v"(b1[(ks) % 3]) :: "memory");    \
                _Pragma("unroll") for (int t = 0; t < 4; ++t) {                                                   \
                    acc[0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][t], __builtin_bit_cast(bf16x8, b0[(ks) % 3]), acc[0][t], 0, 0, 0); \
                    acc[1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[ks][t], __builtin_bit_cast(bf16x8, b1[(ks) % 3]), acc[1][t], 0, 0, 0); \
                }
                SF_READ(0) SF_READ(1)
                SF_KSTEP(0, 2) SF_READ(2)
                SF_KSTEP(1, 2) SF_READ(3)
                SF_KSTEP(2, 2) SF_READ(4)
                SF_KSTEP(3, 2) SF_READ(5)
                SF_KSTEP(4, 2)
                SF_KSTEP(5, 0)
#undef SF_KSTEP
#undef SF_READ
                // bf16 outputs -> obuf[s & 1] (the other group stores them and sums the statistics during the next step)
                const uint32_t ob = (uint32_t)(s & 1) * SF_OBUF;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint32_t pk[8];
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        pk[2 * t] = f2bf(acc[u][t][0]) | ((uint32_t)f2bf(acc[u][t][1]) << 16);
                        pk[2 * t + 1] = f2bf(acc[u][t][2]) | ((uint32_t)f2bf(acc[u][t][3]) << 16);
                    }
                    const f32x4 lo = __builtin_bit_cast(f32x4, make_uint4(pk[0], pk[1], pk[2], pk[3])), hi = __builtin_bit_cast(f32x4, make_uint4(pk[4], pk[5], pk[6], pk[7]));
                    const uint32_t oa = opix[u] + ob, oa2 = opix1[u] + ob;
                    asm volatile("ds_write_b128 %0, %1" :: "v"(oa), "v"(lo) : "memory");
                    asm volatile("ds_write_b128 %0, %1" :: "v"(oa2), "v"(hi) : "memory");
                }
            }
        }
#undef SF_ISSUE
        wait_vmcnt<0>();
    } else {
        // ================= commit / output / statistics waves =================
        // outputs: store i of a tile covers its bytes 4096 i .. + 4095, thread t2 the 16 bytes at 16 t2 of them (whole 128-byte lines per store
        // instruction; four 16-byte stores of ONE pixel per thread -- lines completed over four instructions -- ran at 2.3 TB/s): pixel 32 i + (t2 >> 3), channels 8 (t2 & 7) .. + 7
        const int t2 = tid - 256, opx0 = t2 >> 3, och = t2 & 7;
        int prow6[SF_NPAIR], pcol[SF_NPAIR];                      // commit: pair q = t2 + 256 j of the step's 6 x 132 column pairs
#pragma unroll
        for (int j = 0; j < SF_NPAIR; ++j) {
            const int q = t2 + 256 * j;
            prow6[j] = q < SF_PAIRS ? q / 132 : -1;
            pcol[j] = 2 * (q % 132);
        }
        float ssum[8], ssq[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) ssum[i] = ssq[i] = 0.f;
        SrCursor cn_{0, (int)blockIdx.x, 0, 0, 0, 0};             // the step being committed (s + 1)
        cn_.decode(p, groups, units);
        // tiles of step s (a*) and of step s - 1 (p*: its outputs are in obuf[(s - 1) & 1]) at the top of iteration s
        bool atile = false, ptile = false;
        int av = 0, aox = 0, ab = 0, pv = 0, pox = 0, pb = 0;
        for (int s = -1; s <= nsteps; ++s) {
            wait_vmcnt_and_lds<63>();                              // this wave's LDS traffic of the previous iteration is done; its global stores keep flying
            __builtin_amdgcn_s_barrier();
            bool ntile = false;
            int nv = 0, nox = 0, nb = 0;
            if (s + 1 < nsteps) {
                // rows of step s + 1 (tile row nv of strip nox) from their stage into the ring: packed pairs, copy c at position col - 2 c
                // (immediates c * (SF_CSTRIDE - 4); the first columns fall into the pad in front of the row)
                nv = cn_.v(); nox = cn_.ox0; nb = cn_.b; ntile = cn_.has_tile();
                const bool rows1 = cn_.has_rows();
                cn_.advance(p, groups, units);
                if (rows1) {
                    const uint32_t st_a = stages_a + (uint32_t)((s + 1) % SF_NST) * SF_STAGE;
                    uint2 raw[SF_NPAIR];
#pragma unroll
                    for (int j = 0; j < SF_NPAIR; ++j)
                        if (prow6[j] >= 0) asm volatile("ds_read_b64 %0, %1" : "=v"(raw[j]) : "v"(st_a + (uint32_t)(prow6[j] * SR_IMGROW + pcol[j]) * 4) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    const int sg0 = (2 * nv + 2 + 2 * SF_RING) % SF_RING, sg1 = sg0 + 1 == SF_RING ? 0 : sg0 + 1;
#pragma unroll
                    for (int j = 0; j < SF_NPAIR; ++j) {
                        if (prow6[j] >= 0) {
                            asm volatile("" : "+v"(raw[j]));
                            const int row6 = prow6[j], col = pcol[j], ix = 2 * nox - 4 + col;
                            const bool rowok = (unsigned)(2 * nv + 2 + (row6 & 1)) < (unsigned)p.H;
                            const uint32_t h0 = (rowok && (unsigned)ix < (unsigned)p.W) ? f2bf(__builtin_bit_cast(float, raw[j].x)) : 0u;
                            const uint32_t h1 = (rowok && (unsigned)(ix + 1) < (unsigned)p.W) ? f2bf(__builtin_bit_cast(float, raw[j].y)) : 0u;
                            const uint32_t pk = h0 | (h1 << 16);
                            const uint32_t dst = lds_a + (uint32_t)(SF_P0 + ((row6 >> 1) * SF_RING + ((row6 & 1) ? sg1 : sg0)) * SF_ROWB + col * 2);
                            asm volatile("ds_write_b32 %0, %1" :: "v"(dst), "v"(pk) : "memory");
                            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(dst), "v"(pk), "n"(1 * (SF_CSTRIDE - 4)) : "memory");
                            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(dst), "v"(pk), "n"(2 * (SF_CSTRIDE - 4)) : "memory");
                            asm volatile("ds_write_b32 %0, %1 offset:%2" :: "v"(dst), "v"(pk), "n"(3 * (SF_CSTRIDE - 4)) : "memory");
                        }
                    }
                }
            }
            if (ptile) {
                const uint32_t ob = obuf_a + (uint32_t)((s - 1) & 1) * SF_OBUF;
                f32x4 qf[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int px = 32 * i + opx0;
                    qf[i] = lds_read128_async<0>(ob + (uint32_t)(px * 128 + ((och ^ (px & 7)) << 4)));
                }
                SD_LDS_WAIT4(0, qf[0], qf[1], qf[2], qf[3]);
                uint4* const yrow = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(p.y) + (((int64_t)pb * p.Ho + pv) * p.Wo + pox) * 64) + t2;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (pox + 32 * i + opx0 < p.Wo) {
                        const uint4 q = __builtin_bit_cast(uint4, qf[i]);
                        if (SD_SF_ABL != 1) yrow[256 * i] = q;
                        const uint32_t wd[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float f0 = bf2f((uint16_t)(wd[k] & 0xffff)), f1 = bf2f((uint16_t)(wd[k] >> 16));
                            ssum[2 * k] += f0; ssq[2 * k] += f0 * f0;
                            ssum[2 * k + 1] += f1; ssq[2 * k + 1] += f1 * f1;
                        }
                    }
                }
            }
            ptile = atile; pv = av; pox = aox; pb = ab;
            atile = ntile; av = nv; aox = nox; ab = nb;
        }
        // statistics: the 8 lanes of a wave that share a channel group (lane & 7), then the four waves -- fixed order
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int o = 8; o < 64; o <<= 1) { ssum[i] += __shfl_xor(ssum[i], o); ssq[i] += __shfl_xor(ssq[i], o); }
        float* R = lds;                                            // [wave][2][64] (the ring is dead: every wave is past its last read)
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { R[cw * 128 + 8 * och + i] = ssum[i]; R[cw * 128 + 64 + 8 * och + i] = ssq[i]; }
        }
    }
    __syncthreads();
    const float* R = lds;
    if (tid < 128 && p.stat) p.stat[(int64_t)blockIdx.x * 128 + tid] = ((R[tid] + R[128 + tid]) + R[256 + tid]) + R[384 + tid];
}

// parallel split reduction: 32 float4 outputs (512 contiguous bytes of every partial copy) x 8 lanes over the copies per block, eight loads in
// flight per lane; fixed order (lane l adds copies l, l + 8, ... in turn, lane 0 adds the eight lane sums in turn): deterministic.
// (First form: 8 outputs x 32 lanes, two loads per lane at 64 splits: 11.5 us for 38 MB.)
__global__ __launch_bounds__(256) void k_wgrad_reduce_par(const float* __restrict__ part, float* __restrict__ dw, int64_t n4, int splits,
                                                           int accumulate) {
    __shared__ float4 red[8][32];
    const int lo = threadIdx.x & 31, lr = threadIdx.x >> 5;
    const int64_t i = (int64_t)blockIdx.x * 32 + lo;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        int k = lr;
        for (; k + 56 < splits; k += 64) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ld4(part, (int64_t)(k + 8 * u) * n4 + i);
#pragma unroll
            for (int u = 0; u < 8; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
        }
        for (; k < splits; k += 8) {
            const float4 v = ld4(part, (int64_t)k * n4 + i);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[lr][lo] = s;
    __syncthreads();
    if (lr != 0 || i >= n4) return;
    for (int k = 1; k < 8; ++k) { const float4 v = red[k][lo]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    if (accumulate) { const float4 v = reinterpret_cast<const float4*>(dw)[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
    reinterpret_cast<float4*>(dw)[i] = s;
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, float* __restrict__ dw, int64_t n4, int splits,
                                                       int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 s = accumulate ? reinterpret_cast<const float4*>(dw)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(part)[(int64_t)k * n4 + i];
        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    reinterpret_cast<float4*>(dw)[i] = s;
}

// [N][T][C] -> [C][T][N]  (forward weights -> data-gradient weights, flipping nothing: the
// dgrad coordinate map already walks the taps with rsign = -1)
// (WT = uint16_t: the transposed copy is written as bf16 -- mixed-precision data-gradient weights in one pass)
template <typename WT = float>
__global__ __launch_bounds__(256) void k_transpose_w(const float* __restrict__ w, WT* __restrict__ wt, int N, int T, int C) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const int c0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int j = ty; j < 32; j += 8) {
        const int n = n0 + j, c = c0 + tx;
        tile[j][tx] = (n < N && c < C) ? w[((int64_t)n * T + tap) * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, n = n0 + tx;
        if (n < N && c < C) {
            if constexpr (sizeof(WT) == 2) wt[((int64_t)c * T + tap) * N + n] = f2bf(tile[tx][j]);
            else wt[((int64_t)c * T + tap) * N + n] = tile[tx][j];
        }
    }
}

// every conv's weights in ONE launch (the training step transposes all 41 data-gradient weights once per backward: 42 launches of ~5 us
// were launch-bound).  table: 8 ints per conv {source offset, destination offset (elements), N, T, C, first block, blocks along C, blocks along N}
template <typename WT>
__global__ __launch_bounds__(256) void k_transpose_w_batched(const float* __restrict__ w, WT* __restrict__ wt, const int* __restrict__ table, int nconv) {
    __shared__ float tile[32][33];
    int ci = 0;
    while (ci + 1 < nconv && (int)blockIdx.x >= table[(ci + 1) * 8 + 5]) ++ci;          // (block-uniform: scalar loads)
    const int* t = table + ci * 8;
    const int N = t[2], T = t[3], C = t[4], nbx = t[6], nby = t[7];
    const float* src = w + t[0];
    WT* dst = wt + t[1];
    int b = blockIdx.x - t[5];
    const int bx = b % nbx; b /= nbx;
    const int by = b % nby, tap = b / nby;
    const int c0 = bx * 32, n0 = by * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int n = n0 + j, c = c0 + tx;
        tile[j][tx] = (n < N && c < C) ? src[((int64_t)n * T + tap) * C + c] : 0.f;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int c = c0 + j, n = n0 + tx;
        if (n < N && c < C) {
            if constexpr (sizeof(WT) == 2) dst[((int64_t)c * T + tap) * N + n] = f2bf(tile[tx][j]);
            else dst[((int64_t)c * T + tap) * N + n] = tile[tx][j];
        }
    }
}

// split-K second pass: y = epilogue( sum over K slices of the partial tiles )
template <bool BF16>
__global__ __launch_bounds__(256) void k_splitk_reduce(ConvArgs p) {
    const int64_t n4 = (int64_t)p.M * p.Nn / 4;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < p.splits; ++k) {
        const float4 v = reinterpret_cast<const float4*>(p.part)[(int64_t)k * n4 + i];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const int n = (int)((i * 4) % p.Nn);
    const int m = (int)((i * 4) / p.Nn);
    if (p.scale) { const float4 sc = *reinterpret_cast<const float4*>(p.scale + n); a.x *= sc.x; a.y *= sc.y; a.z *= sc.z; a.w *= sc.w; }
    if (p.shift) { const float4 sh = *reinterpret_cast<const float4*>(p.shift + n); a.x += sh.x; a.y += sh.y; a.z += sh.z; a.w += sh.w; }
    if (p.res) {
        int64_t rm = m;
        if (p.res_up2) {
            const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            rm = ((int64_t)b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
        }
        if (BF16) {
            const ushort4 r = *reinterpret_cast<const ushort4*>(reinterpret_cast<const uint16_t*>(p.res) + rm * p.Nn + n);
            a.x += bf2f(r.x); a.y += bf2f(r.y); a.z += bf2f(r.z); a.w += bf2f(r.w);
        } else {
            const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + rm * p.Nn + n);
            a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
        }
    }
    if (p.relu) { a.x = fmaxf(a.x, 0.f); a.y = fmaxf(a.y, 0.f); a.z = fmaxf(a.z, 0.f); a.w = fmaxf(a.w, 0.f); }
    if (BF16) reinterpret_cast<ushort4*>(p.y)[i] = make_ushort4(f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w));
    else reinterpret_cast<float4*>(p.y)[i] = a;
}

#ifndef SD_IGEMM_BIG64_S2
#define SD_IGEMM_BIG64_S2 0    // 256x64 tiles for the stride-2 data-gradients of 64-channel inputs (experiment)
#endif
#ifndef SD_IGEMM_BIG64
#define SD_IGEMM_BIG64 0       // 256x64 tiles for the 64-channel layers (measured slower than 128x64: off)
#endif
#ifndef SD_IGEMM_BIG
#define SD_IGEMM_BIG 1         // 1 = fp32 MODE 0 / 2 layers with enough 256-row tiles run k_conv_igemm_big
#endif

#ifndef SD_CONV_PATCH
#define SD_CONV_PATCH 1        // 1 = 3x3 / stride 1 / pad 1 convs (forward and data-gradient) run k_conv3x3_patch where its geometry fits
#endif

static thread_local int g_patch_bn64 = 0;            // 64-channel layers: the patch kernel wins the isolated layer benchmark (+4 %) but loses inside the
                                        // training step (754 vs 720 us per launch), so it is off; sd_set_option("conv_patch_bn64", 1)
static thread_local int g_patch_min_tiles = 512;     // two resident blocks per CU; sd_set_option("conv_patch_min_tiles", n) (tests: 1; off: 1 << 30)

// k_conv3x3_patch applies: unit-stride 3x3 with pad 1 (fwd: rsign +1, off -1; dgrad: rsign -1, off +1), map width 16..128 (power
// of two), images that are whole 256-pixel tiles, no split-K, and a grid that fills the chip.  Fills the geometry fields.
static bool conv_patch_geometry(ConvArgs& a, int BN, int mode, bool bf16 = false, bool allow64 = false) {
    // (64-channel tiles: on for bf16, where they gain 2 %; opt-in for fp32)
    if (!SD_CONV_PATCH || (BN == 64 && !g_patch_bn64 && !bf16 && !allow64) || mode != 0 || a.R != 3 || a.S != 3 || a.mul != 1 || a.div != 1 || a.splits > 1) return false;
    if (!((a.rsign == 1 && a.off == -1) || (a.rsign == -1 && a.off == 1))) return false;
    if (a.Ho != a.Hi || a.Wo != a.Wi || a.Ck % (bf16 ? 32 : BKB)) return false;
    int l2 = 0;
    while ((1 << l2) < a.Wo) ++l2;
    if ((1 << l2) != a.Wo || a.Wo < 16 || a.Wo > 128 || (a.Ho * a.Wo) % BMB) return false;
    if ((a.M / BMB) * (a.Nn / BN) < g_patch_min_tiles) return false;
    const int th = BMB / a.Wo;
    a.pt_tw_log2 = l2;
    a.pt_rolling = a.Wo == 128;
    a.pt_pw = a.pt_rolling ? 144 : a.Wo + 2;
    a.pt_pieces = a.pt_rolling ? 36 : cdiv((th + 2) * a.pt_pw, 16);
    a.pt_flip = a.rsign < 0;
    return a.pt_pieces * 256 <= (a.pt_rolling ? PT_FLOATS : PT_STAGE_FLOATS);
}

// Output-channel width of the patch-staging tile that takes this conv (0: none; fills the geometry fields).  128-channel tiles when
// they fill the chip; a layer whose 128-channel tiles do not (layer4 at bs=64: 256 tiles) takes 64-channel tiles if those do
// (sd_set_option("conv_patch_narrow", n): 0 off, 1 fp32 only, 2 fp32 and bf16).
static thread_local int g_patch_narrow = 2;            // 1: fp32 only, 2: bf16 too.  Same-box A/B (tools/ab_option.py): fp32 step -0.4 % (layer4: k_conv_igemm<128> -> k_conv3x3_patch<64>), bf16 eval forward -1.6 %, mixed-precision step -0.7 %
static int patch_tile_bn(ConvArgs& a, int BN, int mode, bool bf16) {
    ConvArgs t = a;
    if (conv_patch_geometry(t, BN, mode, bf16)) { a = t; return BN; }
    if (BN == 128 && (bf16 ? g_patch_narrow >= 2 : g_patch_narrow >= 1)) {
        t = a;
        if (conv_patch_geometry(t, 64, mode, bf16, true)) { a = t; return 64; }
    }
    return 0;
}

// k_conv3x3_c64_rows_bf16 applies: bf16, 64 -> 64 channels, unit-stride 3x3 with pad 1 (forward or flipped data-gradient), map width a
// multiple of 128, plain or same-size residual, no split-K, and enough (image, strip, row range) units to fill the chip.
static thread_local int g_rows64_min_units = 192;    // sd_set_option("conv_rows64_min_units", n) (tests: 1; off: 1 << 30)
static bool conv_rows64_geometry(const ConvArgs& a, int mode, RowsArgs& r) {
    if (mode != 0 || a.Ck != 64 || a.Nn != 64 || a.R != 3 || a.S != 3 || a.mul != 1 || a.div != 1 || a.splits > 1) return false;
    if (!((a.rsign == 1 && a.off == -1) || (a.rsign == -1 && a.off == 1))) return false;
    if (a.Ho != a.Hi || a.Wo != a.Wi || a.Wo % 128 || a.res_up2 || a.bn_x) return false;
    r = RowsArgs{};
    r.B = a.B; r.H = a.Ho; r.W = a.Wo; r.segs = a.Wo / 128;
    const int cols = r.B * r.segs;
    r.rows = std::min(r.H, std::max(8, cdiv(r.H * cols, 256)));        // ~256 units (one persistent block per CU), at least 8 rows each
    r.units_per_col = cdiv(r.H, r.rows);
    r.nunits = cols * r.units_per_col;
    // a unit pays for 288 weight registers and four input rows before its first MFMA: below 16 rows per unit (batches under ~32 at
    // 128 x 128) the tile kernels keep the layer (g_rows64_min_units = 1 lifts both limits: tests)
    if (r.nunits < g_rows64_min_units || (g_rows64_min_units > 1 && r.rows < 16)) return false;
    r.x = (const uint16_t*)a.x; r.w = (const uint16_t*)a.w; r.y = (uint16_t*)a.y; r.scale = a.scale; r.shift = a.shift;
    r.res = (const uint16_t*)a.res; r.stat = a.stat; r.relu = a.relu; r.flip = a.rsign < 0;
    return true;
}

// the swapped-operand form k_conv3x3_c64_rows16_bf16 takes every launch of the row stream except the training forward with fused BatchNorm
// statistics: with 32 registers of running sums beside 288 of weights hipcc's allocation of that instantiation (KIND 2) copies weight
// fragments into AGPRs inside the loop -- behind asm MFMAs that is a hazard nobody covers (tools/check_rows16_isa.py looks for it in every
// instantiation that IS dispatched); it gained 1.5-3 % where the others gain 5-11 %.  sd_set_option("conv_rows16", 0): off.
static thread_local int g_rows16 = 1;
static bool conv_rows16_args(const RowsArgs& r) { return g_rows16 && !r.stat; }

// k_conv3x3_c64_rows_f32 applies: fp32, 64 -> 64 channels, unit-stride 3x3 with pad 1 (forward or flipped data-gradient), map width a
// multiple of 64, plain or same-size residual, no fused BatchNorm-backward reduction, no split-K, and enough units to fill the chip.
static thread_local int g_rowsf32_min_units = 192;   // sd_set_option("conv_rows_f32_min_units", n) (tests: 1; off: 1 << 30)
static thread_local int g_stem_fwd_blocks = 512;      // persistent blocks of the fp32 stem forward (two per CU)
static bool conv_rowsf32_geometry(const ConvArgs& a, int mode, RowsArgsF& r) {
    if (mode != 0 || a.Ck != 64 || a.Nn != 64 || a.R != 3 || a.S != 3 || a.mul != 1 || a.div != 1 || a.splits > 1) return false;
    if (!((a.rsign == 1 && a.off == -1) || (a.rsign == -1 && a.off == 1))) return false;
    if (a.Ho != a.Hi || a.Wo != a.Wi || a.Wo % 64 || a.res_up2 || a.bn_x) return false;
    r = RowsArgsF{};
    r.B = a.B; r.H = a.Ho; r.W = a.Wo; r.segs = a.Wo / 64;
    const int cols = r.B * r.segs;
    r.rows = std::min(r.H, std::max(8, cdiv(r.H * cols, 256)));        // ~256 units (one persistent block per CU), at least 8 rows each
    r.units_per_col = cdiv(r.H, r.rows);
    r.nunits = cols * r.units_per_col;
    // a unit pays for 288 weight registers and four input rows before its first MFMA (g_rowsf32_min_units = 1 lifts both limits: tests)
    if (r.nunits < g_rowsf32_min_units || (g_rowsf32_min_units > 1 && r.rows < 16)) return false;
    r.x = (const float*)a.x; r.w = (const float*)a.w; r.y = (float*)a.y; r.scale = a.scale; r.shift = a.shift;
    r.res = (const float*)a.res; r.stat = a.stat; r.relu = a.relu; r.flip = a.rsign < 0;
    return true;
}

// k_conv3x3_bf16_pp applies: bf16, 128-channel output tiles, the double-buffered patch geometry (maps up to 64 pixels wide as whole
// rows, wider ones as 64-pixel column strips), whole 512-pixel tiles and a grid of at least g_pp_min_tiles blocks (one 512-thread block per CU).  Fills the geometry fields.
static thread_local int g_pp_min_tiles = 200;        // sd_set_option("conv_pp_min_tiles", n) (tests: 1; off: 1 << 30)
static thread_local int g_pp_strips = 1;             // sd_set_option("conv_pp_strips", 0): maps of 128 pixels and wider stay on k_conv3x3_patch (A/B)
static bool conv_pp_geometry(ConvArgs& a, int mode) {
    if (a.Nn % 128 || a.M % PP_BM || (a.M / PP_BM) * (a.Nn / 128) < g_pp_min_tiles || a.res_up2) return false;   // (a half-size residual map: k_conv3x3_patch)
    a.pt_strip_log2 = 0;
    int l2 = 0;
    while ((1 << l2) < a.Wo) ++l2;
    if (g_pp_strips && (1 << l2) == a.Wo && a.Wo >= 128 && a.Wo <= 4096 && a.Ho % 4 == 0) {
        // maps of 128 pixels and wider: a sub-tile is a 64-pixel column strip of four rows (patch 6 x 66 pixels, the geometry of a
        // 64-pixel-wide map) -- k_conv3x3_patch would take them with its single rolling buffer.  Same conditions as conv_patch_geometry.
        if (!SD_CONV_PATCH || mode != 0 || a.R != 3 || a.S != 3 || a.mul != 1 || a.div != 1 || a.splits > 1) return false;
        if (!((a.rsign == 1 && a.off == -1) || (a.rsign == -1 && a.off == 1))) return false;
        if (a.Ho != a.Hi || a.Wo != a.Wi || a.Ck % 32) return false;
        a.pt_tw_log2 = 6; a.pt_strip_log2 = l2 - 6; a.pt_rolling = 0; a.pt_pw = 66; a.pt_pieces = cdiv(6 * 66, 16); a.pt_flip = a.rsign < 0;
        return true;
    }
    const int keep = g_patch_min_tiles;
    g_patch_min_tiles = 1;
    const bool ok = conv_patch_geometry(a, 128, mode, true);
    g_patch_min_tiles = keep;
    return ok && !a.pt_rolling;
}

static thread_local int g_stem_fwd_ring = 1;         // sd_set_option("stem_fwd_ring", 0): the mixed-precision stem forward on k_stem_fwd<true> (A/B)
static thread_local int g_igemm_big_bf16 = 0;        // sd_set_option("igemm_big_bf16", 1): bf16 MODE 0 / 2 layers on the 256-row tiles too.  Measured neutral (bf16 forward
                                                     // +0.3 %, mixed-precision step +0.15 %): both kernels stage A once per TAP and are bound by the LDS-DMA path (24-32 KB per 512
                                                     // MFMA cycles = 48-64 B/clk of the CU's 64), not by how the chunks are pipelined
// two resident blocks per CU: take the 256-row tile when it still fills the chip once (512 blocks); measured: 256x64 tiles
// lose 4 % on the 64-channel layers, so only BN = 128
static int igemm_big_tiles(const ConvArgs& a, int BN, int mode) {
    if (!SD_IGEMM_BIG || (BN != 128 && !SD_IGEMM_BIG64 && !(SD_IGEMM_BIG64_S2 && mode == 2)) || (mode != 0 && mode != 2) || a.splits > 1) return 0;
    if (a.R * a.S * a.Ck < 512) return 0;     // short reductions (1x1 convs of <= 256 channels): the 3-stage pipeline never fills,
                                                // the 128-row tiles are 5-12 % faster
    const int m_per = mode == 2 ? a.M / 4 : a.M;
    const int big_tiles = (mode == 2 ? 4 : 1) * cdiv(m_per, BMB) * (a.Nn / BN);
    return (big_tiles >= 512 && (mode != 2 || m_per % BMB == 0)) ? big_tiles : 0;
}

template <int BN, int MODE, bool BF16 = false>
static void launch_one(const ConvArgs& a, int tiles, size_t lds, hipStream_t st) {
    (void)lds;                             // the tiles are static __shared__ objects (65 KB for BN = 128, 49 KB for BN = 64)
    if constexpr ((BN == 128 || (!BF16 && (SD_IGEMM_BIG64 || (SD_IGEMM_BIG64_S2 && MODE == 2)))) && (MODE == 0 || MODE == 2)) {
        if (const int big_tiles = (!BF16 || g_igemm_big_bf16) ? igemm_big_tiles(a, BN, MODE) : 0) {
            hipLaunchKernelGGL((k_conv_igemm_big<BN, MODE, BF16>), dim3(big_tiles), dim3(256), 0, st, a);
            return;
        }
    }
    hipLaunchKernelGGL((k_conv_igemm<BN, MODE, BF16>), dim3(tiles, a.splits > 1 ? a.splits : 1), dim3(256), 0, st, a);
}

static int launch_igemm(const ConvArgs& a, bool stem, hipStream_t st, bool bf16 = false) {
    const int BN = (a.Nn % 128 == 0) ? 128 : 64;
    const int tiles = cdiv(a.M, BM) * (a.Nn / BN);
    const size_t lds = (size_t)NBUF * (BM + BN) * LDK * sizeof(float) + BM * sizeof(int);
    const int mode = stem ? 1 : (a.par ? 2 : (a.div > 1 ? 3 : 0));
    if (!stem && bf16) {
        if (conv1x1_stream_geometry(a, mode) && conv1x1_stream_args(a)) {
            if (a.Ck == 64) { const int nt = a.M / 32; hipLaunchKernelGGL((k_conv1x1_stream_bf16<64, 2>), dim3(std::min(512, cdiv(nt, 4))), dim3(256), 0, st, a, nt); }
            else { const int nt = a.M / 16; hipLaunchKernelGGL((k_conv1x1_stream_bf16<128, 1>), dim3(std::min(512, cdiv(nt, 4))), dim3(256), 0, st, a, nt); }
            SD_LAUNCH_CHECK();
            return 0;
        }
        RowsArgs ra;
        if (conv_rows64_geometry(a, mode, ra)) {
            static thread_local bool raised64 = false;
            if (!raised64) {
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64_rows_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, RS_LDS_BYTES));
                raised64 = true;
            }
            if (conv_rows16_args(ra)) {
                if (int e = launch_rows16_bf16(ra, st)) return e;
            } else hipLaunchKernelGGL(k_conv3x3_c64_rows_bf16, dim3(std::min(ra.nunits, 256)), dim3(256), RS_LDS_BYTES, st, ra);
            SD_LAUNCH_CHECK();
            return 0;
        }
        ConvArgs pa = a;
        if (conv_pp_geometry(pa, mode)) {
            static thread_local bool raised = false;      // per host thread: cheap, idempotent
            if (!raised) {
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_bf16_pp), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           PP_LDS_FLOATS * (int)sizeof(float)));
                raised = true;
            }
            if (pa.head_y) {
                static thread_local bool raised_h = false;
                if (!raised_h) {
                    SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_bf16_pp_head), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               PP_LDS_FLOATS * (int)sizeof(float)));
                    raised_h = true;
                }
                hipLaunchKernelGGL(k_conv3x3_bf16_pp_head, dim3((pa.M / PP_BM) * (pa.Nn / 128)), dim3(512), PP_LDS_FLOATS * sizeof(float), st, pa);
            } else {
                hipLaunchKernelGGL(k_conv3x3_bf16_pp, dim3((pa.M / PP_BM) * (pa.Nn / 128)), dim3(512), PP_LDS_FLOATS * sizeof(float), st, pa);
            }
            SD_LAUNCH_CHECK();
            return 0;
        }
        SD_REQUIRE(!a.head_y, SD_ERR_INVALID, "sd_conv2d_fwd_bf16_head: this geometry does not take the two-group kernel (ask sd_conv2d_fwd_bf16_head_supported)");
    }
    if (!stem && !bf16) {
        RowsArgsF rf;
        if (conv_rowsf32_geometry(a, mode, rf)) {
            static thread_local bool raisedf = false;
            if (!raisedf) {
                SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64_rows_f32), hipFuncAttributeMaxDynamicSharedMemorySize, RF_LDS_BYTES));
                raisedf = true;
            }
            hipLaunchKernelGGL(k_conv3x3_c64_rows_f32, dim3(std::min(rf.nunits, 256)), dim3(256), RF_LDS_BYTES, st, rf);
            SD_LAUNCH_CHECK();
            return 0;
        }
    }
    if (!stem) {
        ConvArgs pa = a;
        if (const int PBN = patch_tile_bn(pa, BN, mode, bf16)) {
            const int pt_tiles = (pa.M / BMB) * (pa.Nn / PBN);
            if (bf16) {
                if (pa.pt_rolling) {
                    if (PBN == 128) hipLaunchKernelGGL((k_conv3x3_patch_roll<128, true>), dim3(pt_tiles), dim3(256), 0, st, pa);
                    else hipLaunchKernelGGL((k_conv3x3_patch_roll<64, true>), dim3(pt_tiles), dim3(256), 0, st, pa);
                } else if (PBN == 128) hipLaunchKernelGGL((k_conv3x3_patch<128, true>), dim3(pt_tiles), dim3(256), 0, st, pa);
                else hipLaunchKernelGGL((k_conv3x3_patch<64, true>), dim3(pt_tiles), dim3(256), 0, st, pa);
            } else {
                if (pa.pt_rolling) {
                    if (PBN == 128) hipLaunchKernelGGL((k_conv3x3_patch_roll<128, false>), dim3(pt_tiles), dim3(256), 0, st, pa);
                    else hipLaunchKernelGGL((k_conv3x3_patch_roll<64, false>), dim3(pt_tiles), dim3(256), 0, st, pa);
                } else if (PBN == 128) hipLaunchKernelGGL((k_conv3x3_patch<128, false>), dim3(pt_tiles), dim3(256), 0, st, pa);
                else hipLaunchKernelGGL((k_conv3x3_patch<64, false>), dim3(pt_tiles), dim3(256), 0, st, pa);
            }
            SD_LAUNCH_CHECK();
            return 0;
        }
    }
    if (bf16) {                            // forward (MODE 0) and, for mixed-precision training, the data-gradient modes
        if (BN == 128) {
            if (mode == 0) launch_one<128, 0, true>(a, tiles, lds, st);
            else if (mode == 2) launch_one<128, 2, true>(a, tiles, lds, st);
            else launch_one<128, 3, true>(a, tiles, lds, st);
        } else {
            if (mode == 0) launch_one<64, 0, true>(a, tiles, lds, st);
            else if (mode == 2) launch_one<64, 2, true>(a, tiles, lds, st);
            else launch_one<64, 3, true>(a, tiles, lds, st);
        }
    } else
    if (mode == 1) launch_one<64, 1>(a, tiles, lds, st);
    else if (BN == 128) {
        if (mode == 0) launch_one<128, 0>(a, tiles, lds, st);
        else if (mode == 2) launch_one<128, 2>(a, tiles, lds, st);
        else launch_one<128, 3>(a, tiles, lds, st);
    } else {
        if (mode == 0) launch_one<64, 0>(a, tiles, lds, st);
        else if (mode == 2) launch_one<64, 2>(a, tiles, lds, st);
        else launch_one<64, 3>(a, tiles, lds, st);
    }
    SD_LAUNCH_CHECK();
    if (a.splits > 1) {
        if (bf16) hipLaunchKernelGGL(k_splitk_reduce<true>, dim3(cdiv((int64_t)a.M * a.Nn / 4, 256)), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_splitk_reduce<false>, dim3(cdiv((int64_t)a.M * a.Nn / 4, 256)), dim3(256), 0, st, a);
        SD_LAUNCH_CHECK();
    }
    return 0;
}

// Split-K factor of the forward conv: only when the tile grid cannot fill the chip (small batch).
static thread_local int g_fwd_split_k = 1;           // sd_set_option("conv_fwd_split_k", 0): small grids keep the single-pass kernels (tests of those kernels)
static int fwd_splits(const sd_conv_desc* d, int ke = BK) {
    if (!g_fwd_split_k) return 1;
    const int M = d->B * d->Ho * d->Wo;
    const int BN = (d->Cout % 128 == 0) ? 128 : 64;
    const int tiles = cdiv(M, BM) * (d->Cout / BN);
    const int nk = d->R * d->S * (d->Cin / ke);
    if (tiles >= 256 || nk < 8) return 1;
    int s = std::min(cdiv(512, tiles), nk / 4);          // fill ~2 blocks per CU, keep >= 4 chunks per slice
    return std::max(1, std::min(s, 64));
}

// geometry part of the kernel arguments (shared by the launchers and sd_conv2d_kernel_name)
static void fill_fwd(ConvArgs& a, const sd_conv_desc* d) {
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.mul = d->stride; a.div = 1; a.off = -d->pad; a.rsign = 1;
    a.M = d->B * d->Ho * d->Wo; a.kchunks = d->Cin / BK; a.nk = d->R * d->S * a.kchunks;
    a.splits = fwd_splits(d);
}
static void fill_dgrad(ConvArgs& a, const sd_conv_desc* d) {
    a.B = d->B; a.Hi = d->Ho; a.Wi = d->Wo; a.Ck = d->Cout; a.Ho = d->Hi; a.Wo = d->Wi; a.Nn = d->Cin; a.R = d->R; a.S = d->S;
    a.mul = 1; a.div = d->stride; a.off = d->pad; a.rsign = -1;
    a.M = d->B * d->Hi * d->Wi; a.kchunks = d->Cout / BK; a.nk = d->R * d->S * a.kchunks;
    a.par = (d->stride == 2 && d->Hi % 2 == 0 && d->Wi % 2 == 0 && (a.M / 4) % BM == 0) ? 1 : 0;
}

static int check_conv(const char* what, const sd_conv_desc* d) {
    SD_REQUIRE(d != nullptr, SD_ERR_INVALID, "%s: null descriptor", what);
    SD_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->Cin > 0 && d->Cout > 0 && d->R > 0 && d->S > 0 && d->stride > 0 && d->pad >= 0,
               SD_ERR_INVALID, "%s: bad sizes", what);
    SD_REQUIRE(d->Ho == (d->Hi + 2 * d->pad - d->R) / d->stride + 1 && d->Wo == (d->Wi + 2 * d->pad - d->S) / d->stride + 1,
               SD_ERR_INVALID, "%s: Ho/Wo do not match the convolution geometry", what);
    SD_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->B * d->Hi * d->Wi < (1ll << 31), SD_ERR_INVALID,
               "%s: too many pixels for 32-bit pixel indices", what);
    return 0;
}

}  // namespace sd

using namespace sd;

extern "C" {

size_t sd_conv2d_fwd_workspace_bytes(const sd_conv_desc* d) {
    if (!d || d->Cin % 32 || d->Cout % 64) return 0;
    const int s = fwd_splits(d);
    return s > 1 ? (size_t)s * d->B * d->Ho * d->Wo * d->Cout * sizeof(float) : 0;
}

int sd_conv2d_fwd(const float* x, const float* w, float* y, const sd_conv_desc* d, const float* scale, const float* shift,
                  const float* residual, int res_up2, int relu, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_fwd", d)) return e;
    SD_REQUIRE(x && w && y, SD_ERR_INVALID, "sd_conv2d_fwd: null pointer");
    SD_REQUIRE(d->Cin % 32 == 0 && d->Cout % 64 == 0, SD_ERR_INVALID, "sd_conv2d_fwd: needs Cin %% 32 == 0 and Cout %% 64 == 0 (got %d, %d)",
               d->Cin, d->Cout);
    SD_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(residual), SD_ERR_ALIGN,
               "sd_conv2d_fwd: pointers must be 16-byte aligned");
    SD_REQUIRE(!res_up2 || (d->Ho % 2 == 0 && d->Wo % 2 == 0), SD_ERR_INVALID, "sd_conv2d_fwd: res_up2 needs even Ho, Wo");
    ConvArgs a{};
    fill_fwd(a, d);
    a.x = x; a.w = w; a.y = y; a.scale = scale; a.shift = shift; a.res = residual;
    a.relu = relu; a.res_up2 = res_up2;
    if (a.splits > 1) {
        // split-K needs its partial buffer; without one the single-pass kernel is still correct, just slower
        if (workspace && workspace_bytes >= sd_conv2d_fwd_workspace_bytes(d)) a.part = (float*)workspace;
        else a.splits = 1;
    }
    return launch_igemm(a, false, (hipStream_t)stream);
}

// rows of the statistics partial buffer the forward kernel of this geometry writes (0 = the split-K path: no fused statistics)
static int fwd_stat_rows(const sd_conv_desc* d, bool bf16 = false) {
    ConvArgs a{};
    fill_fwd(a, d);
    if (bf16) { a.kchunks = d->Cin / 64; a.nk = d->R * d->S * a.kchunks; a.splits = fwd_splits(d, 64); }
    if (a.splits > 1) return 0;
    const int BN = (a.Nn % 128 == 0) ? 128 : 64;
    ConvArgs t = a;
    RowsArgs ra;
    if (bf16 && conv_rows64_geometry(a, 0, ra)) return ra.nunits;
    RowsArgsF rf;
    if (!bf16 && conv_rowsf32_geometry(a, 0, rf)) return rf.nunits;
    if (bf16 && conv_pp_geometry(t, 0)) return a.M / PP_BM;
    t = a;
    return (patch_tile_bn(t, BN, 0, bf16) || ((!bf16 || (g_igemm_big_bf16 && BN == 128)) && igemm_big_tiles(a, BN, 0))) ? cdiv(a.M, BMB) : cdiv(a.M, BM);
}

size_t sd_conv2d_fwd_bn_stats_workspace_bytes(const sd_conv_desc* d) {
    if (!d || d->Cin % 32 || d->Cout % 64) return 0;
    const int rows = fwd_stat_rows(d);
    const size_t fused = (size_t)(rows + sd_bn_finalize_scratch_rows(rows)) * 2 * d->Cout * sizeof(float);
    const size_t split = sd_conv2d_fwd_workspace_bytes(d) + sd_col_reduce_workspace_bytes((int64_t)d->B * d->Ho * d->Wo, d->Cout);
    return std::max(fused, split);
}

int sd_conv2d_fwd_bn_stats(const float* x, const float* w, float* y, const sd_conv_desc* d, float eps, float momentum, float* running_mean,
                           float* running_var, float* mean, float* invstd, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_fwd_bn_stats", d)) return e;
    SD_REQUIRE(x && w && y && mean && invstd && workspace, SD_ERR_INVALID, "sd_conv2d_fwd_bn_stats: null pointer");
    SD_REQUIRE(d->Cin % 32 == 0 && d->Cout % 64 == 0, SD_ERR_INVALID, "sd_conv2d_fwd_bn_stats: needs Cin %% 32 == 0 and Cout %% 64 == 0 (got %d, %d)",
               d->Cin, d->Cout);
    SD_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), SD_ERR_ALIGN, "sd_conv2d_fwd_bn_stats: pointers must be 16-byte aligned");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_fwd_bn_stats_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_fwd_bn_stats: workspace too small");
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    const int rows = fwd_stat_rows(d);
    if (rows == 0) {        // small batch (split-K): plain conv, then the separate statistics pass
        const size_t cw = sd_conv2d_fwd_workspace_bytes(d);
        if (int e = sd_conv2d_fwd(x, w, y, d, nullptr, nullptr, nullptr, 0, 0, workspace, cw, stream)) return e;
        return sd_bn_train_stats(y, M, d->Cout, eps, momentum, running_mean, running_var, mean, invstd, (char*)workspace + cw,
                                 workspace_bytes - cw, stream);
    }
    ConvArgs a{};
    fill_fwd(a, d);
    a.x = x; a.w = w; a.y = y; a.stat = (float*)workspace;
    if (int e = launch_igemm(a, false, (hipStream_t)stream)) return e;
    return sd_bn_finalize_stats((const float*)workspace, rows, M, d->Cout, eps, momentum, running_mean, running_var, mean, invstd,
                                (float*)workspace + (size_t)rows * 2 * d->Cout, stream);
}

static void stem_args(StemArgs& a, const sd_conv_desc* d) {
    a.B = d->B; a.H = d->Hi; a.W = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo;
    a.tiles_x = cdiv(d->Wo, 128);
    a.ntiles = d->B * d->Ho * a.tiles_x;
}
static bool stem_is_7x7s2(const sd_conv_desc* d) { return d->Cin == 3 && d->Cout == 64 && d->R == 7 && d->S == 7 && d->stride == 2 && d->pad == 3; }
// k_stem_fwd_dma fetches the patch in aligned groups of four image columns: rows must be whole groups and 16-byte aligned
static bool stem_dma_ok(const sd_conv_desc* d, const void* x, const void* y) { return d->Wi % 4 == 0 && aligned16(x) && aligned16(y); }

size_t sd_conv2d_stem_fwd_workspace_bytes(const sd_conv_desc* d) { (void)d; return (size_t)STEM_K * 64 * sizeof(float); }

size_t sd_conv2d_fwd_bf16_workspace_bytes(const sd_conv_desc* d) {
    if (!d || d->Cin % 64 || d->Cout % 64) return 0;
    const int s = fwd_splits(d, 64);
    return s > 1 ? (size_t)s * d->B * d->Ho * d->Wo * d->Cout * sizeof(float) : 0;
}

int sd_conv2d_fwd_bf16(const void* x, const void* w, void* y, const sd_conv_desc* d, const float* scale, const float* shift,
                       const void* residual, int res_up2, int relu, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_fwd_bf16", d)) return e;
    SD_REQUIRE(x && w && y, SD_ERR_INVALID, "sd_conv2d_fwd_bf16: null pointer");
    SD_REQUIRE(d->Cin % 64 == 0 && d->Cout % 64 == 0, SD_ERR_INVALID, "sd_conv2d_fwd_bf16: needs Cin %% 64 == 0 and Cout %% 64 == 0 (got %d, %d)",
               d->Cin, d->Cout);
    SD_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(residual), SD_ERR_ALIGN,
               "sd_conv2d_fwd_bf16: pointers must be 16-byte aligned");
    SD_REQUIRE(!res_up2 || (d->Ho % 2 == 0 && d->Wo % 2 == 0), SD_ERR_INVALID, "sd_conv2d_fwd_bf16: res_up2 needs even Ho, Wo");
    ConvArgs a{};
    a.x = x; a.w = w; a.y = y; a.scale = scale; a.shift = shift; a.res = residual;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.mul = d->stride; a.div = 1; a.off = -d->pad; a.rsign = 1;
    a.relu = relu; a.res_up2 = res_up2;
    a.M = d->B * d->Ho * d->Wo; a.kchunks = d->Cin / 64; a.nk = d->R * d->S * a.kchunks;
    a.splits = fwd_splits(d, 64);
    if (a.splits > 1) {
        if (workspace && workspace_bytes >= sd_conv2d_fwd_bf16_workspace_bytes(d)) a.part = (float*)workspace;
        else a.splits = 1;
    }
    return launch_igemm(a, false, (hipStream_t)stream, true);
}

// ---- inference: the last FPN conv with the 1x1 head in its epilogue (network.py:17-18 + 22-29) ---------------------------------
__global__ __launch_bounds__(256) void k_head_split_bf16(const float* __restrict__ w, const float* __restrict__ bias, int co, uint16_t* __restrict__ hilo,
                                                         float* __restrict__ bias32) {
    for (int i = threadIdx.x; i < 32 * 128; i += 256) {
        const int n = i >> 7;
        const float wv = n < co ? w[i] : 0.f;
        const uint16_t hi = f2bf(wv);
        hilo[i] = hi;
        hilo[32 * 128 + i] = f2bf(wv - bf2f(hi));
    }
    if (threadIdx.x < 32) bias32[threadIdx.x] = (int)threadIdx.x < co ? bias[threadIdx.x] : 0.f;
}

int sd_conv2d_fwd_bf16_head_supported(const sd_conv_desc* d, int head_co) {
    if (!d || d->Cout != 128 || d->Cin % 64 || head_co < 1 || head_co > 32 || d->Wo % 4) return 0;
    ConvArgs a{};
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.mul = d->stride; a.div = 1; a.off = -d->pad; a.rsign = 1;
    a.M = d->B * d->Ho * d->Wo; a.kchunks = d->Cin / 64; a.nk = d->R * d->S * a.kchunks;
    a.splits = fwd_splits(d, 64);
    return (a.splits <= 1 && conv_pp_geometry(a, 0)) ? 1 : 0;
}

size_t sd_head_split_bf16_bytes(void) { return (size_t)2 * 32 * 128 * sizeof(uint16_t) + 32 * sizeof(float); }

int sd_head_split_bf16(const float* head_w, const float* head_bias, int head_co, void* prepared, sd_stream_t stream) {
    SD_REQUIRE(head_w && head_bias && prepared && head_co >= 1 && head_co <= 32, SD_ERR_INVALID, "sd_head_split_bf16: null pointer or head_co outside 1 .. 32");
    SD_REQUIRE(aligned16(prepared), SD_ERR_ALIGN, "sd_head_split_bf16: the prepared buffer must be 16-byte aligned");
    uint16_t* hilo = (uint16_t*)prepared;
    hipLaunchKernelGGL(k_head_split_bf16, dim3(1), dim3(256), 0, (hipStream_t)stream, head_w, head_bias, head_co, hilo, (float*)(hilo + 2 * 32 * 128));
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_conv2d_fwd_bf16_head(const void* x, const void* w, const sd_conv_desc* d, const float* scale, const float* shift, int relu,
                            const void* head_prepared, int head_co, float* head_y, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_fwd_bf16_head", d)) return e;
    SD_REQUIRE(x && w && head_prepared && head_y, SD_ERR_INVALID, "sd_conv2d_fwd_bf16_head: null pointer");
    SD_REQUIRE(sd_conv2d_fwd_bf16_head_supported(d, head_co), SD_ERR_INVALID,
               "sd_conv2d_fwd_bf16_head: needs a 3x3 / stride 1 conv onto 128 channels that takes k_conv3x3_bf16_pp and 1 <= head_co <= 32");
    SD_REQUIRE(aligned16(x) && aligned16(w) && aligned16(scale) && aligned16(shift) && aligned16(head_prepared) && aligned16(head_y), SD_ERR_ALIGN,
               "sd_conv2d_fwd_bf16_head: pointers must be 16-byte aligned");
    ConvArgs a{};
    a.x = x; a.w = w; a.y = nullptr; a.scale = scale; a.shift = shift;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.mul = d->stride; a.div = 1; a.off = -d->pad; a.rsign = 1;
    a.relu = relu;
    a.M = d->B * d->Ho * d->Wo; a.kchunks = d->Cin / 64; a.nk = d->R * d->S * a.kchunks;
    a.splits = 1;
    a.head_w = (const uint16_t*)head_prepared; a.head_b = (const float*)((const uint16_t*)head_prepared + 2 * 32 * 128);
    a.head_y = head_y; a.head_co = head_co;
    return launch_igemm(a, false, (hipStream_t)stream, true);
}

// ---- mixed-precision training (bf16 activations and weights, fp32 accumulation; trainer.py:115-121 autocast) -----------------
size_t sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(const sd_conv_desc* d) {
    if (!d || d->Cin % 64 || d->Cout % 64) return 0;
    const int rows = fwd_stat_rows(d, true);
    const size_t fused = (size_t)(rows + sd_bn_finalize_scratch_rows(rows)) * 2 * d->Cout * sizeof(float);
    const size_t split = sd_conv2d_fwd_bf16_workspace_bytes(d) + sd_col_reduce_workspace_bytes((int64_t)d->B * d->Ho * d->Wo, d->Cout);
    return std::max(std::max(fused, split), (size_t)256);
}

int sd_conv2d_fwd_bf16_bn_stats(const void* x, const void* w, void* y, const sd_conv_desc* d, float eps, float momentum, float* running_mean,
                                float* running_var, float* mean, float* invstd, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_fwd_bf16_bn_stats", d)) return e;
    SD_REQUIRE(x && w && y && mean && invstd && workspace, SD_ERR_INVALID, "sd_conv2d_fwd_bf16_bn_stats: null pointer");
    SD_REQUIRE(d->Cin % 64 == 0 && d->Cout % 64 == 0, SD_ERR_INVALID, "sd_conv2d_fwd_bf16_bn_stats: needs Cin %% 64 == 0 and Cout %% 64 == 0 (got %d, %d)",
               d->Cin, d->Cout);
    SD_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y), SD_ERR_ALIGN, "sd_conv2d_fwd_bf16_bn_stats: pointers must be 16-byte aligned");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_fwd_bf16_bn_stats_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_fwd_bf16_bn_stats: workspace too small");
    const int rows = fwd_stat_rows(d, true);
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    if (rows == 0) {        // small batch (split-K): plain conv, then the separate statistics pass over the bf16 output
        const size_t cw = sd_conv2d_fwd_bf16_workspace_bytes(d);
        if (int e = sd_conv2d_fwd_bf16(x, w, y, d, nullptr, nullptr, nullptr, 0, 0, workspace, cw, stream)) return e;
        return sd_bn_train_stats_bf16(y, M, d->Cout, eps, momentum, running_mean, running_var, mean, invstd, (char*)workspace + cw,
                                      workspace_bytes - cw, stream);
    }
    ConvArgs a{};
    a.x = x; a.w = w; a.y = y; a.stat = (float*)workspace;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.mul = d->stride; a.div = 1; a.off = -d->pad; a.rsign = 1;
    a.M = d->B * d->Ho * d->Wo; a.kchunks = d->Cin / 64; a.nk = d->R * d->S * a.kchunks; a.splits = 1;
    if (int e = launch_igemm(a, false, (hipStream_t)stream, true)) return e;
    return sd_bn_finalize_stats((const float*)workspace, rows, M, d->Cout, eps, momentum, running_mean, running_var, mean, invstd,
                                (float*)workspace + (size_t)rows * 2 * d->Cout, stream);
}

// data-gradient with bf16 dy / transposed weights / dx (+ bf16 residual: res_mode 0 none, 1 same size, 2 half-size map added at even pixels)
int sd_conv2d_dgrad_bf16(const void* dy, const void* w_t, void* dx, const sd_conv_desc* d, const void* residual, int res_mode, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_dgrad_bf16", d)) return e;
    SD_REQUIRE(dy && w_t && dx && (res_mode == 0 || residual), SD_ERR_INVALID, "sd_conv2d_dgrad_bf16: null pointer");
    SD_REQUIRE(res_mode >= 0 && res_mode <= 2, SD_ERR_INVALID, "sd_conv2d_dgrad_bf16: res_mode must be 0, 1 or 2");
    SD_REQUIRE(d->Cout % 64 == 0 && d->Cin % 64 == 0, SD_ERR_INVALID, "sd_conv2d_dgrad_bf16: needs Cout %% 64 == 0 and Cin %% 64 == 0");
    SD_REQUIRE(res_mode != 2 || (d->Hi % 2 == 0 && d->Wi % 2 == 0), SD_ERR_INVALID, "sd_conv2d_dgrad_bf16: a half-size residual needs even Hi, Wi");
    SD_REQUIRE(aligned16(dy) && aligned16(w_t) && aligned16(dx) && aligned16(residual), SD_ERR_ALIGN, "sd_conv2d_dgrad_bf16: pointers must be 16-byte aligned");
    ConvArgs a{};
    fill_dgrad(a, d);
    a.kchunks = d->Cout / 64; a.nk = d->R * d->S * a.kchunks;
    a.x = dy; a.w = w_t; a.y = dx; a.res = res_mode ? residual : nullptr; a.res_up2 = res_mode == 2 ? 2 : 0;
    return launch_igemm(a, false, (hipStream_t)stream, true);
}

int sd_conv2d_stem_fwd(const float* x_nchw, const float* w, void* y, const sd_conv_desc* d, const float* scale, const float* shift, int relu,
                       int out_bf16, void* workspace, size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_stem_fwd", d)) return e;
    SD_REQUIRE(x_nchw && w && y, SD_ERR_INVALID, "sd_conv2d_stem_fwd: null pointer");
    SD_REQUIRE(d->Cin == 3 && d->Cout == 64, SD_ERR_INVALID, "sd_conv2d_stem_fwd: the stem is 3 -> 64 channels (network.py:43)");
    hipStream_t st = (hipStream_t)stream;
    if (stem_is_7x7s2(d) && workspace && workspace_bytes >= sd_conv2d_stem_fwd_workspace_bytes(d)) {
        // LDS-patch kernel: weights are re-laid as [k][cout] once per call (37 KB)
        float* wt = (float*)workspace;
        hipLaunchKernelGGL(k_transpose_w<float>, dim3(cdiv(STEM_K, 32), 2, 1), dim3(256), 0, st, w, wt, 64, 1, STEM_K);
        SD_LAUNCH_CHECK();
        StemArgs a{};
        a.x = x_nchw; a.wt = wt; a.y = (float*)y; a.scale = scale; a.shift = shift; a.relu = relu; a.out_bf16 = out_bf16;
        stem_args(a, d);
        const size_t lds = STEM_FWD_LDS_BYTES;
        // one-time, thread-safe (C++11 static initialisation): allow > 64 KB of dynamic LDS for these kernels
        static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        static const hipError_t attr_once_b = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)attr_once; (void)attr_once_b;
        a.w = w;
        if (out_bf16) hipLaunchKernelGGL(k_stem_fwd<true>, dim3(std::min(a.ntiles, 512)), dim3(256), lds, st, a);     // bf16 backbone: bf16 MFMA, bf16 output
        else if (stem_dma_ok(d, x_nchw, y)) hipLaunchKernelGGL(k_stem_fwd_dma<false>, dim3(std::min(a.ntiles, 512)), dim3(256), (size_t)STEM_KPAD * 64 * sizeof(float), st, a);
        else hipLaunchKernelGGL(k_stem_fwd<false>, dim3(std::min(a.ntiles, 512)), dim3(256), lds, st, a);
        SD_LAUNCH_CHECK();
        return 0;
    }
    SD_REQUIRE(!out_bf16, SD_ERR_INVALID, "sd_conv2d_stem_fwd: bf16 output needs the 7x7/2 geometry and a workspace");
    ConvArgs a{};
    a.x = x_nchw; a.w = w; a.y = y; a.scale = scale; a.shift = shift; a.relu = relu;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = 3; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = 64; a.R = d->R; a.S = d->S;
    a.mul = d->stride; a.div = 1; a.off = -d->pad; a.rsign = 1;
    a.M = d->B * d->Ho * d->Wo; a.kchunks = 1; a.nk = cdiv(d->R * d->S * 3, BK);
    return launch_igemm(a, true, st);
}

size_t sd_conv2d_stem_fwd_bn_stats_workspace_bytes(const sd_conv_desc* d) {
    if (!d) return 0;
    StemArgs a{};
    stem_args(a, d);
    const size_t wt = align_up((size_t)STEM_K * 64 * sizeof(float), 256);
    return wt + (size_t)(a.ntiles + sd_bn_finalize_scratch_rows(a.ntiles)) * 2 * 64 * sizeof(float);
}

int sd_conv2d_stem_fwd_bn_stats(const float* x_nchw, const float* w, float* y, const sd_conv_desc* d, float eps, float momentum,
                                float* running_mean, float* running_var, float* mean, float* invstd, void* workspace, size_t workspace_bytes,
                                sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_stem_fwd_bn_stats", d)) return e;
    SD_REQUIRE(x_nchw && w && y && mean && invstd && workspace, SD_ERR_INVALID, "sd_conv2d_stem_fwd_bn_stats: null pointer");
    SD_REQUIRE(stem_is_7x7s2(d), SD_ERR_INVALID, "sd_conv2d_stem_fwd_bn_stats: the stem is a 7x7 / stride 2 / pad 3 conv, 3 -> 64 channels");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_stem_fwd_bn_stats_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_stem_fwd_bn_stats: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* wt = (float*)workspace;
    float* partial = (float*)((char*)workspace + align_up((size_t)STEM_K * 64 * sizeof(float), 256));
    hipLaunchKernelGGL(k_transpose_w<float>, dim3(cdiv(STEM_K, 32), 2, 1), dim3(256), 0, st, w, wt, 64, 1, STEM_K);
    SD_LAUNCH_CHECK();
    StemArgs a{};
    a.x = x_nchw; a.wt = wt; a.y = y; a.stat = partial;
    stem_args(a, d);
    if (!stem_dma_ok(d, x_nchw, y)) {                // odd widths / unaligned views: the register-staged kernel, one partial row per tile
        const size_t lds = STEM_FWD_LDS_BYTES;
        static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)attr_once;
        hipLaunchKernelGGL(k_stem_fwd<false>, dim3(std::min(a.ntiles, 512)), dim3(256), lds, st, a);
        SD_LAUNCH_CHECK();
        return sd_bn_finalize_stats(partial, a.ntiles, (int64_t)d->B * d->Ho * d->Wo, 64, eps, momentum, running_mean, running_var, mean, invstd,
                                    partial + (size_t)a.ntiles * 128, stream);
    }
    // k_stem_fwd_dma: one partial statistics row per BLOCK (the workspace holds one per tile: more than enough)
    const int blocks = std::min(a.ntiles, std::max(1, g_stem_fwd_blocks));
    hipLaunchKernelGGL(k_stem_fwd_dma<true>, dim3(blocks), dim3(256), (size_t)STEM_KPAD * 64 * sizeof(float), st, a);
    SD_LAUNCH_CHECK();
    return sd_bn_finalize_stats(partial, blocks, (int64_t)d->B * d->Ho * d->Wo, 64, eps, momentum, running_mean, running_var, mean, invstd,
                                partial + (size_t)a.ntiles * 128, stream);
}

// the same stem launch with the product on the bf16 MFMA (image patch and weights rounded to bf16 on the way into the operands, fp32
// accumulation, fp32 output + statistics): the mixed-precision training step (under autocast the reference's conv1 runs in bf16 too)
int sd_conv2d_stem_fwd_bn_stats_bf16mm(const float* x_nchw, const float* w, float* y, const sd_conv_desc* d, float eps, float momentum,
                                       float* running_mean, float* running_var, float* mean, float* invstd, void* workspace,
                                       size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_stem_fwd_bn_stats_bf16mm", d)) return e;
    SD_REQUIRE(x_nchw && w && y && mean && invstd && workspace, SD_ERR_INVALID, "sd_conv2d_stem_fwd_bn_stats_bf16mm: null pointer");
    SD_REQUIRE(stem_is_7x7s2(d), SD_ERR_INVALID, "sd_conv2d_stem_fwd_bn_stats_bf16mm: the stem is a 7x7 / stride 2 / pad 3 conv, 3 -> 64 channels");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_stem_fwd_bn_stats_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_stem_fwd_bn_stats_bf16mm: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)((char*)workspace + align_up((size_t)STEM_K * 64 * sizeof(float), 256));
    StemArgs a{};
    a.x = x_nchw; a.w = w; a.y = y; a.stat = partial;
    stem_args(a, d);
    const size_t lds = STEM_FWD_LDS_BYTES;
    static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr_once;
    hipLaunchKernelGGL(k_stem_fwd<true>, dim3(std::min(a.ntiles, 512)), dim3(256), lds, st, a);
    SD_LAUNCH_CHECK();
    return sd_bn_finalize_stats(partial, a.ntiles, (int64_t)d->B * d->Ho * d->Wo, 64, eps, momentum, running_mean, running_var, mean, invstd,
                                partial + (size_t)a.ntiles * 128, stream);
}

// the same with a bf16 NHWC output (mixed-precision training: the conv output is read twice more by the stem tail, 1 GB in fp32 at bs=64)
int sd_conv2d_stem_fwd_bn_stats_bf16(const float* x_nchw, const float* w, void* y_bf16, const sd_conv_desc* d, float eps, float momentum,
                                     float* running_mean, float* running_var, float* mean, float* invstd, void* workspace,
                                     size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_stem_fwd_bn_stats_bf16", d)) return e;
    SD_REQUIRE(x_nchw && w && y_bf16 && mean && invstd && workspace, SD_ERR_INVALID, "sd_conv2d_stem_fwd_bn_stats_bf16: null pointer");
    SD_REQUIRE(stem_is_7x7s2(d), SD_ERR_INVALID, "sd_conv2d_stem_fwd_bn_stats_bf16: the stem is a 7x7 / stride 2 / pad 3 conv, 3 -> 64 channels");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_stem_fwd_bn_stats_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_stem_fwd_bn_stats_bf16: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* partial = (float*)((char*)workspace + align_up((size_t)STEM_K * 64 * sizeof(float), 256));
    StemArgs a{};
    a.x = x_nchw; a.w = w; a.y = (float*)y_bf16; a.stat = partial; a.out_bf16 = 1;
    stem_args(a, d);
    if (g_stem_fwd_ring && d->Wi % 4 == 0 && aligned16(x_nchw) && aligned16(y_bf16)) {        // row-ring kernel: image rows by LDS-DMA in groups of four columns
        a.rg = 64;
        while (a.rg > 8 && d->B * a.tiles_x * cdiv(d->Ho, a.rg) < 512) a.rg >>= 1;             // one block per CU, two units each
        const int units = d->B * a.tiles_x * cdiv(d->Ho, a.rg);
        const int blocks = std::max(1, std::min(std::min(256, a.ntiles), units));
        static const hipError_t attr_ring = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd_bf16_ring), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF_LDS_BYTES);
        (void)attr_ring;
        hipLaunchKernelGGL(k_stem_fwd_bf16_ring, dim3(blocks), dim3(512), SF_LDS_BYTES, st, a);
        SD_LAUNCH_CHECK();
        return sd_bn_finalize_stats(partial, blocks, (int64_t)d->B * d->Ho * d->Wo, 64, eps, momentum, running_mean, running_var, mean, invstd,
                                    partial + (size_t)a.ntiles * 128, stream);
    }
    const size_t lds = STEM_FWD_LDS_BYTES;
    static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_fwd<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr_once;
    hipLaunchKernelGGL(k_stem_fwd<true>, dim3(std::min(a.ntiles, 512)), dim3(256), lds, st, a);
    SD_LAUNCH_CHECK();
    return sd_bn_finalize_stats(partial, a.ntiles, (int64_t)d->B * d->Ho * d->Wo, 64, eps, momentum, running_mean, running_var, mean, invstd,
                                partial + (size_t)a.ntiles * 128, stream);
}

int sd_conv2d_dgrad_half_res(const float* dy, const float* w_t, float* dx, const sd_conv_desc* d, const float* residual_half,
                             sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_dgrad_half_res", d)) return e;
    SD_REQUIRE(dy && w_t && dx && residual_half, SD_ERR_INVALID, "sd_conv2d_dgrad_half_res: null pointer");
    SD_REQUIRE(d->Cout % 32 == 0 && d->Cin % 64 == 0, SD_ERR_INVALID, "sd_conv2d_dgrad_half_res: needs Cout %% 32 == 0 and Cin %% 64 == 0");
    SD_REQUIRE(d->Hi % 2 == 0 && d->Wi % 2 == 0, SD_ERR_INVALID, "sd_conv2d_dgrad_half_res: needs even Hi, Wi");
    SD_REQUIRE(aligned16(dy) && aligned16(w_t) && aligned16(dx) && aligned16(residual_half), SD_ERR_ALIGN,
               "sd_conv2d_dgrad_half_res: pointers must be 16-byte aligned");
    ConvArgs a{};
    fill_dgrad(a, d);
    a.x = dy; a.w = w_t; a.y = dx; a.res = residual_half; a.res_up2 = 2;
    return launch_igemm(a, false, (hipStream_t)stream);
}

int sd_conv2d_dgrad(const float* dy, const float* w_t, float* dx, const sd_conv_desc* d, const float* residual, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_dgrad", d)) return e;
    SD_REQUIRE(dy && w_t && dx, SD_ERR_INVALID, "sd_conv2d_dgrad: null pointer");
    SD_REQUIRE(d->Cout % 32 == 0 && d->Cin % 64 == 0, SD_ERR_INVALID, "sd_conv2d_dgrad: needs Cout %% 32 == 0 and Cin %% 64 == 0");
    SD_REQUIRE(aligned16(dy) && aligned16(w_t) && aligned16(dx) && aligned16(residual), SD_ERR_ALIGN, "sd_conv2d_dgrad: pointers must be 16-byte aligned");
    ConvArgs a{};
    fill_dgrad(a, d);
    a.x = dy; a.w = w_t; a.y = dx; a.res = residual;
    return launch_igemm(a, false, (hipStream_t)stream);
}

int sd_conv2d_transpose_weights(const float* w, float* w_t, int Cout, int taps, int Cin, sd_stream_t stream) {
    SD_REQUIRE(w && w_t && Cout > 0 && taps > 0 && Cin > 0, SD_ERR_INVALID, "sd_conv2d_transpose_weights: bad arguments");
    hipLaunchKernelGGL(k_transpose_w<float>, dim3(cdiv(Cin, 32), cdiv(Cout, 32), taps), dim3(256), 0, (hipStream_t)stream, w, w_t, Cout, taps, Cin);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_conv2d_transpose_weights_bf16(const float* w, void* w_t_bf16, int Cout, int taps, int Cin, sd_stream_t stream) {
    SD_REQUIRE(w && w_t_bf16 && Cout > 0 && taps > 0 && Cin > 0, SD_ERR_INVALID, "sd_conv2d_transpose_weights_bf16: bad arguments");
    hipLaunchKernelGGL(k_transpose_w<uint16_t>, dim3(cdiv(Cin, 32), cdiv(Cout, 32), taps), dim3(256), 0, (hipStream_t)stream, w, (uint16_t*)w_t_bf16,
                       Cout, taps, Cin);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_conv2d_transpose_weights_batched(const float* w_base, void* w_t_base, const int* table, int nconv, int total_blocks, int out_bf16,
                                        sd_stream_t stream) {
    SD_REQUIRE(w_base && w_t_base && table && nconv > 0 && total_blocks > 0, SD_ERR_INVALID, "sd_conv2d_transpose_weights_batched: bad arguments");
    if (out_bf16) hipLaunchKernelGGL(k_transpose_w_batched<uint16_t>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, w_base, (uint16_t*)w_t_base, table, nconv);
    else hipLaunchKernelGGL(k_transpose_w_batched<float>, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, w_base, (float*)w_t_base, table, nconv);
    SD_LAUNCH_CHECK();
    return 0;
}

// layer1-type weight gradients (3x3 / 1 / 1, 64 -> 64 channels, rows that are whole 32-pixel chunks): all taps in one block
static bool wgrad_all_taps(const sd_conv_desc* d) {
    return d->Cin % 64 == 0 && d->Cout % 64 == 0 && d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 &&
           (d->Wo % 32 == 0 || (d->Wo == 16 && d->Ho % 2 == 0));
}

static thread_local int g_wgrad_f32_ring = 2;        // sd_set_option("wgrad_f32_ring", n): 2 = k_wgrad3x3_ring2 (two groups per 512-thread block), 1 = k_wgrad3x3_ring, 0 = k_wgrad3x3<32> (A/B, tests)
static thread_local int g_wgrad_bf16_ring = 5;       // sd_set_option("wgrad_bf16_ring", n): 5 (default) = k_wgrad3x3_bf16_ring2 (two groups per 512-thread block); 2 .. 4 = k_wgrad3x3_bf16_ring with that prefetch distance; 0 = k_wgrad3x3_bf16<32> (A/B, tests)
static int wgrad_splits(const sd_conv_desc* d, int tiles) {
    if (wgrad_all_taps(d)) {
        const int chunks = d->B * d->Ho * d->Wo / 32;
        return std::max(1, std::min(cdiv(512, tiles), chunks / 8));      // two resident blocks per CU, >= 8 chunks each
    }
    // Pick the split count so that tiles*splits fills whole "rounds" of the chip (256 CUs x 2 resident 128x128
    // blocks, x4 for the 64x64 tile): a last round that is mostly empty costs as much as a full one.
    const int M = d->B * d->Ho * d->Wo;
    const bool small = (d->Cout % 128 != 0) && (d->Cin % 128 != 0);
    const int slots = 256 * (small ? 4 : 2);
    const int max_s = std::max(1, cdiv(M, 512));           // at least 512 pixels (16 chunks) per split
    int best_s = 1;
    double best_fill = 0.0;
    for (int k = 1; k <= 4; ++k) {
        int s = std::min(std::max(1, k * slots / tiles), max_s);
        const int rounds = cdiv(tiles * s, slots);
        const double fill = (double)tiles * s / ((double)rounds * slots);
        if (fill > best_fill + 0.03) { best_fill = fill; best_s = s; }
    }
    return best_s;
}

size_t sd_conv2d_wgrad_workspace_bytes(const sd_conv_desc* d) {
    if (!d || d->Cin % 64 || d->Cout % 64) return 0;
    const int TN = d->Cout % 128 == 0 ? 128 : 64, TC = d->Cin % 128 == 0 ? 128 : 64;
    const int tiles = wgrad_all_taps(d) ? (d->Cout / 64) * (d->Cin / 64) : d->R * d->S * (d->Cout / TN) * (d->Cin / TC);
    return (size_t)wgrad_splits(d, tiles) * d->Cout * d->R * d->S * d->Cin * sizeof(float);
}

int sd_conv2d_wgrad(const float* dy, const float* x, float* dw, const sd_conv_desc* d, int accumulate, void* workspace,
                    size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_wgrad", d)) return e;
    SD_REQUIRE(dy && x && dw && workspace, SD_ERR_INVALID, "sd_conv2d_wgrad: null pointer");
    SD_REQUIRE(d->Cin % 64 == 0 && d->Cout % 64 == 0, SD_ERR_INVALID, "sd_conv2d_wgrad: needs Cin, Cout %% 64 == 0");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_wgrad_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_wgrad: workspace too small");
    const int TN = d->Cout % 128 == 0 ? 128 : 64, TC = d->Cin % 128 == 0 ? 128 : 64;
    const int tiles = wgrad_all_taps(d) ? (d->Cout / 64) * (d->Cin / 64) : d->R * d->S * (d->Cout / TN) * (d->Cin / TC);
    WgradArgs a{};
    a.dy = dy; a.x = x; a.part = (float*)workspace;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.splits = wgrad_splits(d, tiles);
    a.m_per_split = cdiv(cdiv(a.M, a.splits), 32) * 32;
    hipStream_t st = (hipStream_t)stream;
    const int64_t n4 = (int64_t)d->Cout * d->R * d->S * d->Cin / 4;
    if (wgrad_all_taps(d)) {
        if (d->Wo % 32 == 0 && g_wgrad_f32_ring >= 2) {
            a.strips = d->Wo / 32; a.chunks_total = d->B * a.strips * d->Ho;
            a.splits = std::max(1, std::min(cdiv(256, tiles), a.chunks_total / 16));
            a.chunks_per_split = cdiv(a.chunks_total, a.splits);
            a.splits = cdiv(a.chunks_total, a.chunks_per_split);
            hipLaunchKernelGGL(k_wgrad3x3_ring2, dim3(a.splits, tiles), dim3(512), 0, st, a);
        } else if (d->Wo % 32 == 0 && g_wgrad_f32_ring) {
            a.strips = d->Wo / 32; a.chunks_total = d->B * a.strips * d->Ho; a.chunks_per_split = cdiv(a.chunks_total, a.splits);
            hipLaunchKernelGGL(k_wgrad3x3_ring, dim3(a.splits, tiles), dim3(256), 0, st, a);
        } else if (d->Wo % 32 == 0) hipLaunchKernelGGL(k_wgrad3x3<32>, dim3(a.splits, tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_wgrad3x3<16>, dim3(a.splits, tiles), dim3(256), 0, st, a);
        SD_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_wgrad_reduce_par, dim3(cdiv(n4, 32)), dim3(256), 0, st, (const float*)workspace, dw, n4, a.splits, accumulate);
        SD_LAUNCH_CHECK();
        return 0;
    }
    const size_t lds = (size_t)2 * 32 * (TN + 4 + TC + 4) * sizeof(float);
    dim3 grid(tiles, a.splits);
    static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_wgrad<128, 128>),
                                                            hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 32 * (128 + 4 + 128 + 4) * 4);
    (void)attr_once;
    if (TN == 128 && TC == 128) hipLaunchKernelGGL((k_conv_wgrad<128, 128>), grid, dim3(256), lds, st, a);
    else if (TN == 128) hipLaunchKernelGGL((k_conv_wgrad<128, 64>), grid, dim3(256), lds, st, a);
    else if (TC == 128) hipLaunchKernelGGL((k_conv_wgrad<64, 128>), grid, dim3(256), lds, st, a);
    else hipLaunchKernelGGL((k_conv_wgrad<64, 64>), grid, dim3(256), lds, st, a);
    SD_LAUNCH_CHECK();
    if (a.splits >= 16) hipLaunchKernelGGL(k_wgrad_reduce_par, dim3(cdiv(n4, 32)), dim3(256), 0, st, (const float*)workspace, dw, n4, a.splits, accumulate);
    else hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(n4, 256)), dim3(256), 0, st, (const float*)workspace, dw, n4, a.splits, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

// ---- mixed-precision weight gradient: dY and X bf16, dW fp32.  3x3 / stride 1 layers (32 of the 42 weight-gradient launches of a
// step, 97 % of the flops) run k_wgrad3x3_bf16; the strided and 1x1 convs widen their operands to fp32 in the workspace and take
// the fp32 kernels (exact: a bf16 value is an fp32 value).
static bool wgrad_tap_bf16(const sd_conv_desc* d) { return !wgrad_all_taps(d) && d->Cout % 128 == 0 && d->Cin % 64 == 0; }
static int wgrad_tap_bf16_tiles(const sd_conv_desc* d) { return d->R * d->S * (d->Cout / 128) * (d->Cin / (d->Cin % 128 == 0 ? 128 : 64)); }
static int wgrad_tap_bf16_splits(const sd_conv_desc* d) {
    // fill two blocks per CU (512 slots) with whole rounds, at least 8 chunks (256 pixels) per split
    const int tiles = wgrad_tap_bf16_tiles(d), M = d->B * d->Ho * d->Wo;
    return std::max(1, std::min(cdiv(1024, tiles), M / 256));
}

size_t sd_conv2d_wgrad_bf16_workspace_bytes(const sd_conv_desc* d) {
    if (!d || d->Cin % 64 || d->Cout % 64) return 0;
    const size_t base = align_up(sd_conv2d_wgrad_workspace_bytes(d), 256);
    if (wgrad_all_taps(d)) return base;
    if (wgrad_tap_bf16(d)) return align_up((size_t)wgrad_tap_bf16_splits(d) * d->Cout * d->R * d->S * d->Cin * sizeof(float), 256);
    return base + align_up((size_t)d->B * d->Ho * d->Wo * d->Cout * 4, 256) + align_up((size_t)d->B * d->Hi * d->Wi * d->Cin * 4, 256);
}

int sd_conv2d_wgrad_bf16(const void* dy, const void* x, float* dw, const sd_conv_desc* d, int accumulate, void* workspace,
                         size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_wgrad_bf16", d)) return e;
    SD_REQUIRE(dy && x && dw && workspace, SD_ERR_INVALID, "sd_conv2d_wgrad_bf16: null pointer");
    SD_REQUIRE(d->Cin % 64 == 0 && d->Cout % 64 == 0, SD_ERR_INVALID, "sd_conv2d_wgrad_bf16: needs Cin, Cout %% 64 == 0");
    SD_REQUIRE(aligned16(dy) && aligned16(x), SD_ERR_ALIGN, "sd_conv2d_wgrad_bf16: pointers must be 16-byte aligned");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_wgrad_bf16_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_wgrad_bf16: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    if (wgrad_tap_bf16(d)) {
        WgradArgs16t a{};
        a.dy = (const uint16_t*)dy; a.x = (const uint16_t*)x; a.part = (float*)workspace;
        a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
        a.stride = d->stride; a.pad = d->pad;
        a.M = d->B * d->Ho * d->Wo;
        a.splits = wgrad_tap_bf16_splits(d);
        a.m_per_split = cdiv(cdiv(a.M, a.splits), 32) * 32;
        const dim3 grid(a.splits, wgrad_tap_bf16_tiles(d));
        if (d->Cin % 128 == 0) hipLaunchKernelGGL(k_wgrad_tap_bf16<128>, grid, dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_wgrad_tap_bf16<64>, grid, dim3(256), 0, st, a);
        SD_LAUNCH_CHECK();
        const int64_t n4 = (int64_t)d->Cout * d->R * d->S * d->Cin / 4;
        if (a.splits >= 16) hipLaunchKernelGGL(k_wgrad_reduce_par, dim3(cdiv(n4, 32)), dim3(256), 0, st, (const float*)workspace, dw, n4, a.splits, accumulate);
        else hipLaunchKernelGGL(k_wgrad_reduce, dim3(cdiv(n4, 256)), dim3(256), 0, st, (const float*)workspace, dw, n4, a.splits, accumulate);
        SD_LAUNCH_CHECK();
        return 0;
    }
    if (!wgrad_all_taps(d)) {       // (Cout not a multiple of 128: no such layer in SDNet) widen to fp32 in the workspace, fp32 kernels
        const size_t base = align_up(sd_conv2d_wgrad_workspace_bytes(d), 256);
        const int64_t ndy = (int64_t)d->B * d->Ho * d->Wo * d->Cout, nx = (int64_t)d->B * d->Hi * d->Wi * d->Cin;
        float* dy32 = reinterpret_cast<float*>((char*)workspace + base);
        float* x32 = reinterpret_cast<float*>((char*)workspace + base + align_up((size_t)ndy * 4, 256));
        if (int e = sd_cast_bf16_to_f32(dy, dy32, ndy, stream)) return e;
        if (int e = sd_cast_bf16_to_f32(x, x32, nx, stream)) return e;
        return sd_conv2d_wgrad(dy32, x32, dw, d, accumulate, workspace, base, stream);
    }
    const int tiles = (d->Cout / 64) * (d->Cin / 64);
    WgradArgs16 a{};
    a.dy = (const uint16_t*)dy; a.x = (const uint16_t*)x; a.part = (float*)workspace;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout;
    a.M = d->B * d->Ho * d->Wo;
    a.splits = wgrad_splits(d, tiles);
    a.m_per_split = cdiv(cdiv(a.M, a.splits), 32) * 32;
    const int64_t n4 = (int64_t)d->Cout * 9 * d->Cin / 4;
    if (d->Wo % 32 == 0 && g_wgrad_bf16_ring >= 5) {
        // one 512-thread block per CU, two groups half a chunk apart: half as many (twice as long) splits
        a.strips = d->Wo / 32; a.chunks_total = d->B * a.strips * d->Ho;
        a.splits = std::max(1, std::min(cdiv(256, tiles), a.chunks_total / 16));
        a.chunks_per_split = cdiv(a.chunks_total, a.splits);
        a.splits = cdiv(a.chunks_total, a.chunks_per_split);
        hipLaunchKernelGGL(k_wgrad3x3_bf16_ring2, dim3(a.splits, tiles), dim3(512), 0, st, a);
    } else if (d->Wo % 32 == 0 && g_wgrad_bf16_ring) {
        a.strips = d->Wo / 32; a.chunks_total = d->B * a.strips * d->Ho; a.chunks_per_split = cdiv(a.chunks_total, a.splits);
        if (g_wgrad_bf16_ring == 2) hipLaunchKernelGGL(k_wgrad3x3_bf16_ring<2>, dim3(a.splits, tiles), dim3(256), 0, st, a);
        else if (g_wgrad_bf16_ring == 3) hipLaunchKernelGGL(k_wgrad3x3_bf16_ring<3>, dim3(a.splits, tiles), dim3(256), 0, st, a);
        else if (g_wgrad_bf16_ring >= 4) hipLaunchKernelGGL(k_wgrad3x3_bf16_ring<4>, dim3(a.splits, tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL(k_wgrad3x3_bf16_ring<3>, dim3(a.splits, tiles), dim3(256), 0, st, a);
    } else if (d->Wo % 32 == 0) hipLaunchKernelGGL(k_wgrad3x3_bf16<32>, dim3(a.splits, tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_wgrad3x3_bf16<16>, dim3(a.splits, tiles), dim3(256), 0, st, a);
    SD_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wgrad_reduce_par, dim3(cdiv(n4, 32)), dim3(256), 0, st, (const float*)workspace, dw, n4, a.splits, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

int sd_stem_bn_relu_maxpool_fwd_bf16(const float* x_nchw, const float* w, const float* scale, const float* shift, void* y, const sd_conv_desc* d,
                                     sd_stream_t stream) {
    if (int e = check_conv("sd_stem_bn_relu_maxpool_fwd_bf16", d)) return e;
    SD_REQUIRE(x_nchw && w && scale && shift && y, SD_ERR_INVALID, "sd_stem_bn_relu_maxpool_fwd_bf16: null pointer");
    SD_REQUIRE(stem_is_7x7s2(d), SD_ERR_INVALID, "sd_stem_bn_relu_maxpool_fwd_bf16: the stem is 7x7 / stride 2 / pad 3, 3 -> 64 (network.py:43)");
    SD_REQUIRE(d->Ho % 2 == 0 && d->Wo % 2 == 0 && d->Wo <= 4096, SD_ERR_INVALID, "sd_stem_bn_relu_maxpool_fwd_bf16: needs even Ho, Wo and Wo <= 4096");
    SD_REQUIRE(d->Wi % 4 == 0, SD_ERR_INVALID, "sd_stem_bn_relu_maxpool_fwd_bf16: the image width must be a multiple of 4 (16-byte row groups)");
    SD_REQUIRE(aligned16(y) && aligned16(x_nchw), SD_ERR_ALIGN, "sd_stem_bn_relu_maxpool_fwd_bf16: x and y must be 16-byte aligned");
    StemPoolArgs a{};
    a.x = x_nchw; a.w = w; a.scale = scale; a.shift = shift; a.y = (uint16_t*)y;
    a.B = d->B; a.H = d->Hi; a.W = d->Wi; a.Ho = d->Ho; a.Wo = d->Wo; a.Hp = d->Ho / 2; a.Wp = d->Wo / 2;
    a.tiles_x = cdiv(d->Wo, 128);
    a.alias_rowbuf = stem_pool_lds_bytes(a.Wp) > 80 * 1024 && stem_pool_lds_bytes(a.Wp, 1) <= 80 * 1024;
    const size_t lds = stem_pool_lds_bytes(a.Wp, a.alias_rowbuf);
    const int per_cu = lds <= 80 * 1024 ? 2 : 1, grid_max = 256 * per_cu;
    // pooled rows per work unit: enough units to fill the persistent grid, at most 16 rows (a unit recomputes one conv row)
    a.rows = std::max(1, std::min(16, (int)((int64_t)a.B * a.Hp / grid_max)));
    a.units_per_img = cdiv(a.Hp, a.rows);
    a.nunits = a.B * a.units_per_img;
    static thread_local size_t raised = 0;
    if (lds > raised) {
        SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_stem_pool_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        raised = lds;
    }
    hipLaunchKernelGGL(k_stem_pool_bf16, dim3(std::min(a.nunits, grid_max)), dim3(256), lds, (hipStream_t)stream, a);
    SD_LAUNCH_CHECK();
    return 0;
}

static int stem_blocks(const sd_conv_desc* d) {
    const int ntiles = d->B * d->Ho * cdiv(d->Wo, 128);
    return std::max(1, std::min(512, ntiles));          // two persistent blocks per CU (252 VGPRs, 55 KB LDS each)
}

size_t sd_conv2d_stem_wgrad_workspace_bytes(const sd_conv_desc* d) {
    if (!d) return 0;
    return (size_t)stem_blocks(d) * 64 * STEM_K * sizeof(float);
}

int sd_conv2d_stem_wgrad(const float* dy, const float* x_nchw, float* dw, const sd_conv_desc* d, int accumulate, void* workspace,
                         size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_stem_wgrad", d)) return e;
    SD_REQUIRE(dy && x_nchw && dw && workspace, SD_ERR_INVALID, "sd_conv2d_stem_wgrad: null pointer");
    SD_REQUIRE(stem_is_7x7s2(d), SD_ERR_INVALID, "sd_conv2d_stem_wgrad: the stem is 7x7 / stride 2 / pad 3, 3 -> 64");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_stem_wgrad_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_stem_wgrad: workspace too small");
    StemArgs a{};
    a.x = x_nchw; a.dy = dy; a.y = (float*)workspace;
    stem_args(a, d);
    const int blocks = stem_blocks(d);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)(SP_ROWS * SP_PITCH + 128 * 64) * sizeof(float);
    static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)attr_once;
    hipLaunchKernelGGL(k_stem_wgrad2, dim3(blocks), dim3(256), lds, st, a);
    SD_LAUNCH_CHECK();
    const int64_t n4 = 64 * STEM_K / 4;
    hipLaunchKernelGGL(k_wgrad_reduce_par, dim3(cdiv(n4, 32)), dim3(256), 0, st, (const float*)workspace, dw, n4, blocks, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

// the same gradient with the product on the bf16 MFMA (dy and the image are rounded to bf16 on their way into the operands; fp32 sums)
int sd_conv2d_stem_wgrad_bf16mm(const float* dy, const float* x_nchw, float* dw, const sd_conv_desc* d, int accumulate, void* workspace,
                                size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_stem_wgrad_bf16mm", d)) return e;
    SD_REQUIRE(dy && x_nchw && dw && workspace, SD_ERR_INVALID, "sd_conv2d_stem_wgrad_bf16mm: null pointer");
    SD_REQUIRE(stem_is_7x7s2(d), SD_ERR_INVALID, "sd_conv2d_stem_wgrad_bf16mm: the stem is 7x7 / stride 2 / pad 3, 3 -> 64");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_stem_wgrad_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_stem_wgrad_bf16mm: workspace too small");
    StemArgs a{};
    a.x = x_nchw; a.dy = dy; a.y = (float*)workspace;
    stem_args(a, d);
    const int blocks = stem_blocks(d);
    hipStream_t st = (hipStream_t)stream;
    static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SW_LDS_BYTES);
    (void)attr_once;
    hipLaunchKernelGGL(k_stem_wgrad_bf16, dim3(blocks), dim3(256), SW_LDS_BYTES, st, a);
    SD_LAUNCH_CHECK();
    const int64_t n4 = 64 * STEM_K / 4;
    hipLaunchKernelGGL(k_wgrad_reduce_par, dim3(cdiv(n4, 32)), dim3(256), 0, st, (const float*)workspace, dw, n4, blocks, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

// the same gradient from a bf16 dy (the mixed-precision step's stem: sd_maxpool_bn_relu_bwd_bf16_dx16 stores it): row-ring kernel
int sd_conv2d_stem_wgrad_bf16(const void* dy_bf16, const float* x_nchw, float* dw, const sd_conv_desc* d, int accumulate, void* workspace,
                              size_t workspace_bytes, sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_stem_wgrad_bf16", d)) return e;
    SD_REQUIRE(dy_bf16 && x_nchw && dw && workspace, SD_ERR_INVALID, "sd_conv2d_stem_wgrad_bf16: null pointer");
    SD_REQUIRE(stem_is_7x7s2(d), SD_ERR_INVALID, "sd_conv2d_stem_wgrad_bf16: the stem is 7x7 / stride 2 / pad 3, 3 -> 64");
    SD_REQUIRE(aligned16(dy_bf16) && aligned16(x_nchw) && d->Wi % 4 == 0, SD_ERR_INVALID,
               "sd_conv2d_stem_wgrad_bf16: dy and the image must be 16-byte aligned and the image width a multiple of 4 (LDS-DMA in 16-byte groups)");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_stem_wgrad_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_stem_wgrad_bf16: workspace too small");
    StemArgs a{};
    a.x = x_nchw; a.dy16 = (const uint16_t*)dy_bf16; a.y = (float*)workspace;
    stem_args(a, d);
    // rows per unit: the longest strips (3 extra row pairs per unit) that still give every CU two units
    a.rg = 64;
    while (a.rg > 8 && d->B * a.tiles_x * cdiv(d->Ho, a.rg) < 512) a.rg >>= 1;
    const int units = d->B * a.tiles_x * cdiv(d->Ho, a.rg);
    const int blocks = std::max(1, std::min(std::min(256, stem_blocks(d)), units));      // one persistent block per CU
    hipStream_t st = (hipStream_t)stream;
    static const hipError_t attr_once = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stem_wgrad_bf16_ring), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SR_LDS_BYTES);
    (void)attr_once;
    hipLaunchKernelGGL(k_stem_wgrad_bf16_ring, dim3(blocks), dim3(256), SR_LDS_BYTES, st, a);
    SD_LAUNCH_CHECK();
    const int64_t n4 = 64 * STEM_K / 4;
    hipLaunchKernelGGL(k_wgrad_reduce_par, dim3(cdiv(n4, 32)), dim3(256), 0, st, (const float*)workspace, dw, n4, blocks, accumulate);
    SD_LAUNCH_CHECK();
    return 0;
}

// partial rows the data-gradient kernel of this geometry writes for the fused BatchNorm-backward reduction
static int dgrad_stat_rows(const sd_conv_desc* d) {
    ConvArgs a{};
    fill_dgrad(a, d);
    const int BN = (a.Nn % 128 == 0) ? 128 : 64;
    const int mode = a.par ? 2 : (a.div > 1 ? 3 : 0);
    ConvArgs t = a;
    return (patch_tile_bn(t, BN, mode, false) || igemm_big_tiles(a, BN, mode)) ? cdiv(a.M, BMB) : cdiv(a.M, BM);
}

size_t sd_conv2d_dgrad_bn_reduce_workspace_bytes(const sd_conv_desc* d) {
    if (!d || d->Cin % 64 || d->Cout % 32) return 0;
    const int rows = dgrad_stat_rows(d);
    return (size_t)(rows + sd_bn_finalize_scratch_rows(rows)) * 2 * d->Cin * sizeof(float);
}

int sd_conv2d_dgrad_bn_reduce(const float* dy, const float* w_t, float* dx, const sd_conv_desc* d, const float* residual, const float* bn_x,
                              const float* bn_y, int relu, const float* mean, const float* invstd, const float* gamma, const float* beta,
                              float* dgamma, float* dbeta, int accumulate, float* means_out, void* workspace, size_t workspace_bytes,
                              sd_stream_t stream) {
    if (int e = check_conv("sd_conv2d_dgrad_bn_reduce", d)) return e;
    SD_REQUIRE(dy && w_t && dx && bn_x && mean && invstd && gamma && dgamma && dbeta && means_out && workspace, SD_ERR_INVALID,
               "sd_conv2d_dgrad_bn_reduce: null pointer");
    SD_REQUIRE(relu >= 0 && relu <= 3 && ((relu != 1 && relu != 3) || bn_y) && (relu != 2 || beta), SD_ERR_INVALID,
               "sd_conv2d_dgrad_bn_reduce: relu must be 0, 1 / 3 (need bn_y) or 2 (needs beta)");
    SD_REQUIRE(d->Cout % 32 == 0 && d->Cin % 64 == 0, SD_ERR_INVALID, "sd_conv2d_dgrad_bn_reduce: needs Cout %% 32 == 0 and Cin %% 64 == 0");
    SD_REQUIRE(aligned16(dy) && aligned16(w_t) && aligned16(dx) && aligned16(residual) && aligned16(bn_x) && aligned16(bn_y) && aligned16(mean) &&
               aligned16(invstd) && aligned16(gamma) && aligned16(beta), SD_ERR_ALIGN, "sd_conv2d_dgrad_bn_reduce: pointers must be 16-byte aligned");
    SD_REQUIRE(workspace_bytes >= sd_conv2d_dgrad_bn_reduce_workspace_bytes(d), SD_ERR_WORKSPACE, "sd_conv2d_dgrad_bn_reduce: workspace too small");
    ConvArgs a{};
    fill_dgrad(a, d);
    a.x = dy; a.w = w_t; a.y = dx; a.res = residual;
    a.bn_x = bn_x; a.bn_y = bn_y; a.bn_relu = relu; a.bn_mean = mean; a.bn_invstd = invstd; a.bn_gamma = gamma; a.bn_beta = beta;
    a.stat = (float*)workspace;
    if (int e = launch_igemm(a, false, (hipStream_t)stream)) return e;
    const int rows = dgrad_stat_rows(d);
    return sd_bn_bwd_finalize((const float*)workspace, rows, (int64_t)a.M, d->Cin, dgamma, dbeta, accumulate, means_out,
                              (float*)workspace + (size_t)rows * 2 * d->Cin, stream);
}

#ifdef SD_PP_TRACE
int sd_debug_pp_trace(unsigned long long* out32) { return (int)hipMemcpyFromSymbol(out32, HIP_SYMBOL(sd::g_pp_trace), sizeof(unsigned long long) * 64); }
int sd_debug_stem_trace(unsigned long long* out64) { return (int)hipMemcpyFromSymbol(out64, HIP_SYMBOL(sd::g_stem_trace), sizeof(unsigned long long) * 64); }
int sd_debug_pp_timeline(unsigned long long* out, int blocks) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sd::g_pp_tl), sizeof(unsigned long long) * 4 * (blocks < 4096 ? blocks : 4096)); }
#endif

// Dispatch thresholds are THREAD-LOCAL (default-initialised in every host thread): two engines driven from two threads of one process
// cannot change each other's kernel choice.
int sd_set_option(const char* name, int value) {
    if (name && !strcmp(name, "conv_patch_min_tiles")) { g_patch_min_tiles = value; return 0; }
    if (name && !strcmp(name, "conv_patch_bn64")) { g_patch_bn64 = value; return 0; }
    if (name && !strcmp(name, "conv_pp_min_tiles")) { g_pp_min_tiles = value; return 0; }
    if (name && !strcmp(name, "conv1x1_stream_min_pixels")) { g_conv1x1_stream_min_px = value; return 0; }
    if (name && !strcmp(name, "igemm_big_bf16")) { g_igemm_big_bf16 = value; return 0; }
    if (name && !strcmp(name, "stem_fwd_ring")) { g_stem_fwd_ring = value; return 0; }
    if (name && !strcmp(name, "pool_fwd_pair")) { sd_nn_set_pool_pair(value); return 0; }
    if (name && !strcmp(name, "wgrad_bf16_ring")) { g_wgrad_bf16_ring = value; return 0; }
    if (name && !strcmp(name, "wgrad_f32_ring")) { g_wgrad_f32_ring = value; return 0; }
    if (name && !strcmp(name, "conv_pp_strips")) { g_pp_strips = value; return 0; }
    if (name && !strcmp(name, "conv_patch_narrow")) { g_patch_narrow = value; return 0; }
    if (name && !strcmp(name, "conv_fwd_split_k")) { g_fwd_split_k = value; return 0; }
    if (name && !strcmp(name, "conv_rows64_min_units")) { g_rows64_min_units = value; return 0; }
    if (name && !strcmp(name, "conv_rows16")) { g_rows16 = value; return 0; }
    if (name && !strcmp(name, "conv_rows_f32_min_units")) { g_rowsf32_min_units = value; return 0; }
    if (name && !strcmp(name, "stem_fwd_blocks")) { g_stem_fwd_blocks = value; return 0; }
    sd::set_error("sd_set_option: unknown option '%s'", name ? name : "(null)");
    return SD_ERR_INVALID;
}

const char* sd_conv2d_kernel_name(const sd_conv_desc* d, int pass) {
    static thread_local char name[64];
    if (!d || d->Cin <= 0 || d->Cout <= 0) return "";
    if (pass == 2) {
        if (wgrad_all_taps(d)) return d->Wo % 32 == 0 ? (g_wgrad_f32_ring >= 2 ? "k_wgrad3x3_ring2" : g_wgrad_f32_ring ? "k_wgrad3x3_ring" : "k_wgrad3x3<32>") : "k_wgrad3x3<16>";
        snprintf(name, sizeof(name), "k_conv_wgrad<%d, %d>", d->Cout % 128 == 0 ? 128 : 64, d->Cin % 128 == 0 ? 128 : 64);
        return name;
    }
    ConvArgs a{};
    const bool bf16 = (pass & 16) != 0;
    pass &= 15;
    if (pass == 0) fill_fwd(a, d); else fill_dgrad(a, d);
    const int BN = (a.Nn % 128 == 0) ? 128 : 64;
    const int mode = a.par ? 2 : (a.div > 1 ? 3 : 0);
    ConvArgs t = a;
    if (bf16) {          // the bf16 dispatch of launch_igemm (chunks of 64 channels; forward: split-K for small grids)
        a.kchunks = a.Ck / 64; a.nk = a.R * a.S * a.kchunks;
        a.splits = pass == 0 ? fwd_splits(d, 64) : 1;
        t = a;
        if (conv1x1_stream_geometry(a, mode)) return a.Ck == 64 ? "k_conv1x1_stream_bf16<64, 2>" : "k_conv1x1_stream_bf16<128, 1>";   // (launches with a scale or statistics: k_conv_igemm)
        RowsArgs ra;
        if (conv_rows64_geometry(a, mode, ra)) return conv_rows16_args(ra) ? "k_conv3x3_c64_rows16_bf16" : "k_conv3x3_c64_rows_bf16";
        if (conv_pp_geometry(t, mode)) return "k_conv3x3_bf16_pp";
        t = a;
        if (const int PBN = patch_tile_bn(t, BN, mode, true)) snprintf(name, sizeof(name), "k_conv3x3_patch%s<%d, true>", t.pt_rolling ? "_roll" : "", PBN);
        else if (g_igemm_big_bf16 && BN == 128 && igemm_big_tiles(a, BN, mode)) snprintf(name, sizeof(name), "k_conv_igemm_big<%d, %d, true>", BN, mode);
        else snprintf(name, sizeof(name), "k_conv_igemm<%d, %d, true>", BN, mode);
        return name;
    }
    RowsArgsF rf;
    if (conv_rowsf32_geometry(a, mode, rf)) return "k_conv3x3_c64_rows_f32";
    if (const int PBN = patch_tile_bn(t, BN, mode, false)) snprintf(name, sizeof(name), "k_conv3x3_patch%s<%d, false>", t.pt_rolling ? "_roll" : "", PBN);
    else if (igemm_big_tiles(a, BN, mode)) snprintf(name, sizeof(name), "k_conv_igemm_big<%d, %d, false>", BN, mode);
    else snprintf(name, sizeof(name), "k_conv_igemm<%d, %d, false>", BN, mode);
    return name;
}

}  // extern "C"
