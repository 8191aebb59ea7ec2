// Small-batch inference forward convolution (BASELINE configs[1]: bs = 1, 512x512) for gfx950.
// Replaces the ATen / cuDNN convolutions behind `net(batch["image"])` in the reference's batch-1 evaluate loop
// (src/sdnet/cli/evaluate.py:34-45 -> src/sdnet/model/network.py:59-84) for the layers whose 128-row tile grid cannot fill
// 256 CUs.  fp32 (v_mfma_f32_32x32x2_f32) and bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogue).
//
// Why a kernel of its own: at one image the trunk's GEMMs are 16384 x 64 x 576 ... 256 x 512 x 4608 (pixels x Cout x K): the
// 128 x 128 tiles of k_conv_igemm give 8 ... 128 blocks, so K was split 4 ... 36 ways into 288 unbalanced blocks whose 17-19 MB of
// partial tiles a second launch (k_splitk_reduce) summed: 26 + 6.7 us per conv for 7.7 us of MFMA work.  Here
//   * tile 64 (pixels) x 64 (channels), four waves, one 32 x 32 accumulator each: 256 / 128 / 64 / 32 tiles for layer1 .. layer4,
//     so K splits 1 / 2 / 4 / 8 ways give 256 equal blocks (one per CU) of 18 chunks each;
//   * four LDS stages of 16 KB (64 rows x 128 B per operand), filled by LDS-DMA three chunks ahead, counted vmcnt waits and a bare
//     s_barrier per chunk (a __syncthreads() would drain the DMAs in flight); fragment reads are inline-asm ds_read_b128;
//   * the split-K combine happens INSIDE the launch: every slice block stores its 16 KB fp32 slab write-through (sc1), drains,
//     and draws an arrival ticket (agent-scope atomic add); the block that draws the last ticket sums the slabs IN SLICE ORDER
//     (deterministic: the order does not depend on which block arrives last) with sc1 loads and runs the epilogue.  No second
//     launch, no memset: the reducer re-zeroes its ticket (guide: Guideline 16 / projection-GEMM recipe, write-through form);
//   * block -> (pixel tile, channel tile, K slice): where (channel tiles x slices) is a multiple of 8, all pixel tiles of one
//     (channel tile, slice) pair run on ONE XCD (blocks b, b + 8, ... share an XCD), so a layer's weights cross the fabric once
//     (layer4: 9.4 MB instead of 8 x 9.4 MB per conv).
#include "sd_common.h"
#include "sd_mfma.h"

namespace sd {

struct SbArgs {
    const void* x;        // NHWC [B][Hi][Wi][Ck], fp32 or bf16
    const void* w;        // [Nn][R*S][Ck]
    void* y;              // [M][Nn]
    const float* scale;   // per-channel multiplier (nullable)
    const float* shift;   // per-channel addend (nullable): bias or folded BatchNorm
    const void* res;      // residual [M][Nn], or [B][Ho/2][Wo/2][Nn] when res_up2 (nullable)
    float* slabs;         // [tile][slice][64][64] fp32 partial tiles (splits > 1)
    unsigned* tickets;    // [tiles] arrival counters: zero before the first launch, left zero by every launch
    int B, Hi, Wi, Ck, Ho, Wo, Nn, R, S, stride, pad;
    int relu, res_up2;
    int M, nk, per;       // nk = R*S*(Ck / KE) chunks of 128 bytes per row; a slice multiplies `per` consecutive chunks
    int m_tiles, n_tiles, splits, grouped;
};

__device__ __attribute__((aligned(128))) float g_sb_zero_line[64];   // zero-initialised: source of padded rows and past-the-end chunks

constexpr int SB_ST = 64 * 32;          // floats per operand stage: 64 rows x 128 bytes
constexpr int SB_TP = 36;               // row pitch (floats) of a wave's 32 x 32 transposition tile

__device__ __forceinline__ void sb_store16_sc1(float* dst, f32x4 v) {
    // write-through store (visible to every XCD once the wave's vmcnt has drained); s_nop: VMEM store data hazard inside asm
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" :: "v"(dst), "v"(v) : "memory");
}
__device__ __forceinline__ f32x4 sb_load16_sc1(const float* src) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(src) : "memory");
    return v;
}
#define SB_VM_WAIT4(a, b, c, d) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) :: "memory")

template <bool BF16>
__global__ __launch_bounds__(256, 2) void k_conv_fwd_sb(SbArgs p) {
    using T = typename std::conditional<BF16, uint16_t, float>::type;
    constexpr int KE = BF16 ? 64 : 32;     // K elements per 128-byte chunk row
    constexpr int VE = BF16 ? 8 : 4;       // elements per 16-byte slot
    __shared__ __attribute__((aligned(16))) float As0[SB_ST];
    __shared__ __attribute__((aligned(16))) float As1[SB_ST];
    __shared__ __attribute__((aligned(16))) float As2[SB_ST];
    __shared__ __attribute__((aligned(16))) float As3[SB_ST];
    __shared__ __attribute__((aligned(16))) float Bs0[SB_ST];
    __shared__ __attribute__((aligned(16))) float Bs1[SB_ST];
    __shared__ __attribute__((aligned(16))) float Bs2[SB_ST];
    __shared__ __attribute__((aligned(16))) float Bs3[SB_ST];
    __shared__ int orow[64];               // output pixel of each tile row, -1 = none
    __shared__ int rrow[64];               // its row in a half-size residual map (res_up2)
    __shared__ unsigned ticket_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const T* const px_ = reinterpret_cast<const T*>(p.x);
    const T* const pw_ = reinterpret_cast<const T*>(p.w);

    // ---- block -> (pixel tile, channel tile, K slice)
    int mt, nt, sl;
    if (p.grouped) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        mt = i % p.m_tiles;
        const int g = (i / p.m_tiles) * 8 + xcd;
        nt = g % p.n_tiles; sl = g / p.n_tiles;
    } else {
        const int t = xcd_remap(blockIdx.x, gridDim.x);      // XCD-contiguous pixel tiles: neighbours share input rows in L2
        sl = t % p.splits;
        const int u = t / p.splits;
        nt = u % p.n_tiles; mt = u / p.n_tiles;
    }
    const int m0 = mt * 64, n0 = nt * 64;
    const int kbeg = sl * p.per, nkl = min(p.per, p.nk - kbeg);
    const int ntap = p.R * p.S;
    int ld_c0, ld_r, ld_s;
    {
        const int cc = kbeg / ntap, tap = kbeg - cc * ntap;    // chunk index = channel chunk * taps + tap (taps innermost: L2 reuse)
        ld_c0 = cc * KE; ld_r = tap / p.S; ld_s = tap - ld_r * p.S;
    }

    if (tid < 64) {
        const int m = m0 + tid;
        int pix = -1, rr = -1;
        if (m < p.M) {
            const int ox = m % p.Wo, t = m / p.Wo, oy = t % p.Ho, b = t / p.Ho;
            pix = m;
            rr = (b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
        }
        orow[tid] = pix; rrow[tid] = rr;
    }

    // ---- staging: wave w owns A pieces 2w, 2w+1 and B pieces 2w, 2w+1 of every chunk (a piece = 8 rows x 128 B = one
    // wave-instruction of LDS-DMA: lane l -> row l/8, physical 16-byte slot l%8).  LDS image: row r keeps logical slot q at
    // physical slot q ^ ((r >> 1) & 7) (conflict-free ds_read_b128), so the swizzle goes on the SOURCE address.
    const int prow = lane >> 3, pslot = lane & 7;
    const T* const zsrc = reinterpret_cast<const T*>(g_sb_zero_line);
    const T* abase[2];
    int aty[2], atx[2], aq[2];
    const T* bbase[2];
    const int wk = ntap * p.Ck;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (wave * 2 + j) * 8 + prow;
        const int q = (pslot ^ ((row >> 1) & 7)) * VE;
        aq[j] = q;
        const int m = m0 + row;
        const bool ok = m < p.M;
        const int mm = ok ? m : 0;
        const int ox = mm % p.Wo, t = mm / p.Wo, oy = t % p.Ho, b = t / p.Ho;
        aty[j] = ok ? oy * p.stride - p.pad : -(1 << 28);       // rows past the end fail every range check
        atx[j] = ox * p.stride - p.pad;
        abase[j] = px_ + (int64_t)b * p.Hi * p.Wi * p.Ck + q;
        bbase[j] = pw_ + (int64_t)(n0 + row) * wk + q;
    }
    int issued = 0;
#define SB_ISSUE(AD, BD)                                                                           \
    {                                                                                              \
        if (issued < nkl) {                                                                        \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                        \
                const int ty = aty[j] + ld_r, tx = atx[j] + ld_s;                                  \
                const bool ok = (unsigned)ty < (unsigned)p.Hi && (unsigned)tx < (unsigned)p.Wi;    \
                const T* src = ok ? abase[j] + ((int64_t)(ty * p.Wi + tx) * p.Ck + ld_c0) : zsrc + aq[j]; \
                lds_dma16(src, (AD) + (wave * 2 + j) * 256);                                       \
            }                                                                                      \
            const int woff = (ld_r * p.S + ld_s) * p.Ck + ld_c0;                                   \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) lds_dma16(bbase[j] + woff, (BD) + (wave * 2 + j) * 256); \
            if (++ld_s >= p.S) { ld_s = 0; if (++ld_r >= p.R) { ld_r = 0; ld_c0 += KE; } }        \
        } else {      /* past the end: keep the per-iteration DMA count (the counted waits rely on it) */ \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) lds_dma16(zsrc + aq[j], (AD) + (wave * 2 + j) * 256); \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) lds_dma16(zsrc + aq[j], (BD) + (wave * 2 + j) * 256); \
        }                                                                                          \
        ++issued;                                                                                  \
    }

    // ---- wave tile: 32 (m) x 32 (n)
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int rd_swz = (fr >> 1) & 7;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // per-lane byte offsets of the four k-groups' 16-byte fragments inside a stage (A row wm*32 + fr, B row wn*32 + fr)
    uint32_t fo[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fo[ks] = (uint32_t)((((ks * 2 + fh) ^ rd_swz) << 2) * 4);
    const uint32_t a_row = (uint32_t)((wm * 32 + fr) * 32 * 4), b_row = (uint32_t)((wn * 32 + fr) * 32 * 4);

    SB_ISSUE(As0, Bs0)
    SB_ISSUE(As1, Bs1)
    SB_ISSUE(As2, Bs2)
    wait_vmcnt<8>();                                          // chunk 0 has landed (chunks 1, 2 may be in flight)
    __builtin_amdgcn_s_barrier();

#define SB_COMPUTE(AB, BB)                                                                         \
    {                                                                                              \
        const uint32_t ab = lds_addr(AB) + a_row, bb = lds_addr(BB) + b_row;                       \
        f32x4 a0 = lds_read128_async<0>(ab + fo[0]), b0 = lds_read128_async<0>(bb + fo[0]);        \
        f32x4 a1 = lds_read128_async<0>(ab + fo[1]), b1 = lds_read128_async<0>(bb + fo[1]);        \
        f32x4 a2 = lds_read128_async<0>(ab + fo[2]), b2 = lds_read128_async<0>(bb + fo[2]);        \
        f32x4 a3 = lds_read128_async<0>(ab + fo[3]), b3 = lds_read128_async<0>(bb + fo[3]);        \
        SD_LDS_WAIT2(6, a0, b0); SB_MFMA(a0, b0)                                                   \
        SD_LDS_WAIT2(4, a1, b1); SB_MFMA(a1, b1)                                                   \
        SD_LDS_WAIT2(2, a2, b2); SB_MFMA(a2, b2)                                                   \
        SD_LDS_WAIT2(0, a3, b3); SB_MFMA(a3, b3)                                                   \
    }
#define SB_MFMA(FA, FB)                                                                            \
    if (BF16) {                                                                                    \
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, FA), __builtin_bit_cast(bf16x8, FB), acc, 0, 0, 0); \
    } else {                                                                                       \
        _Pragma("unroll") for (int t = 0; t < 4; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(FA[t], FB[t], acc, 0, 0, 0); \
    }
    // one pipeline step, stage names literal: issue chunk kc+3 into the stage read at step kc-1, multiply chunk kc, then make sure
    // this wave's pieces of chunk kc+1 have landed (two younger chunks = 8 DMAs may stay in flight) before the barrier publishes them
#define SB_ITER(AC, BC, AN, BN_)                                                                   \
    {                                                                                              \
        SB_ISSUE(AN, BN_)                                                                          \
        SB_COMPUTE(AC, BC)                                                                         \
        wait_vmcnt_and_lds<8>();                                                                   \
        __builtin_amdgcn_s_barrier();                                                              \
        ++kc;                                                                                      \
    }
    int kc = 0;
    while (kc < nkl) {
        SB_ITER(As0, Bs0, As3, Bs3)
        if (kc < nkl) SB_ITER(As1, Bs1, As0, Bs0)
        if (kc < nkl) SB_ITER(As2, Bs2, As1, Bs1)
        if (kc < nkl) SB_ITER(As3, Bs3, As2, Bs2)
    }
    wait_vmcnt<0>();
#undef SB_ITER
#undef SB_COMPUTE
#undef SB_MFMA
#undef SB_ISSUE
    __syncthreads();                                          // every (past-the-end) DMA has landed: the stages are free

    // ---- the wave's 32 x 32 accumulator -> rows, through a wave-private LDS tile (C/D map: n = lane & 31,
    // m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)); afterwards lane -> (row it*8 + lane/8, four channels (lane%8)*4)
    float* const Tw = wave == 0 ? As0 : wave == 1 ? As1 : wave == 2 ? As2 : As3;
#pragma unroll
    for (int e = 0; e < 16; ++e) Tw[((e & 3) + 8 * (e >> 2) + 4 * fh) * SB_TP + fr] = acc[e];
    const int er = lane >> 3, c4 = (lane & 7) * 4;
    const int n = n0 + wn * 32 + c4;
    const int tile = mt * p.n_tiles + nt;

    if (p.splits > 1) {
        float* const slab = p.slabs + ((int64_t)tile * p.splits + sl) * 4096 + (wm * 32) * 64 + wn * 32 + c4;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + er;
            const float4 v = *reinterpret_cast<const float4*>(Tw + row * SB_TP + c4);
            f32x4 vv; vv[0] = v.x; vv[1] = v.y; vv[2] = v.z; vv[3] = v.w;
            sb_store16_sc1(slab + row * 64, vv);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
        __syncthreads();                                      // ... before ONE lane signals for the block
        if (tid == 0) ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (ticket_s != (unsigned)(p.splits - 1)) return;     // not the last slice of this tile to arrive
        if (tid == 0) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // state left zero
    }

    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n);
    if (p.shift) sh4 = *reinterpret_cast<const float4*>(p.shift + n);
    const float* const slab0 = p.slabs + (int64_t)tile * p.splits * 4096 + (wm * 32) * 64 + wn * 32 + c4;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + er, trow = wm * 32 + row;
        const int m = orow[trow];
        float4 v;
        if (p.splits > 1) {
            // sum of the slices in slice order, four sc1 loads in flight (every load of a handed-off byte is sc1: no acquire needed)
            f32x4 s; s[0] = s[1] = s[2] = s[3] = 0.f;
            const float* src = slab0 + row * 64;
            for (int k = 0; k < p.splits; k += 4) {
                const int k1 = min(k + 1, p.splits - 1), k2 = min(k + 2, p.splits - 1), k3 = min(k + 3, p.splits - 1);
                f32x4 v0 = sb_load16_sc1(src + (int64_t)k * 4096), v1 = sb_load16_sc1(src + (int64_t)k1 * 4096);
                f32x4 v2 = sb_load16_sc1(src + (int64_t)k2 * 4096), v3 = sb_load16_sc1(src + (int64_t)k3 * 4096);
                SB_VM_WAIT4(v0, v1, v2, v3);
                s += v0;
                if (k + 1 < p.splits) s += v1;
                if (k + 2 < p.splits) s += v2;
                if (k + 3 < p.splits) s += v3;
            }
            v = make_float4(s[0], s[1], s[2], s[3]);
        } else {
            v = *reinterpret_cast<const float4*>(Tw + row * SB_TP + c4);
        }
        if (m < 0) continue;
        v.x = v.x * sc4.x + sh4.x; v.y = v.y * sc4.y + sh4.y; v.z = v.z * sc4.z + sh4.z; v.w = v.w * sc4.w + sh4.w;
        if (p.res) {
            const int64_t rm = p.res_up2 ? (int64_t)rrow[trow] : (int64_t)m;
            if (BF16) {
                const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(p.res) + rm * p.Nn + n);
                v.x += bf2f((uint16_t)(r.x & 0xffff)); v.y += bf2f((uint16_t)(r.x >> 16));
                v.z += bf2f((uint16_t)(r.y & 0xffff)); v.w += bf2f((uint16_t)(r.y >> 16));
            } else {
                const float4 r = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.res) + rm * p.Nn + n);
                v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
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
        }
    }
}

// Decomposition (host): K slices so that tiles x slices is about one block per CU, at least two chunks per slice, at most 16 slabs
// for the reducer to read.
static bool sb_plan(SbArgs& a, const sd_conv_desc* d, bool bf16) {
    const int KE = bf16 ? 64 : 32;
    if (d->Cin % KE || d->Cout % 64) return false;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.nk = d->R * d->S * (d->Cin / KE);
    a.m_tiles = cdiv(a.M, 64); a.n_tiles = d->Cout / 64;
    const int tiles = a.m_tiles * a.n_tiles;
    int s = 1;
    if (tiles < 192) s = std::max(1, std::min(std::min(16, a.nk / 2), (256 + tiles / 2) / tiles));
    a.per = cdiv(a.nk, s);
    a.splits = cdiv(a.nk, a.per);
    a.grouped = (a.n_tiles * a.splits) % 8 == 0 && a.splits > 1;
    return true;
}

}  // namespace sd

using namespace sd;

extern "C" {

// Geometries the small-batch kernel takes: where the 128-row tile grid of sd_conv2d_fwd cannot fill the chip (< 256 tiles).
int sd_conv2d_fwd_sb_supported(const sd_conv_desc* d, int bf16) {
    if (!d || d->B <= 0 || d->Cin % (bf16 ? 64 : 32) || d->Cout % 64 || d->R != d->S || d->R < 1 || d->stride < 1) return 0;
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    const int BN = (d->Cout % 128 == 0) ? 128 : 64;
    return cdiv(M, 128) * (d->Cout / BN) < 256 ? 1 : 0;
}

size_t sd_conv2d_fwd_sb_workspace_bytes(const sd_conv_desc* d, int bf16) {
    SbArgs a{};
    if (!d || !sb_plan(a, d, bf16 != 0) || a.splits <= 1) return 0;
    return (size_t)a.m_tiles * a.n_tiles * a.splits * 4096 * sizeof(float);
}

size_t sd_conv2d_fwd_sb_state_bytes(const sd_conv_desc* d, int bf16) {
    SbArgs a{};
    if (!d || !sb_plan(a, d, bf16 != 0)) return 0;
    return align_up((size_t)a.m_tiles * a.n_tiles * sizeof(unsigned), 256);
}

int sd_conv2d_fwd_sb(const void* x, const void* w, void* y, const sd_conv_desc* d, const float* scale, const float* shift,
                     const void* residual, int res_up2, int relu, int bf16, void* workspace, size_t workspace_bytes, void* state,
                     size_t state_bytes, sd_stream_t stream) {
    SD_REQUIRE(d != nullptr && x && w && y, SD_ERR_INVALID, "sd_conv2d_fwd_sb: null pointer");
    SD_REQUIRE(d->B > 0 && d->Hi > 0 && d->Wi > 0 && d->R > 0 && d->S > 0 && d->stride > 0 && d->pad >= 0, SD_ERR_INVALID, "sd_conv2d_fwd_sb: bad sizes");
    SD_REQUIRE(d->Ho == (d->Hi + 2 * d->pad - d->R) / d->stride + 1 && d->Wo == (d->Wi + 2 * d->pad - d->S) / d->stride + 1, SD_ERR_INVALID,
               "sd_conv2d_fwd_sb: Ho/Wo do not match the convolution geometry");
    SD_REQUIRE((int64_t)d->B * d->Ho * d->Wo < (1ll << 31) && (int64_t)d->B * d->Hi * d->Wi * d->Cin < (1ll << 31), SD_ERR_INVALID,
               "sd_conv2d_fwd_sb: tensor too large for 32-bit element offsets (this is the small-batch kernel)");
    SbArgs a{};
    SD_REQUIRE(sb_plan(a, d, bf16 != 0), SD_ERR_INVALID, "sd_conv2d_fwd_sb: needs Cin %% %d == 0 and Cout %% 64 == 0 (got %d, %d)", bf16 ? 64 : 32,
               d->Cin, d->Cout);
    SD_REQUIRE(aligned16(x) && aligned16(w) && aligned16(y) && aligned16(scale) && aligned16(shift) && aligned16(residual) && aligned16(workspace),
               SD_ERR_ALIGN, "sd_conv2d_fwd_sb: pointers must be 16-byte aligned");
    SD_REQUIRE(!res_up2 || (residual && d->Ho % 2 == 0 && d->Wo % 2 == 0), SD_ERR_INVALID, "sd_conv2d_fwd_sb: res_up2 needs a residual and even Ho, Wo");
    if (a.splits > 1) {
        SD_REQUIRE(workspace && workspace_bytes >= sd_conv2d_fwd_sb_workspace_bytes(d, bf16), SD_ERR_WORKSPACE, "sd_conv2d_fwd_sb: workspace too small");
        SD_REQUIRE(state && state_bytes >= sd_conv2d_fwd_sb_state_bytes(d, bf16), SD_ERR_WORKSPACE, "sd_conv2d_fwd_sb: state buffer too small");
    }
    a.x = x; a.w = w; a.y = y; a.scale = scale; a.shift = shift; a.res = residual; a.relu = relu; a.res_up2 = res_up2;
    a.slabs = (float*)workspace; a.tickets = (unsigned*)state;
    const int blocks = a.m_tiles * a.n_tiles * a.splits;
    if (bf16) hipLaunchKernelGGL(k_conv_fwd_sb<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_conv_fwd_sb<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    SD_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
