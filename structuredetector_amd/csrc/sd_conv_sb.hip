// Small-batch inference forward convolution (BASELINE configs[1]: bs = 1, 512x512) for gfx950.
// Replaces the ATen / cuDNN convolutions behind `net(batch["image"])` in the reference's batch-1 evaluate loop
// (src/sdnet/cli/evaluate.py:34-45 -> src/sdnet/model/network.py:59-84) for the layers whose 128-row tile grid cannot fill
// 256 CUs.  fp32 (v_mfma_f32_32x32x2_f32) and bf16 (v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogue).
//
// Why a kernel of its own: at one image the trunk's GEMMs are 16384 x 64 x 576 ... 256 x 512 x 4608 (pixels x Cout x K): the
// 128 x 128 tiles of k_conv_igemm give 8 ... 128 blocks, so K was split 4 ... 36 ways into 288 unbalanced blocks whose 17-19 MB of
// partial tiles a second launch (k_splitk_reduce) summed: 26 + 6.7 us per conv for 7.7 us of MFMA work.  Here
//   * tile 64 (pixels) x 64 (channels), four waves of one 32 x 32 output tile each: 256 / 128 / 64 / 32 tiles for layer1 .. layer4,
//     so K splits 1 / 2 / 4 / 8 ways give 256 equal blocks (one per CU) of 18 chunks each; taken up to 512 tiles of 128 rows (bs ~16);
//   * four LDS stages of 16 KB (64 rows x 128 B per operand) filled by LDS-DMA FOUR chunks ahead, a counted vmcnt(8) + lgkmcnt(0)
//     wait and a bare s_barrier per chunk (a __syncthreads() would drain the DMAs in flight);
//   * software-pipelined steps: while the MFMAs of chunk i run from one register set, the fragments of chunk i+1 are read
//     (inline-asm ds_read_b128, the stage an immediate offset of ONE LDS array) into the other and the four LDS-DMA pieces of chunk
//     i+4 are issued one per MFMA shadow; the filter tap of every step is a compile-time literal (36-step unrolled body for 3x3
//     filters) and the per-row source pointers of all taps live in registers (padding = pointer into a zero region): a chunk's
//     addresses cost four 64-bit adds.  In-kernel clocks: 1253 cycles per chunk of 1024 MFMA cycles (each LDS-DMA instruction costs
//     the issuing wave ~35 of them: one wave per SIMD); DESIGN.md section 7 has the ablations;
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
    int M, nk, per;       // nk = R*S*(Ck / KE) chunks of 128 bytes per row; a slice multiplies `per` consecutive chunks (a multiple of R*S)
    int m_tiles, n_tiles, splits, grouped;
    int cper;             // channel chunks per slice (per = R*S*cper)
    unsigned mg_wo, mg_ho, mg_mt, mg_nt, mg_sp;   // ceil(2^32 / d) for d = Wo, Ho, m_tiles, n_tiles, splits: exact quotients by one multiply-high while n * d < 2^32 (d = 1: 0xffffffff, exact for n < 2^31 ... see sb_magic)
};

// zero-initialised: source of padded rows and past-the-end chunks (a padded row's pointer advances by the channel offset like a real one:
// Ck * element size <= 4 KB)
__device__ __attribute__((aligned(128))) float g_sb_zero[1024 + 32];

#ifdef SD_SB_TRACE
// timing experiment (make SUFFIX=_sbtrace EXTRA=-DSD_SB_TRACE): 100 MHz timestamps of every block's phases
__device__ unsigned long long g_sb_trace[1024][8];
#define SB_T(i) if (tid == 0 && blockIdx.x < 1024) g_sb_trace[blockIdx.x][i] = wall_clock64();
#else
#define SB_T(i)
#endif

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
// end of a pipeline step: this wave's pieces of the chunk after next have landed (8 younger DMAs may be in flight) and the fragments
// read during the step are in their registers
#define SB_STEP_WAIT(f) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(f[4]), "+v"(f[5]), "+v"(f[6]), "+v"(f[7]) :: "memory")

// n / d with magic = ceil(2^32 / d), exact while n * d < 2^32; d = 1 is encoded as magic 0 (2^32 does not fit)
__device__ __forceinline__ int sb_div(int n, unsigned magic) { return magic ? (int)__umulhi((unsigned)n, magic) : n; }

// Reducer: sum of the slices' slab rows in slice order.  ITS rows of this lane x KP slices are in flight at once (ITS * KP = 16 or 32 sc1
// loads of 16 bytes; every load of a handed-off byte is sc1, so no acquire is needed), then the next group of rows.
template <int ITS, int KP>
__device__ __forceinline__ void sb_reduce(const float* slab0, int splits, int er, f32x4 (&sum)[4]) {
#pragma unroll
    for (int g = 0; g < 4 / ITS; ++g) {
        f32x4 v[ITS * KP];
#pragma unroll
        for (int i = 0; i < ITS; ++i)
#pragma unroll
            for (int k = 0; k < KP; ++k)
                v[i * KP + k] = sb_load16_sc1(slab0 + ((g * ITS + i) * 8 + er) * 64 + (int64_t)min(k, splits - 1) * 4096);
#pragma unroll
        for (int q = 0; q < ITS * KP; q += 4) SB_VM_WAIT4(v[q], v[q + 1], v[q + 2], v[q + 3]);
#pragma unroll
        for (int i = 0; i < ITS; ++i) {
            f32x4 s = v[i * KP];
#pragma unroll
            for (int k = 1; k < KP; ++k)
                if (k < splits) s += v[i * KP + k];
            sum[g * ITS + i] = s;
        }
    }
}

// NTAP = R * S as a compile-time constant (9 or 1): the filter tap of every pipeline step is then a literal, and the per-row source
// pointers of all taps (padding resolved: a padded tap points at the zero region) live in registers -- a chunk's four LDS-DMA
// addresses cost four 64-bit adds.  NTAP = 0: any R == S filter, tap arithmetic at run time.
template <bool BF16, int NTAP>
__global__ __launch_bounds__(256, 2) void k_conv_fwd_sb(SbArgs p) {
    using T = typename std::conditional<BF16, uint16_t, float>::type;
    constexpr int KE = BF16 ? 64 : 32;     // K elements per 128-byte chunk row
    constexpr int VE = BF16 ? 8 : 4;       // elements per 16-byte slot
    constexpr int NT = NTAP ? NTAP : 1;
    // ONE LDS object: stage i of A at float offset i * SB_ST, of B at (4 + i) * SB_ST, so that a stage is an immediate offset of the
    // fragment reads (inline-asm ds_read_b128: the compiler's wait insertion never sees them beside the LDS-DMA in flight)
    __shared__ __attribute__((aligned(16))) float lds[8 * SB_ST];
    __shared__ int orow[64];               // output pixel of each tile row, -1 = none
    __shared__ int rrow[64];               // its row in a half-size residual map (res_up2)
    __shared__ unsigned ticket_s;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    SB_T(0)
    const T* const px_ = reinterpret_cast<const T*>(p.x);
    const T* const pw_ = reinterpret_cast<const T*>(p.w);

    // ---- block -> (pixel tile, channel tile, K slice); quotients by multiply-high with host-made reciprocals (scalar unit)
    int mt, nt, sl;
    if (p.grouped) {
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        const int gq = sb_div(i, p.mg_mt);
        mt = i - gq * p.m_tiles;
        const int g = gq * 8 + xcd;
        sl = sb_div(g, p.mg_nt); nt = g - sl * p.n_tiles;
    } else {
        const int t = xcd_remap(blockIdx.x, gridDim.x);      // XCD-contiguous pixel tiles: neighbours share input rows in L2
        const int u = sb_div(t, p.mg_sp);
        sl = t - u * p.splits;
        mt = sb_div(u, p.mg_nt); nt = u - mt * p.n_tiles;
    }
    const int m0 = mt * 64, n0 = nt * 64;
    const int kbeg = sl * p.per, nkl = min(p.per, p.nk - kbeg);
#ifdef SD_SB_TRACE
    if (tid == 0 && blockIdx.x < 1024 && nkl > -5) g_sb_trace[blockIdx.x][7] = wall_clock64();     // (after the first uses of the kernel arguments)
#endif
    const int ntap = p.R * p.S;
    int ic0 = sl * p.cper * KE;                                // channel offset of the next chunk to issue (a slice = whole channel chunks)
    int itap = 0;                                              // NTAP == 0: its tap

    if (tid < 64) {
        const int m = m0 + tid;
        int pix = -1, rr = -1;
        if (m < p.M) {
            const int t = sb_div(m, p.mg_wo), ox = m - t * p.Wo, b = sb_div(t, p.mg_ho), oy = t - b * p.Ho;
            pix = m;
            rr = (b * (p.Ho >> 1) + (oy >> 1)) * (p.Wo >> 1) + (ox >> 1);
        }
        orow[tid] = pix; rrow[tid] = rr;
    }

    // ---- staging: wave w owns A pieces 2w, 2w+1 and B pieces 2w, 2w+1 of every chunk (a piece = 8 rows x 128 B = one
    // wave-instruction of LDS-DMA: lane l -> row l/8, physical 16-byte slot l%8).  LDS image: row r keeps logical slot q at
    // physical slot q ^ ((r >> 1) & 7) (conflict-free ds_read_b128), so the swizzle goes on the SOURCE address.
    const int prow = lane >> 3, pslot = lane & 7;
    const T* const zsrc = reinterpret_cast<const T*>(g_sb_zero);
    const T* aptr[2][NT];                  // NTAP > 0: source of (row j, tap) at channel 0, or the zero region
    const T* abase[2];                     // NTAP == 0: image base + slot; coordinates checked per chunk
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
        const int t = sb_div(mm, p.mg_wo), ox = mm - t * p.Wo, b = sb_div(t, p.mg_ho), oy = t - b * p.Ho;
        aty[j] = ok ? oy * p.stride - p.pad : -64;             // rows past the end fail every range check (R <= 32)
        atx[j] = ox * p.stride - p.pad;
        abase[j] = px_ + (int64_t)b * p.Hi * p.Wi * p.Ck + q;
        bbase[j] = pw_ + (int64_t)(n0 + row) * wk + q;
        if (NTAP) {
            // tap (r, s): pointer of tap (0, 0) + a wave-uniform offset; padding by three row checks x three column checks
            const T* const p00 = abase[j] + (int64_t)(aty[j] * p.Wi + atx[j]) * p.Ck;
            bool vr[3], vc[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) { vr[r] = (unsigned)(aty[j] + r) < (unsigned)p.Hi; vc[r] = (unsigned)(atx[j] + r) < (unsigned)p.Wi; }
#pragma unroll
            for (int tp = 0; tp < NT; ++tp) {
                const int r = NTAP == 9 ? tp / 3 : 0, s_ = NTAP == 9 ? tp % 3 : 0;
                aptr[j][tp] = (vr[r] && vc[s_]) ? p00 + (int64_t)(r * p.Wi + s_) * p.Ck : zsrc + q;
            }
        }
    }
    const int cmax = p.Ck - KE;
    float* const ldsA = lds + wave * 512;                      // this wave's first piece inside stage 0 of A (B: + 4 * SB_ST)
    // The next chunk (filter tap TP when NTAP > 0) into stage ST, one LDS-DMA piece at a time (J = 0, 1: the wave's A pieces, 2, 3: its B
    // pieces) so that the pipeline steps can put ONE piece into each MFMA shadow (one wave gets a piece out per ~66 cycles).  Every step
    // issues its four pieces, also past the end of the slice (the counted waits rely on it, and a conditional issue costs selects or
    // branches between the MFMAs): the channel offset is clamped to the last chunk, so those pieces re-read valid data that no MFMA uses.
#define SB_ISSUE_J(TP, ST, J)                                                                      \
    {                                                                                              \
        float* const dst = ldsA + (ST) * SB_ST + ((J) >> 1) * 4 * SB_ST + ((J) & 1) * 256;         \
        if (NTAP) {                                                                                \
            if ((J) < 2) lds_dma16(aptr[(J) & 1][(TP) % NT] + ic0, dst);                           \
            else lds_dma16(bbase[(J) & 1] + (((TP) % NT) * p.Ck + ic0), dst);                      \
            if ((J) == 3 && (TP) % NT == NT - 1) ic0 = min(ic0 + KE, cmax);                        \
        } else {                                                                                   \
            if ((J) < 2) {                                                                         \
                const int r = itap / p.S, s_ = itap - r * p.S;                                     \
                const int ty = aty[(J) & 1] + r, tx = atx[(J) & 1] + s_;                           \
                const bool in = (unsigned)ty < (unsigned)p.Hi && (unsigned)tx < (unsigned)p.Wi;    \
                lds_dma16(in ? abase[(J) & 1] + ((int64_t)(ty * p.Wi + tx) * p.Ck + ic0) : zsrc + aq[(J) & 1], dst); \
            } else lds_dma16(bbase[(J) & 1] + (itap * p.Ck + ic0), dst);                           \
            if ((J) == 3 && ++itap == ntap) { itap = 0; ic0 = min(ic0 + KE, cmax); }               \
        }                                                                                          \
    }
#define SB_ISSUE(TP, ST) SB_ISSUE_J(TP, ST, 0) SB_ISSUE_J(TP, ST, 1) SB_ISSUE_J(TP, ST, 2) SB_ISSUE_J(TP, ST, 3)

    // ---- wave tile: 32 (m) x 32 (n)
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 31, fh = lane >> 5;
    const int rd_swz = (fr >> 1) & 7;
    // two accumulators, alternating k-steps: a dependent v_mfma issues a few cycles after its predecessor has left the pipe (measured:
    // 1253 instead of 1024 cycles per chunk with one accumulator), two independent chains keep it full; summed once at the end
    f32x16 acc, acc1;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = acc1[e] = 0.f;
    // LDS byte addresses (stage 0) of the four k-groups' 16-byte fragments: A row wm*32 + fr, B row wn*32 + fr
    uint32_t fa[4], fb[4];
    {
        const uint32_t base = lds_addr(lds);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const uint32_t slot = (uint32_t)((((ks * 2 + fh) ^ rd_swz) << 2) * 4);
            fa[ks] = base + (uint32_t)((wm * 32 + fr) * 128) + slot;
            fb[ks] = base + (uint32_t)(4 * SB_ST * 4 + (wn * 32 + fr) * 128) + slot;
        }
    }
    // fragment registers of two chunks: set 0 / 1, [0..3] = A k-groups, [4..7] = B k-groups
    f32x4 fx[2][8];
#define SB_READ1(SET, ST, KS)                                                                       \
    { fx[SET][KS] = lds_read128_async<(ST) * SB_ST * 4>(fa[KS]); fx[SET][4 + (KS)] = lds_read128_async<(ST) * SB_ST * 4>(fb[KS]); }
#define SB_READ(SET, ST) SB_READ1(SET, ST, 0) SB_READ1(SET, ST, 1) SB_READ1(SET, ST, 2) SB_READ1(SET, ST, 3)
    // MFMA(s) of k-group KS: fp32 = four dependent v_mfma_f32_32x32x2_f32 (T = 0 .. 3 singly), bf16 = one v_mfma_f32_32x32x16_bf16 (at T = 0)
#define SB_MFMA1(SET, KS, T_)                                                                      \
    if (BF16) {                                                                                    \
        if ((T_) == 0) {                                                                           \
            if ((KS) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fx[SET][KS]), __builtin_bit_cast(bf16x8, fx[SET][4 + (KS)]), acc1, 0, 0, 0); \
            else acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fx[SET][KS]), __builtin_bit_cast(bf16x8, fx[SET][4 + (KS)]), acc, 0, 0, 0); \
        }                                                                                          \
    } else if ((T_) & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(fx[SET][KS][T_], fx[SET][4 + (KS)][T_], acc1, 0, 0, 0); \
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fx[SET][KS][T_], fx[SET][4 + (KS)][T_], acc, 0, 0, 0);
#define SB_PIN __builtin_amdgcn_sched_barrier(0);

    // ---- prologue: chunks 0 .. 3 on their way, chunk 0 in registers, chunk 1 published
    SB_ISSUE(0, 0)
    SB_ISSUE(1, 1)
    SB_ISSUE(2, 2)
    SB_ISSUE(3, 3)
    SB_T(1)
    wait_vmcnt<12>();                                         // chunk 0 has landed
    __builtin_amdgcn_s_barrier();
    SB_READ(0, 0)
    SB_STEP_WAIT(fx[0]);                                      // chunk 0 in registers, this wave's pieces of chunk 1 landed
    __builtin_amdgcn_s_barrier();                             // chunk 1 published; every wave is done with stage 0
    SB_T(2)
#ifdef SD_SB_TRACE
    const long long cyc0 = clock64();
#endif

    // One pipeline step (I literal): multiply chunk I from register set I & 1 while the fragments of chunk I+1 are read into the other
    // set and chunk I+4 is issued into the stage chunk I came from (free: every wave had chunk I in registers before the last barrier).
    // The LDS reads and the LDS-DMA issue sit between the MFMAs (a dependent MFMA issues 64 cycles after its predecessor: what stands
    // between them runs in that shadow); sched_barrier pins that order.
#ifndef SD_SB_ABL
#define SD_SB_ABL 0     // timing-only ablations of the pipeline step (WRONG RESULTS): 1 no LDS-DMA, 2 no barrier, 3 no fragment reads, 4 no MFMA
#endif
#define SB_ABL_DMA(x) if (SD_SB_ABL != 1) { x }
#define SB_ABL_RD(x) if (SD_SB_ABL != 3) { x }
#define SB_ABL_MM(x) if (SD_SB_ABL != 4) { x }
#define SB_STEP(I)                                                                                 \
    {                                                                                              \
        SB_ABL_MM(SB_MFMA1((I) & 1, 0, 0)) SB_PIN SB_ABL_RD(SB_READ1(((I) + 1) & 1, ((I) + 1) & 3, 0)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 0, 1)) SB_PIN SB_ABL_RD(SB_READ1(((I) + 1) & 1, ((I) + 1) & 3, 1)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 0, 2)) SB_PIN SB_ABL_RD(SB_READ1(((I) + 1) & 1, ((I) + 1) & 3, 2)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 0, 3)) SB_PIN SB_ABL_RD(SB_READ1(((I) + 1) & 1, ((I) + 1) & 3, 3)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 1, 0)) SB_PIN SB_ABL_DMA(SB_ISSUE_J((I) + 4, (I) & 3, 0)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 1, 1) SB_MFMA1((I) & 1, 1, 2) SB_MFMA1((I) & 1, 1, 3))         \
        SB_ABL_MM(SB_MFMA1((I) & 1, 2, 0)) SB_PIN SB_ABL_DMA(SB_ISSUE_J((I) + 4, (I) & 3, 1)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 2, 1)) SB_PIN SB_ABL_DMA(SB_ISSUE_J((I) + 4, (I) & 3, 2)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 2, 2) SB_MFMA1((I) & 1, 2, 3))                                 \
        SB_ABL_MM(SB_MFMA1((I) & 1, 3, 0)) SB_PIN SB_ABL_DMA(SB_ISSUE_J((I) + 4, (I) & 3, 3)) SB_PIN  \
        SB_ABL_MM(SB_MFMA1((I) & 1, 3, 1) SB_MFMA1((I) & 1, 3, 2) SB_MFMA1((I) & 1, 3, 3))         \
        SB_PIN                                                                                     \
        SB_STEP_WAIT(fx[((I) + 1) & 1]);                                                           \
        if (SD_SB_ABL != 2) __builtin_amdgcn_s_barrier();                                          \
        if (++kc >= nkl) break;                                                                    \
    }
#define SB_STEP4(I) SB_STEP(I) SB_STEP((I) + 1) SB_STEP((I) + 2) SB_STEP((I) + 3)
    int kc = 0;
    for (;;) {
        if (NTAP == 9) {
            SB_STEP4(0) SB_STEP4(4) SB_STEP4(8) SB_STEP4(12) SB_STEP4(16) SB_STEP4(20) SB_STEP4(24) SB_STEP4(28) SB_STEP4(32)
        } else {
            SB_STEP4(0)
        }
    }
    wait_vmcnt<0>();
#undef SB_STEP4
#undef SB_STEP
#undef SB_MFMA1
#undef SB_READ
#undef SB_READ1
#undef SB_ISSUE
#undef SB_ISSUE_J
#undef SB_PIN
    __syncthreads();                                          // every (past-the-end) DMA has landed: the stages are free
    SB_T(3)
#ifdef SD_SB_TRACE
    if (tid == 0 && blockIdx.x < 1024) g_sb_trace[blockIdx.x][4] = (unsigned long long)(clock64() - cyc0);      // (splits == 1 only: slot 4 is rewritten below)
#endif

    // ---- the wave's 32 x 32 accumulator -> rows, through a wave-private LDS tile (C/D map: n = lane & 31,
    // m = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5)); afterwards lane -> (row it*8 + lane/8, four channels (lane%8)*4)
    float* const Tw = lds + wave * SB_ST;
#pragma unroll
    for (int e = 0; e < 16; ++e) Tw[((e & 3) + 8 * (e >> 2) + 4 * fh) * SB_TP + fr] = acc[e] + acc1[e];
    const int er = lane >> 3, c4 = (lane & 7) * 4;
    const int n = n0 + wn * 32 + c4;
    const int tile = mt * p.n_tiles + nt;
    f32x4 out[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const float4 v = *reinterpret_cast<const float4*>(Tw + (it * 8 + er) * SB_TP + c4);
        out[it][0] = v.x; out[it][1] = v.y; out[it][2] = v.z; out[it][3] = v.w;
    }

    if (p.splits > 1) {
        float* const slab = p.slabs + ((int64_t)tile * p.splits + sl) * 4096 + (wm * 32) * 64 + wn * 32 + c4;
#pragma unroll
        for (int it = 0; it < 4; ++it) sb_store16_sc1(slab + (it * 8 + er) * 64, out[it]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave drains its write-through stores ...
        __syncthreads();                                      // ... before ONE lane signals for the block
        SB_T(4)
        if (tid == 0) ticket_s = __hip_atomic_fetch_add(p.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        SB_T(5)
        if (ticket_s != (unsigned)(p.splits - 1)) return;     // not the last slice of this tile to arrive
        if (tid == 0) __hip_atomic_store(p.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // state left zero
        const float* const slab0 = p.slabs + (int64_t)tile * p.splits * 4096 + (wm * 32) * 64 + wn * 32 + c4;
        if (p.splits <= 4) sb_reduce<4, 4>(slab0, p.splits, er, out);
        else if (p.splits <= 8) sb_reduce<4, 8>(slab0, p.splits, er, out);      // 32 loads in flight (the fragment registers are dead)
        else sb_reduce<2, 16>(slab0, p.splits, er, out);
    }

    float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.scale) sc4 = *reinterpret_cast<const float4*>(p.scale + n);
    if (p.shift) sh4 = *reinterpret_cast<const float4*>(p.shift + n);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int trow = wm * 32 + it * 8 + er;
        const int m = orow[trow];
        if (m < 0) continue;
        float4 v = make_float4(out[it][0], out[it][1], out[it][2], out[it][3]);
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
#ifdef SD_SB_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SB_T(6)
#endif
}

// Decomposition (host): K slices so that tiles x slices is about one block per CU, at least two chunks per slice, at most 16 slabs
// for the reducer to read.
static bool sb_plan(SbArgs& a, const sd_conv_desc* d, bool bf16) {
    const int KE = bf16 ? 64 : 32;
    if (d->Cin % KE || d->Cout % 64 || d->Cin * (bf16 ? 2 : 4) > 4096 || d->R != d->S || d->R > 32) return false;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.Ck = d->Cin; a.Ho = d->Ho; a.Wo = d->Wo; a.Nn = d->Cout; a.R = d->R; a.S = d->S;
    a.stride = d->stride; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    const int ntap = d->R * d->S, kchunks = d->Cin / KE;
    a.nk = ntap * kchunks;
    a.m_tiles = cdiv(a.M, 64); a.n_tiles = d->Cout / 64;
    const int tiles = a.m_tiles * a.n_tiles;
    int s = 1;
    // about one block per CU (measured: 512 blocks of half the work are 5 % slower, the combine grows with the slices), at least two
    // chunks per slice, at most 16 slabs for the reducer
#ifndef SD_SB_TARGET
#define SD_SB_TARGET 256     // (timing experiments: blocks per launch the K split aims at)
#endif
    if (tiles < SD_SB_TARGET * 3 / 4) s = std::max(1, std::min(std::min(16, a.nk / 2), (SD_SB_TARGET + tiles / 2) / tiles));
    if (bf16 && tiles >= 128) s = 1;                         // bf16: the loop is a fraction of a split's combine (measured: 1.8 vs 3.4 us)
    a.cper = cdiv(kchunks, std::min(s, kchunks));            // whole channel chunks per slice: every slice starts at filter tap 0
    a.per = ntap * a.cper;
    a.splits = cdiv(a.nk, a.per);
    a.grouped = (a.n_tiles * a.splits) % 8 == 0 && a.splits > 1;
    // quotients by multiply-high: exact while n * d < 2^32 (pixels: n < M <= 2^20, 2 <= d <= 2^12; blocks: n, d <= 2^15)
    if (a.M > (1 << 20) || d->Wo < 2 || d->Wo > 4096 || d->Ho < 2 || d->Ho > 4096 || (int64_t)tiles * a.splits > (1 << 15)) return false;
    auto magic = [](int dd) { return dd <= 1 ? 0u : (unsigned)(((1ull << 32) + dd - 1) / dd); };
    a.mg_wo = magic(d->Wo); a.mg_ho = magic(d->Ho); a.mg_mt = magic(a.m_tiles); a.mg_nt = magic(a.n_tiles); a.mg_sp = magic(a.splits);
    return true;
}

}  // namespace sd

using namespace sd;

extern "C" {

// Geometries the small-batch kernel takes: where the 128-row tile grid of sd_conv2d_fwd does not fill the chip twice (< 512 tiles).
int sd_conv2d_fwd_sb_supported(const sd_conv_desc* d, int bf16) {
    if (!d || d->B <= 0 || d->Cin % (bf16 ? 64 : 32) || d->Cout % 64 || d->R != d->S || d->R < 1 || d->stride < 1) return 0;
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    const int BN = (d->Cout % 128 == 0) ? 128 : 64;
    SbArgs a{};
    // < 512 tiles of 128 rows (two resident blocks per CU): measured on the whole eval forward, bs = 2 .. 16 (tools/bs1_bench.py): 256 -> 512
    // is 2-4 % faster at every batch, 1024 the same, 4096 mixed
    return (cdiv(M, 128) * (d->Cout / BN) < 512 && M <= (1 << 20) && sb_plan(a, d, bf16 != 0)) ? 1 : 0;
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
    SD_REQUIRE(sb_plan(a, d, bf16 != 0), SD_ERR_INVALID, "sd_conv2d_fwd_sb: needs Cin %% %d == 0, Cin <= %d, Cout %% 64 == 0, at most 2^20 output pixels and "
               "2 <= Ho, Wo <= 4096 (got Cin %d, Cout %d, %d x %d x %d)", bf16 ? 64 : 32, bf16 ? 2048 : 1024, d->Cin, d->Cout, d->B, d->Ho, d->Wo);
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
    const int ntap = d->R * d->S;
#define SB_LAUNCH(BF, NTAP) hipLaunchKernelGGL((k_conv_fwd_sb<BF, NTAP>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a)
    if (bf16) { if (ntap == 9) SB_LAUNCH(true, 9); else if (ntap == 1) SB_LAUNCH(true, 1); else SB_LAUNCH(true, 0); }
    else { if (ntap == 9) SB_LAUNCH(false, 9); else if (ntap == 1) SB_LAUNCH(false, 1); else SB_LAUNCH(false, 0); }
#undef SB_LAUNCH
    SD_LAUNCH_CHECK();
    return 0;
}

#ifdef SD_SB_TRACE
int sd_debug_sb_trace(unsigned long long* out, int blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(sd::g_sb_trace), sizeof(unsigned long long) * 8 * (blocks < 1024 ? blocks : 1024));
}
#endif

}  // extern "C"
