// Swapped-operand bf16 row-stream kernel of the 64 -> 64 channel 3x3 / stride 1 layers (layer1: src/sdnet/model/network.py:47, torchvision
// resnet34 layer1 -- forward and flipped-tap data-gradient), a translation unit of its own since round 5: its inline-asm MFMAs make its ISA
// worth checking (tools/check_rows16_isa.py compiles THIS file in seconds), and sd_conv.hip is large enough.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/sdnet_hip.h"
#include "sd_conv_rows.h"

namespace sd {

__device__ __attribute__((aligned(128))) static float g_zero_line[64];   // zero-initialised: source of padded (out-of-image) rows and pieces

// ---------------------------------------------------------------------------------------------
// k_conv3x3_c64_rows16_bf16 (round 5): the row stream of k_conv3x3_c64_rows_bf16 with the MFMA operands SWAPPED and the epilogue in
// registers.  The 32x32x16 form traced at 4935 cycles per row for 2304 cycles of MFMA (`profiles/r05_rows_bf16_ablations.txt`): 2950 in the
// MFMA stream even with nothing between the MFMAs, +350 for the LDS-DMA issue, +900 for the row form of the previous row (scratch reads,
// residual, rounding, statistics, stores), +640 for the accumulators' trip into the LDS scratch while the matrix pipe idles, +80 barrier.
// Here:
//   * D = W x X^T on `v_mfma_f32_16x16x32_bf16`: the weights are the A operand (72 fragments of 16 channels x 32 k in registers, as before),
//     the pixels the B operand (one `ds_read_b128` per 16 pixels x 32 k from the same ring; 36 reads per row as before), so a lane ends up
//     with the channels of ONE pixel: rows 4 g + e of a 16-channel tile, g = lane >> 4.  The weight ROWS are permuted at load time (tile
//     ct, row m <-> channel 32 (ct >> 1) + 8 (m >> 2) + 4 (ct & 1) + (m & 3)), which makes the eight values a lane holds in tiles 2 j and
//     2 j + 1 eight CONSECUTIVE channels 32 j + 8 g ... of its pixel: affine, residual, ReLU, rounding and the BatchNorm sums happen in
//     registers and the 16-byte store goes straight out (the four lane groups of a pixel write 64 contiguous bytes per instruction).  No
//     scratch, no transposition, no row form: ~35 vector instructions per item instead of ~75 + 16 LDS operations;
//   * TWO accumulator sets (64 AGPRs; 48 of the 72 weight fragments in AGPRs, 24 in VGPRs): row y accumulates into one set while the
//     epilogue of row y - 1 reads the other, cut into eight pieces between the MFMAs of steps 3 .. 10 -- the matrix pipe never waits for an
//     epilogue; per row one counted `vmcnt`, one `s_barrier`;
//   * 16x16x32 is also the shape the chip holds a higher clock on under its power limit (`tools/micro/mfma_bf16_shape.hip`).
// Same ring, LDS-DMA pieces, units and statistics layout as k_conv3x3_c64_rows_bf16; one instantiation per epilogue kind (plain / affine /
// statistics, with or without a residual: the 32 auxiliary registers of the affine or of the sums, and the residual's 16, only where used).  sd_set_option("conv_rows16", 0) = the 32x32x16 kernel.
// ---------------------------------------------------------------------------------------------
#ifndef SD_R16_WA1
#define SD_R16_WA1 0
#endif
#ifndef SD_R16_WA2
#define SD_R16_WA2 0
#endif
constexpr int R16_RES_ROW_BYTES = 128 * 128;
constexpr int R16_LDS_BYTES = RS_NR * RS_ROW_BYTES + 4 * 128 * 4 + 1024 + 3 * R16_RES_ROW_BYTES;     // input ring, statistics, dump line, residual ring
constexpr int R16_W_AGPR = 40;      // 64 accumulator + 160 weight AGPRs; with 46 the allocator kept 40 there anyway and copied the other six per use
template <bool W_IN_AGPR, bool ZERO>
__device__ __forceinline__ void r16_mfma(f32x4& acc, const bf16x8& w, const f32x4& px) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(SD_R16_BUILTIN)
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, px), ZERO ? zero : acc, 0, 0, 0);
#elif defined(__HIP_DEVICE_COMPILE__)
    if constexpr (ZERO) {
        if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "a"(w), "v"(px));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(acc) : "v"(w), "v"(px));
    } else {
        if constexpr (W_IN_AGPR) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(px));
        else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(px));
    }
#else
    (void)acc; (void)w; (void)px;
#endif
}

// KIND 0: plain (data-gradient), 1: affine epilogue (inference), 2: BatchNorm statistics (training forward); RES: a residual tensor is added
template <int KIND, bool RES>
__global__ __launch_bounds__(256) void k_conv3x3_c64_rows16_bf16(RowsArgs p) {
    extern __shared__ __attribute__((aligned(16))) float rs_lds[];
    char* const ring = reinterpret_cast<char*>(rs_lds);
    float* const sred = reinterpret_cast<float*>(ring + RS_NR * RS_ROW_BYTES);                     // [4][128]
    float* const dump = sred + 4 * 128;                                                            // 1 KB: where a piece that does not exist lands
    char* const rres = reinterpret_cast<char*>(dump + 256);                                        // [3][128 pixels][128 bytes] (RES)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g = lane >> 4;
    const uint16_t* const zero_ = reinterpret_cast<const uint16_t*>(g_zero_line);
    const int H = p.H, W = p.W;
    constexpr bool has_res = RES, has_stat = KIND == 2, has_affine = KIND == 1;
    // weight fragments held in AGPRs (beside the 64 accumulator AGPRs); the kinds with 32 auxiliary VGPRs move a few more there
    constexpr int R16_WA = R16_W_AGPR + (KIND == 2 ? SD_R16_WA2 : (KIND == 1 ? SD_R16_WA1 : 0));
    const bool relu = p.relu != 0;
    const uint16_t* const xg = p.x;
    const uint16_t* const resg = p.res;
    uint16_t* const yg = p.y;

    // ---- every weight fragment of the layer: lane (row m = n, k group g) of tile ct holds w[chan(ct, n)][tap][32 ks + 8 g .. + 7]
    bf16x8 Wf[72];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int ch = 32 * (ct >> 1) + 8 * (n >> 2) + 4 * (ct & 1) + (n & 3);
                Wf[(t * 2 + ks) * 4 + ct] = *reinterpret_cast<const bf16x8*>(p.w + (ch * 9 + (p.flip ? 8 - t : t)) * 64 + ks * 32 + g * 8);
            }
    // The fragments the MFMAs take from AGPRs are RE-DEFINED by an (empty) asm statement with an "=a" result tied to the loaded value: from
    // here on they are values of the AGPR class and stay there.  Left as loaded (a VGPR-class value merely USED under an "a" constraint) the
    // register allocator is free to keep them in VGPRs and copy them into one scratch AGPR quad in front of every MFMA pair -- it did, in the
    // build whose LDS-DMA sat late in the row: 4 v_accvgpr_write per pair, and, since the asm MFMAs are invisible to the hazard recogniser,
    // without the wait states between that write and the MFMA reading it: wrong, run-to-run different sums in the first MFMA behind each
    // copy (pixel tile 0 of channel tiles 1 .. 3), `profiles/r05_rows16_agpr_copy_hazard.txt`.
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
    for (int i = 0; i < R16_WA; ++i) {
        bf16x8 pinned;
        asm volatile("" : "=a"(pinned) : "0"(Wf[i]));
        Wf[i] = pinned;
    }
#endif
    // aux: scale / shift of the lane's 16 channels (32 j + 8 g + i at [8 j + i]) -- or, in a statistics launch, their running sums
    float aux[KIND == 0 ? 1 : 32];
#pragma unroll
    for (int k = 0; k < (KIND == 0 ? 1 : 32); ++k) aux[k] = (k < 16 && has_affine) ? 1.f : 0.f;
    if constexpr (has_affine) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (p.scale) aux[8 * j + i] = p.scale[32 * j + 8 * g + i];
                if (p.shift) aux[16 + 8 * j + i] = p.shift[32 * j + 8 * g + i];
            }
    }
    // pixel fragment offsets inside a ring row: output pixel 32 wave + 16 pt + n, tap column s -> ring pixel + s (ring pixel 0 = image
    // column x0 - 1).  Pixel tile pt = 1 is 16 pixels on: the same swizzle ((pixel >> 1) & 7), 2048 bytes further -- an immediate; k half
    // ks = 1 is slot (4 + g) ^ swizzle = the ks = 0 offset with bit 6 flipped: one v_xor per read instead of nine more registers.
    uint32_t poff[3];
#pragma unroll
    for (int s_ = 0; s_ < 3; ++s_) {
        const int pxr = wave * 32 + n + s_;
        poff[s_] = (uint32_t)pxr * 128u + (uint32_t)((g ^ ((pxr >> 1) & 7)) << 4);
    }
    // the residual ring (RES): three rows of 128 pixels x 128 bytes, slot c of pixel p at c ^ ((p >> 1) & 7) like the input ring; this lane's
    // 16 bytes of item (pt, j): pixel 32 wave + 16 pt + n, slot 4 j + g
    const uint32_t roff = (uint32_t)(wave * 32 + n) * 128u + (uint32_t)((g ^ (((wave * 32 + n) >> 1) & 7)) << 4);
    const uint32_t ring_base = lds_addr(ring);
    const uint32_t rres_base = lds_addr(rres);
    const int dpx = lane >> 3, dslot = lane & 7;          // LDS-DMA: lane -> (pixel within an 8-pixel piece, physical 16-byte slot)
    const int rows_per_unit = p.rows, units_per_col = p.units_per_col, segs = p.segs, nunits = p.nunits;

    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const int col = unit / units_per_col, yu = unit - col * units_per_col;
        const int b = col / segs, x0 = (col - b * segs) * 128;
        const int y0 = yu * rows_per_unit, y1 = min(y0 + rows_per_unit, H);
        const uint16_t* const img = xg + (int64_t)b * H * W * 64;
        // this lane's 16 bytes of an output row: pixel 32 wave + 16 pt + n, channels 32 j + 8 g
        const int lane_off32 = (wave * 32 + n) * 64 + 8 * g;
        // LDS-DMA of input row iy: piece slot k = 0 .. 4 of this wave is piece wave + 4 k (17 pieces of 8 pixels; waves 1 .. 3 have no fifth
        // piece: theirs goes to the dump line, so that every wave issues the SAME number of vector-memory operations per row -- the
        // counted vmcnt at the end of a row depends on it)
#define R16_IN_PIECE(iy_, k_)                                                                                                           \
        {                                                                                                                              \
            int dpx_ = dpx;                                                                                                            \
            asm volatile("" : "+v"(dpx_));       /* (opaque: the piece's lane offsets are re-formed here, not kept live across the row loop) */ \
            const int pc = wave + 4 * (k_), pxr = pc * 8 + dpx_, ix = x0 - 1 + pxr;                                                    \
            const bool ok = pxr < 130 && (unsigned)ix < (unsigned)W && (unsigned)(iy_) < (unsigned)H;                                  \
            float* const dst_ = pc < 17 ? reinterpret_cast<float*>(ring + (((iy_) - y0 + 1) % RS_NR) * RS_ROW_BYTES) + pc * 256 : dump; \
            lds_dma16(ok ? img + ((int64_t)(iy_) * W + ix) * 64 + ((dslot ^ ((pxr >> 1) & 7)) << 3) : zero_ + (dslot << 3), dst_);      \
        }
        // LDS-DMA of residual row ry (pieces wave, wave + 4, wave + 8, wave + 12 of 16); rows outside the unit: the zero line
#define R16_RES_PIECE(ry_, k_)                                                                                                          \
        {                                                                                                                              \
            int dpx_ = dpx;                                                                                                            \
            asm volatile("" : "+v"(dpx_));                                                                                             \
            const int pc = wave + 4 * (k_), pl = pc * 8 + dpx_;                                                                        \
            const bool ok = (ry_) < y1;                                                                                                \
            float* const dst_ = reinterpret_cast<float*>(rres + (((ry_) - y0) % 3) * R16_RES_ROW_BYTES) + pc * 256;                     \
            lds_dma16(ok ? resg + (((int64_t)b * H + (ry_)) * W + x0 + pl) * 64 + ((dslot ^ ((pl >> 1) & 7)) << 3) : zero_ + (dslot << 3), dst_); \
        }
        if constexpr (has_stat) {
#pragma unroll
            for (int k = 0; k < 32; ++k) aux[k] = 0.f;
        }
        f32x4 acc[2][2][4];                                                   // [set][pixel tile][channel tile]: AGPRs
        f32x4 rr = {0.f, 0.f, 0.f, 0.f};                                      // the residual's 16 bytes of the item in flight
        float f[8];
        uint4 o = make_uint4(0, 0, 0, 0);
        // epilogue item (pt, j) of output row yy from accumulator set `S_`, in two halves (A: values, B: rounding + statistics + store);
        // R16_RR issues the LDS read of the item's residual one step before its half A
#define R16_RR(yy_, pt_, j_)                                                                                                           \
        if constexpr (has_res) rr = lds_read128_async<2048 * (pt_)>(rres_base + (uint32_t)(((yy_) - y0) % 3) * R16_RES_ROW_BYTES + (roff ^ ((j_) ? 64u : 0u)));
        // the item's work in EIGHT slices (A0 .. A3: values, B0 .. B3: rounding, statistics, store), one per MFMA pair of two steps: a slice is
        // <= 12 vector instructions, what fits under the two MFMAs in front of it -- a whole half item (30+) between two steps left the
        // matrix pipe idle behind its one queued MFMA
#define R16_A0(S_, pt_, j_) { const f32x4 a0_ = acc[S_][pt_][2 * (j_)]; f[0] = a0_[0]; f[1] = a0_[1]; f[2] = a0_[2]; f[3] = a0_[3];                    \
            if constexpr (has_affine) { _Pragma("unroll") for (int i = 0; i < 4; ++i) f[i] = f[i] * aux[8 * (j_) + i] + aux[16 + 8 * (j_) + i]; } }
#define R16_A1(S_, pt_, j_) { const f32x4 a1_ = acc[S_][pt_][2 * (j_) + 1]; f[4] = a1_[0]; f[5] = a1_[1]; f[6] = a1_[2]; f[7] = a1_[3];                \
            if constexpr (has_affine) { _Pragma("unroll") for (int i = 4; i < 8; ++i) f[i] = f[i] * aux[8 * (j_) + i] + aux[16 + 8 * (j_) + i]; } }
#define R16_A2(S_, pt_, j_) { if constexpr (has_res) {                                                                                                  \
            /* rr was read in the last slice of the step before; the only LDS operations issued since are this step's two fragment reads, and   \
               LDS operations return in order */                                                                                                       \
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(rr) :: "memory");                                                                               \
            const uint32_t rx = __float_as_uint(rr[0]), ry_ = __float_as_uint(rr[1]);                                                                   \
            f[0] += __uint_as_float(rx << 16); f[1] += __uint_as_float(rx & 0xffff0000u); f[2] += __uint_as_float(ry_ << 16); f[3] += __uint_as_float(ry_ & 0xffff0000u); } }
#define R16_A3(S_, pt_, j_) { if constexpr (has_res) {                                                                                                  \
            const uint32_t rz = __float_as_uint(rr[2]), rw = __float_as_uint(rr[3]);                                                                    \
            f[4] += __uint_as_float(rz << 16); f[5] += __uint_as_float(rz & 0xffff0000u); f[6] += __uint_as_float(rw << 16); f[7] += __uint_as_float(rw & 0xffff0000u); } }
#define R16_STAT2(j_, k_, word_) { const float a0 = __uint_as_float((word_) << 16), a1 = __uint_as_float((word_) & 0xffff0000u);                       \
            aux[8 * (j_) + 2 * (k_)] += a0; aux[16 + 8 * (j_) + 2 * (k_)] += a0 * a0; aux[8 * (j_) + 2 * (k_) + 1] += a1; aux[16 + 8 * (j_) + 2 * (k_) + 1] += a1 * a1; }
#define R16_B0(yy_, pt_, j_) { o.x = rs_pack2(f[0], f[1]); o.y = rs_pack2(f[2], f[3]); if (relu) { o.x = rs_relu2(o.x); o.y = rs_relu2(o.y); } }
#define R16_B1(yy_, pt_, j_) { o.z = rs_pack2(f[4], f[5]); o.w = rs_pack2(f[6], f[7]); if (relu) { o.z = rs_relu2(o.z); o.w = rs_relu2(o.w); } }
#define R16_B2(yy_, pt_, j_) { if constexpr (has_stat) { R16_STAT2(j_, 0, o.x) R16_STAT2(j_, 1, o.y) } }
#define R16_B3(yy_, pt_, j_) { if constexpr (has_stat) { R16_STAT2(j_, 2, o.z) R16_STAT2(j_, 3, o.w) }                                                   \
            int lo_ = lane_off32;                                                                                                                       \
            asm volatile("" : "+v"(lo_));                                                                                                               \
            *reinterpret_cast<uint4*>(yg + (((int64_t)b * H + (yy_)) * W + x0) * 64 + lo_ + (pt_) * 1024 + (j_) * 32) = o; }
#define R16_ITEM_A(S_, pt_, j_) { R16_A0(S_, pt_, j_) R16_A1(S_, pt_, j_) R16_A2(S_, pt_, j_) R16_A3(S_, pt_, j_) }
#define R16_ITEM_B(yy_, pt_, j_) { R16_B0(yy_, pt_, j_) R16_B1(yy_, pt_, j_) R16_B2(yy_, pt_, j_) R16_B3(yy_, pt_, j_) }
        // prologue of the unit: input rows y0 - 1 .. y0 + 2 (rows beyond the unit's last + 1 are not needed yet: the zero line) and the first
        // residual row
        for (int iy = y0 - 1; iy <= y0 + 2; ++iy) {
            const bool need = iy <= y1;
            if (need) { R16_IN_PIECE(iy, 0) R16_IN_PIECE(iy, 1) R16_IN_PIECE(iy, 2) R16_IN_PIECE(iy, 3) R16_IN_PIECE(iy, 4) }
        }
        if constexpr (has_res) { R16_RES_PIECE(y0, 0) R16_RES_PIECE(y0, 1) R16_RES_PIECE(y0, 2) R16_RES_PIECE(y0, 3) }
        wait_vmcnt<0>();
        __syncthreads();

        // one output row: 18 steps (tap t = q / 2, k half ks = q % 2) of 2 pixel-fragment reads + 8 MFMAs.  Between the steps: the epilogue
        // of the previous row out of the other accumulator set (steps 1 .. 9), then the LDS-DMA of input row y + 3 and of residual row y + 1
        // (steps 10 .. 13) -- the row's YOUNGEST vector-memory operations: the counted wait at the end of the row lets exactly those stay
        // in flight (they are needed a row later) and covers the previous row's pieces; loads return in order, so "all but the NP youngest"
        // cannot be satisfied while an older piece is out, whatever the stores do.
#define R16_ADDR(q_) (sb[((q_) / 2) / 3] + (poff[((q_) / 2) % 3] ^ (((q_) % 2) ? 64u : 0u)))
#define R16_MF(S_, q_, ct_)                                                                                                            \
            r16_mfma<((q_) * 4 + (ct_) < R16_WA), (q_) == 0>(acc[S_][0][ct_], Wf[(q_) * 4 + (ct_)], Bp[((q_) % 3) * 2 + 0]);       \
            r16_mfma<((q_) * 4 + (ct_) < R16_WA), (q_) == 0>(acc[S_][1][ct_], Wf[(q_) * 4 + (ct_)], Bp[((q_) % 3) * 2 + 1]);
#define R16_NOP_
        // a step: the two fragment reads of step q + 2, the wait for step q's, then four MFMA pairs (channel tiles 0 .. 3 x both pixel tiles)
        // with one slice of side work (X0_ .. X3_: an epilogue slice, an LDS-DMA piece, a residual read -- or nothing) behind each pair
#define R16_STEP(S_, q_, X0_, X1_, X2_, X3_)                                                                                           \
        {                                                                                                                              \
            /* window of three steps: the fragments of step q + 2 go into the registers step q - 1 read */                              \
            if ((q_) + 2 < 18) {                                                                                                       \
                const uint32_t ad_ = R16_ADDR((q_) + 2 < 18 ? (q_) + 2 : 0);                                                           \
                Bp[(((q_) + 2) % 3) * 2 + 0] = lds_read128_async<0>(ad_);                                                              \
                Bp[(((q_) + 2) % 3) * 2 + 1] = lds_read128_async<2048>(ad_);                                                           \
            }                                                                                                                          \
            if ((q_) + 2 < 18) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Bp[((q_) % 3) * 2]), "+v"(Bp[((q_) % 3) * 2 + 1]) :: "memory"); \
            else if ((q_) == 16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Bp[((q_) % 3) * 2]), "+v"(Bp[((q_) % 3) * 2 + 1]) :: "memory"); \
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bp[((q_) % 3) * 2]), "+v"(Bp[((q_) % 3) * 2 + 1]) :: "memory");            \
            R16_MF(S_, q_, 0) X0_ __builtin_amdgcn_sched_barrier(0);                                                                   \
            R16_MF(S_, q_, 1) X1_ __builtin_amdgcn_sched_barrier(0);                                                                   \
            R16_MF(S_, q_, 2) X2_ __builtin_amdgcn_sched_barrier(0);                                                                   \
            R16_MF(S_, q_, 3) X3_ __builtin_amdgcn_sched_barrier(0);                                                                   \
        }
        // the same step with its four slices of side work BEHIND the eight MFMAs and one scheduling barrier (the kinds with 32 auxiliary registers)
#define R16_STEPC(S_, q_, X0_, X1_, X2_, X3_)                                                                                          \
        {                                                                                                                              \
            if ((q_) + 2 < 18) {                                                                                                       \
                const uint32_t ad_ = R16_ADDR((q_) + 2 < 18 ? (q_) + 2 : 0);                                                           \
                Bp[(((q_) + 2) % 3) * 2 + 0] = lds_read128_async<0>(ad_);                                                              \
                Bp[(((q_) + 2) % 3) * 2 + 1] = lds_read128_async<2048>(ad_);                                                           \
            }                                                                                                                          \
            if ((q_) + 2 < 18) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Bp[((q_) % 3) * 2]), "+v"(Bp[((q_) % 3) * 2 + 1]) :: "memory"); \
            else if ((q_) == 16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Bp[((q_) % 3) * 2]), "+v"(Bp[((q_) % 3) * 2 + 1]) :: "memory"); \
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bp[((q_) % 3) * 2]), "+v"(Bp[((q_) % 3) * 2 + 1]) :: "memory");            \
            R16_MF(S_, q_, 0) R16_MF(S_, q_, 1) R16_MF(S_, q_, 2) R16_MF(S_, q_, 3)                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
            X0_ X1_ X2_ X3_                                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                                                         \
        }
        // side work of a row, guarded by `prev` (the first row of a unit has no previous row) -- wave-uniform branches
#define P_(X_) if (prev) { X_ }
#define R16_ROW_HEAD                                                                                                                   \
            const bool prev = y > y0;                                                                                                  \
            uint32_t sb[3];                                                                                                            \
            _Pragma("unroll") for (int r = 0; r < 3; ++r) sb[r] = ring_base + (uint32_t)((y - 1 + r - y0 + 1) % RS_NR) * RS_ROW_BYTES; \
            f32x4 Bp[6];                                                                                                               \
            { const uint32_t ad_ = R16_ADDR(0); Bp[0] = lds_read128_async<0>(ad_); Bp[1] = lds_read128_async<2048>(ad_); }             \
            { const uint32_t ad_ = R16_ADDR(1); Bp[2] = lds_read128_async<0>(ad_); Bp[3] = lds_read128_async<2048>(ad_); }
#define R16_ROW_TAIL(S_, STEP_)                                                                                                        \
            STEP_(S_, 10, R16_IN_PIECE(y + 3, 0), R16_NOP_, R16_IN_PIECE(y + 3, 1), R16_NOP_)                                          \
            STEP_(S_, 11, R16_IN_PIECE(y + 3, 2), R16_NOP_, R16_IN_PIECE(y + 3, 3), R16_NOP_)                                          \
            STEP_(S_, 12, R16_IN_PIECE(y + 3, 4), R16_NOP_, R16_RS_(y + 1, 0), R16_NOP_)                                               \
            STEP_(S_, 13, R16_RS_(y + 1, 1), R16_NOP_, R16_RS_(y + 1, 2), R16_NOP_)                                                    \
            STEP_(S_, 14, R16_RS_(y + 1, 3), R16_NOP_, R16_NOP_, R16_NOP_)                                                             \
            STEP_(S_, 15, R16_NOP_, R16_NOP_, R16_NOP_, R16_NOP_)                                                                      \
            STEP_(S_, 16, R16_NOP_, R16_NOP_, R16_NOP_, R16_NOP_)                                                                      \
            STEP_(S_, 17, R16_NOP_, R16_NOP_, R16_NOP_, R16_NOP_)                                                                      \
            /* (a compiler-made copy of an accumulator on the loop's exit edge must not read an MFMA result in flight: the asm MFMAs are invisible \
               to the hazard recogniser; 4 passes + write-back are over after these wait states and the barrier) */                             \
            asm volatile("s_nop 7" ::: "memory");                                                                                      \
            /* the row's nine (five without a residual) LDS-DMA pieces are its YOUNGEST vector-memory operations: they may stay in flight    \
               (needed a row later); everything older -- the previous row's pieces -- is waited for */                                       \
            if constexpr (has_res) wait_vmcnt<9>(); else wait_vmcnt<5>();                                                              \
            __builtin_amdgcn_s_barrier();      /* every wave is done with input row y - 1; the pieces issued a row ago are published */
        // the plain kind: an item's eight slices behind the eight MFMA pairs of two steps (84 instead of 88 us per launch)
#define R16_ROW_SLICED(S_)                                                                                                             \
        {                                                                                                                              \
            R16_ROW_HEAD                                                                                                               \
            R16_STEP(S_, 0, R16_NOP_, R16_NOP_, R16_NOP_, R16_NOP_)                                                                    \
            R16_STEP(S_, 1, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_RR(y - 1, 0, 0)))                                                     \
            R16_STEP(S_, 2, P_(R16_A0(1 - (S_), 0, 0)), P_(R16_A1(1 - (S_), 0, 0)), P_(R16_A2(1 - (S_), 0, 0)), P_(R16_A3(1 - (S_), 0, 0) R16_RR(y - 1, 0, 1))) \
            R16_STEP(S_, 3, P_(R16_B0(y - 1, 0, 0)), P_(R16_B1(y - 1, 0, 0)), P_(R16_B2(y - 1, 0, 0)), P_(R16_B3(y - 1, 0, 0)))         \
            R16_STEP(S_, 4, P_(R16_A0(1 - (S_), 0, 1)), P_(R16_A1(1 - (S_), 0, 1)), P_(R16_A2(1 - (S_), 0, 1)), P_(R16_A3(1 - (S_), 0, 1) R16_RR(y - 1, 1, 0))) \
            R16_STEP(S_, 5, P_(R16_B0(y - 1, 0, 1)), P_(R16_B1(y - 1, 0, 1)), P_(R16_B2(y - 1, 0, 1)), P_(R16_B3(y - 1, 0, 1)))         \
            R16_STEP(S_, 6, P_(R16_A0(1 - (S_), 1, 0)), P_(R16_A1(1 - (S_), 1, 0)), P_(R16_A2(1 - (S_), 1, 0)), P_(R16_A3(1 - (S_), 1, 0) R16_RR(y - 1, 1, 1))) \
            R16_STEP(S_, 7, P_(R16_B0(y - 1, 1, 0)), P_(R16_B1(y - 1, 1, 0)), P_(R16_B2(y - 1, 1, 0)), P_(R16_B3(y - 1, 1, 0)))         \
            R16_STEP(S_, 8, P_(R16_A0(1 - (S_), 1, 1)), P_(R16_A1(1 - (S_), 1, 1)), P_(R16_A2(1 - (S_), 1, 1)), P_(R16_A3(1 - (S_), 1, 1))) \
            R16_STEP(S_, 9, P_(R16_B0(y - 1, 1, 1)), P_(R16_B1(y - 1, 1, 1)), P_(R16_B2(y - 1, 1, 1)), P_(R16_B3(y - 1, 1, 1)))         \
            R16_ROW_TAIL(S_, R16_STEP)                                                                                                           \
        }
        // the kinds with 32 auxiliary registers (affine, statistics): whole half items behind a step -- with a scheduling barrier after every MFMA
        // pair the register allocator spills (statistics: 91 registers, 305 us per launch; affine + residual 139 us)
#define R16_ROW_COARSE(S_)                                                                                                             \
        {                                                                                                                              \
            R16_ROW_HEAD                                                                                                               \
            R16_STEPC(S_, 0, R16_NOP_, R16_NOP_, R16_NOP_, R16_NOP_)                                                                    \
            R16_STEPC(S_, 1, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_RR(y - 1, 0, 0)))                                                     \
            R16_STEPC(S_, 2, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_A(1 - (S_), 0, 0) R16_RR(y - 1, 0, 1)))                           \
            R16_STEPC(S_, 3, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_B(y - 1, 0, 0)))                                                 \
            R16_STEPC(S_, 4, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_A(1 - (S_), 0, 1) R16_RR(y - 1, 1, 0)))                           \
            R16_STEPC(S_, 5, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_B(y - 1, 0, 1)))                                                 \
            R16_STEPC(S_, 6, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_A(1 - (S_), 1, 0) R16_RR(y - 1, 1, 1)))                           \
            R16_STEPC(S_, 7, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_B(y - 1, 1, 0)))                                                 \
            R16_STEPC(S_, 8, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_A(1 - (S_), 1, 1)))                                              \
            R16_STEPC(S_, 9, R16_NOP_, R16_NOP_, R16_NOP_, P_(R16_ITEM_B(y - 1, 1, 1)))                                                 \
            R16_ROW_TAIL(S_, R16_STEPC)                                                                                                           \
        }
#define R16_ROW(S_) { if constexpr (KIND == 0) R16_ROW_SLICED(S_) else R16_ROW_COARSE(S_) }
#define R16_RS_(ry_, k_) { if constexpr (has_res) { R16_RES_PIECE(ry_, k_) } }
        int y = y0;
        for (; y + 1 < y1; y += 2) {
            R16_ROW(0)
            ++y;
            R16_ROW(1)
            --y;
        }
        int last_set = 1;
        if (y < y1) { R16_ROW(0) last_set = 0; ++y; }
        // the last row's epilogue: nothing to hide it under.  (The asm MFMAs are invisible to the hazard recogniser: their results must
        // not be read for 18 wait states.)  Its residual row was issued a row ago (or in the prologue) and is covered by the last counted
        // wait + barrier.
#define R16_SETTLE(S_) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[S_][0][0]), "+a"(acc[S_][0][1]), "+a"(acc[S_][0][2]), "+a"(acc[S_][0][3]),       \
                                                           "+a"(acc[S_][1][0]), "+a"(acc[S_][1][1]), "+a"(acc[S_][1][2]), "+a"(acc[S_][1][3]) :: "memory");
        // (the accumulators are operands of the nops: without the dependency the compiler is free to read them above the nops -- it did, in the
        // statistics instantiation: the last item of a unit's last row came out of an unfinished MFMA)
        if (last_set == 0) { R16_SETTLE(0) } else { R16_SETTLE(1) }
#undef R16_SETTLE
#define R16_LAST(S_, pt_, j_)                                                                                                          \
        { R16_RR(y1 - 1, pt_, j_)                                                                                                      \
          if constexpr (has_res) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr) :: "memory");                                          \
          R16_ITEM_A(S_, pt_, j_) R16_ITEM_B(y1 - 1, pt_, j_) }
        if (last_set == 0) { R16_LAST(0, 0, 0) R16_LAST(0, 0, 1) R16_LAST(0, 1, 0) R16_LAST(0, 1, 1) }
        else { R16_LAST(1, 0, 0) R16_LAST(1, 0, 1) R16_LAST(1, 1, 0) R16_LAST(1, 1, 1) }
#undef R16_LAST
#undef R16_ROW
#undef R16_ROW_SLICED
#undef R16_ROW_COARSE
#undef R16_ROW_HEAD
#undef R16_ROW_TAIL
#undef R16_RS_
#undef P_
#undef R16_NOP_
#undef R16_STEP
#undef R16_STEPC
#undef R16_MF
#undef R16_ADDR
#undef R16_ITEM_A
#undef R16_ITEM_B
#undef R16_A0
#undef R16_A1
#undef R16_A2
#undef R16_A3
#undef R16_B0
#undef R16_B1
#undef R16_B2
#undef R16_B3
#undef R16_STAT2
#undef R16_RR
#undef R16_RES_PIECE
#undef R16_IN_PIECE
        if constexpr (has_stat) {
            // the 16 lanes n of a lane group hold the same channels of different pixels
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                float a = aux[k];
                for (int o_ = 1; o_ < 16; o_ <<= 1) a += __shfl_xor(a, o_);
                if (n == 0) sred[wave * 128 + (k >> 4) * 64 + 32 * ((k & 15) >> 3) + 8 * g + (k & 7)] = a;
            }
            __syncthreads();
            if (tid < 128) p.stat[(int64_t)unit * 128 + tid] = (sred[tid] + sred[128 + tid]) + (sred[256 + tid] + sred[384 + tid]);
        }
        wait_vmcnt<0>();
        __syncthreads();                     // the rings and sred are reused by the next unit
    }
}


int launch_rows16_bf16(const RowsArgs& ra, hipStream_t st) {
    const dim3 grid(std::min(ra.nunits, 256));
#define SD_R16(KIND_, RES_) { static thread_local bool up = false;                                                                          \
    if (!up) { SD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c64_rows16_bf16<KIND_, RES_>), hipFuncAttributeMaxDynamicSharedMemorySize, R16_LDS_BYTES)); up = true; } \
    hipLaunchKernelGGL((k_conv3x3_c64_rows16_bf16<KIND_, RES_>), grid, dim3(256), R16_LDS_BYTES, st, ra); }
    const bool affine = ra.scale || ra.shift;
    SD_REQUIRE(!ra.stat, SD_ERR_INVALID, "launch_rows16_bf16: the statistics launch belongs to k_conv3x3_c64_rows_bf16");
    if (affine && ra.res) SD_R16(1, true)
    else if (affine) SD_R16(1, false)
    else if (ra.res) SD_R16(0, true)
    else SD_R16(0, false)
#undef SD_R16
    SD_LAUNCH_CHECK();
    return 0;
}

}  // namespace sd
