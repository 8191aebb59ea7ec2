// Shared by sd_conv.hip (dispatch, the 32x32x16 row-stream kernel) and sd_conv_rows16.hip (the swapped-operand row-stream kernel):
// the arguments and ring geometry of the bf16 row-stream kernels of the 64 -> 64 channel layers, and the packed bf16 helpers.
#pragma once
#include "sd_common.h"
#include "sd_mfma.h"

namespace sd {

constexpr int RS_PX = 136, RS_ROW_BYTES = RS_PX * 128, RS_NR = 5;     // ring row: 17 LDS-DMA pieces of 8 pixels x 128 bytes; five rows

struct RowsArgs {
    const uint16_t* x;     // [B][H][W][64] bf16
    const uint16_t* w;     // [64 n][9][64 c] bf16 (data-gradient: the transposed weights, flip = 1)
    uint16_t* y;           // [B][H][W][64] bf16
    const float* scale;    // nullable
    const float* shift;    // nullable
    const uint16_t* res;   // nullable, same shape as y
    float* stat;           // nullable: [nunits][2][64] column sums / sums of squares of the rounded output
    int B, H, W, relu, flip, rows, units_per_col, segs, nunits;
};

// packed bf16 helpers
typedef __bf16 rs_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rs_f32x2 __attribute__((ext_vector_type(2)));
typedef short rs_i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t rs_pack2(float lo, float hi) {            // two fp32 -> one dword of two bf16 (round-to-nearest-even)
    const rs_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rs_bf16x2));
}
__device__ __forceinline__ uint32_t rs_relu2(uint32_t two_bf16) {             // negative bf16 are negative int16: max(x, 0) per half
    const rs_i16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(rs_i16x2, two_bf16), z));
}

// sd_conv_rows16.hip: k_conv3x3_c64_rows16_bf16 for a launch conv_rows16_args() accepted (no statistics); 0 or an SD_ERR_* / hipError_t code
int launch_rows16_bf16(const RowsArgs& ra, hipStream_t st);

}  // namespace sd
