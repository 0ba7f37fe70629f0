// Device helpers of the segment-attention kernels (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
// Fragment / tile layout algebra: see ttt_mfma_dev.h (A: lane=i, regs=k; B: lane=j, regs=k; C/D: lane=col,
// regs=rows row_of(r,h); a C tile is re-usable in place as an operand contracting over its row index, with
// k-slot order pi_s(h,e) = 16s + 8(e>>2) + 4h + (e&3)).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ttt_mfma_dev.h"

namespace ttt {
namespace attn {
using namespace ttt::mf;

constexpr int AS = 72;                 // LDS row stride (elements) of a [rows][64] bf16 tile: 144 B, conflict-free b128 / tr_b64
constexpr int ATILE = 64 * AS;         // elements of one [64][64] tile

// transposed LDS read (ds_read_b64_tr_b16): operand fragment with outer index = column, contraction index = row,
// of a row-major bf16 image; rows r0..r0+3 and r1..r1+3 fill the 8 k-slots, 32 outer columns start at col0.
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* img, int stride, int r0, int r1, int col0, int l) {
    const int i = l & 15, g1 = (l >> 4) & 1;
    const int off = (i >> 2) * stride + col0 + 16 * g1 + 4 * (i & 3);
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + r0 * stride + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + r1 * stride + off));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// same, in the pi k-slot order of in-place fragment s of the 32-row block starting at row0
__device__ __forceinline__ bf16x8 tr_frag_pi(const __bf16* img, int stride, int row0, int s, int col0, int l) {
    const int h = l >> 5;
    return tr_frag(img, stride, row0 + 16 * s + 4 * h, row0 + 16 * s + 8 + 4 * h, col0, l);
}
// plain operand fragment: lane c reads 8 contiguous elements of its row (outer index = row, contraction = column)
__device__ __forceinline__ bf16x8 row_frag(const __bf16* img, int stride, int row0, int col0, int l) {
    return *reinterpret_cast<const bf16x8*>(img + (row0 + (l & 31)) * stride + col0 + 8 * (l >> 5));
}
// fragment whose k-slots follow the pi order along the COLUMNS of a row-major image (outer index = row)
__device__ __forceinline__ bf16x8 row_frag_pi(const __bf16* img, int stride, int row0, int col0, int s, int l) {
    return pi_read(img + (row0 + (l & 31)) * stride, col0, s, l >> 5);
}

__device__ __forceinline__ float half_swap(float v) { return __shfl_xor(v, 32, 64); }

}  // namespace attn
}  // namespace ttt
