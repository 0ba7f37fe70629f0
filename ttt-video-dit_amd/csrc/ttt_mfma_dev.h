// Device-side building blocks of the MFMA TTT kernels (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
// Layout algebra used everywhere (lane l, h = l>>5, c = l&31):
//   MFMA D[32x32] += A[32x16] * B[16x32]
//     A fragment: lane holds A[i = c][k = 8h + e], e = 0..7        (8 bf16)
//     B fragment: lane holds B[k = 8h + e][j = c]
//     C/D tile  : lane holds D[row(r,h)][col = c], r = 0..15, row(r,h) = (r&3) + 8*(r>>2) + 4h
//   A C/D tile X (rows = R-index, lane = C-index) can be re-used IN PLACE as an operand that
//   contracts over its ROW index: registers 8s..8s+7 (s = 0,1) form one K=16 fragment whose
//   k-slot (h,e) carries row  pi_s(h,e) = 16s + 8*(e>>2) + 4h + (e&3)  of the tile:
//     as B operand:  B[k][j=c]  = X[pi_s(k)][c]          (k-dim = rows, N-dim = lane index)
//     as A operand:  A[i=c][k]  = X[pi_s(k)][c] = X^T    (M-dim = lane index)
//   The partner operand only has to present the SAME k-slot order: another tile's registers
//   (automatically), or an LDS row read as two 8-byte chunks at columns 16s+4h and 16s+8+4h
//   ("pi read").  Contraction over the LANE index is impossible in place; a 32x32 tile is
//   transposed with two MFMAs against constant identity fragments:  X^T = (X as A) * I_pi.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "ttt_common.h"

namespace ttt {
namespace mf {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int TS = 72;            // LDS row stride of a [64][64] bf16 tile, in elements (144 B: 16-B aligned, conflict-free b128)
constexpr int PS = 68;            // LDS row stride of a [64][64] fp32 partial tile, in floats (272 B)

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mma(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

// registers 8s..8s+7 of a C tile -> one K=16 operand fragment (round-to-nearest-even bf16)
__device__ __forceinline__ bf16x8 pack(const f32x16& t, int s) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)t[8 * s + e];
    return r;
}
// same, with a per-row scale: sc[r] multiplies register r (rows live in registers)
__device__ __forceinline__ bf16x8 pack_scaled(const f32x16& t, const f32x16& sc, int s) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)(t[8 * s + e] * sc[8 * s + e]);
    return r;
}

// "pi read": operand fragment for k-slice (s) of a 32-wide column block starting at col0, from a
// row-major bf16 LDS tile; `rowp` points at this lane's row.  Two 8-byte reads.
__device__ __forceinline__ bf16x8 pi_read(const __bf16* rowp, int col0, int s, int h) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(rowp + col0 + 16 * s + 4 * h);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(rowp + col0 + 16 * s + 8 + 4 * h);
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 4; ++e) { r[e] = lo[e]; r[4 + e] = hi[e]; }
    return r;
}

// identity fragment (as B operand) for the pi slot order: I[slot(h,e)][j=c] = (pi_s(h,e) == c)
__device__ __forceinline__ bf16x8 ident_pi(int s, int h, int c) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)((16 * s + 8 * (e >> 2) + 4 * h + (e & 3)) == c ? 1.0f : 0.0f);
    return r;
}

// transpose a 32x32 tile given as two packed fragments (a0,a1 = pack(X,0), pack(X,1)): returns
// X^T as a C tile (rows = X's lane index, lane = X's row index)
__device__ __forceinline__ f32x16 transpose_tile(bf16x8 a0, bf16x8 a1, bf16x8 i0, bf16x8 i1) {
    f32x16 d = zero16();
    d = mma(a0, i0, d);
    d = mma(a1, i1, d);
    return d;
}

__device__ __forceinline__ float xor_add(float v, int mask) { return v + __shfl_xor(v, mask, 64); }

// fast tanh-GELU pieces (same constants as the reference, ops/utils.py:51-54)
__device__ __forceinline__ float sigmoid2u(float x, float x2) {   // sigmoid(2u), u = a*x*(1+c*x^2)
    const float u2 = 2.0f * GELU_A * x * (1.0f + GELU_C * x2);
    return __builtin_amdgcn_rcpf(1.0f + __expf(-u2));
}
__device__ __forceinline__ void gelu_fwd_grad(float x, float& y, float& dy) {
    const float x2 = x * x;
    const float s = sigmoid2u(x, x2);            // (1+tanh u)/2
    y = x * s;
    // gelu' = s + x * (1 - t^2)/2 * u' ,  (1 - t^2) = 4 s (1 - s),  u' = a + 3ac x^2
    dy = s + 2.0f * x * s * (1.0f - s) * (GELU_A + GELU_3AC * x2);
}
__device__ __forceinline__ float gelu_fwd(float x) { return x * sigmoid2u(x, x * x); }

}  // namespace mf
}  // namespace ttt
