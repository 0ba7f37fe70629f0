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

// fast tanh-GELU pieces (same constants as the reference, ops/utils.py:51-54).  With u = a x (1 + c x^2):
//   gelu(x) = x * sig(2u),  sig(2u) = 1 / (1 + 2^(x * (k0 + k1 x^2))),  k0 = -2 a log2(e), k1 = k0 * c
//   gelu'(x) = s + x s (1 - s) * 2 u' = s + (y - y s) (2a + 6ac x^2)
constexpr float GELU_K0 = -2.0f * GELU_A * 1.4426950408889634f;
constexpr float GELU_K1 = GELU_K0 * GELU_C;
__device__ __forceinline__ float sig2u(float x, float x2) {
    const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x2, GELU_K1, GELU_K0));
    return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float sigmoid2u(float x, float x2) { return sig2u(x, x2); }
__device__ __forceinline__ void gelu_fwd_grad(float x, float& y, float& dy) {
    const float x2 = x * x;
    const float s = sig2u(x, x2);                // (1 + tanh u) / 2
    y = x * s;
    dy = __builtin_fmaf(__builtin_fmaf(-y, s, y), __builtin_fmaf(x2, 2.0f * GELU_3AC, 2.0f * GELU_A), s);
}
__device__ __forceinline__ float gelu_fwd(float x) { return x * sig2u(x, x * x); }

__device__ __forceinline__ float gelu_grad_only(float x) {
    float y, dy;
    gelu_fwd_grad(x, y, dy);
    return dy;
}
// y = gelu, dy = gelu', d2y = gelu''   (gelu'' = q u' + x q (u''/2 - t u'^2), q = 1 - t^2)
__device__ __forceinline__ void gelu_fwd_grad2(float x, float& y, float& dy, float& d2y) {
    const float x2 = x * x;
    const float s = sigmoid2u(x, x2);
    const float t = 2.0f * s - 1.0f;
    const float q = 4.0f * s * (1.0f - s);
    const float du = GELU_A + GELU_3AC * x2;
    const float d2u = 2.0f * GELU_3AC * x;
    y = x * s;
    dy = s + 0.5f * x * q * du;
    d2y = q * du + 0.5f * x * q * (d2u - 2.0f * t * du * du);
}

// store the two halves of a packed C-tile fragment (regs 8s..8s+7 of a tile with rows = R, lane = C) into the
// image img[C][R]: lane c writes rows (16s + 4h .. +3) and (16s + 8 + 4h .. +3) of its image row as 2 x 8 bytes
__device__ __forceinline__ void st_image(__bf16* img_row, int rbase, int s, int h, bf16x8 v) {
    bf16x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
    *reinterpret_cast<bf16x4*>(img_row + rbase + 16 * s + 4 * h) = lo;
    *reinterpret_cast<bf16x4*>(img_row + rbase + 16 * s + 8 + 4 * h) = hi;
}

// gelu_tanh over a C tile in place, on ALIGNED register pairs (r, r + 1) with the packed f32 instructions: the same IEEE
// operations in the same order as gelu_fwd(), two elements per v_pk_mul / v_pk_fma / v_pk_add.  Why (tools/isa_mix.py): left
// to itself the SLP vectorizer pairs the scalar form's multiplies across MISALIGNED registers of the tile and then spends
// 24 v_mov + 12 v_alignbit + 4 v_perm per step and wave (in this block and again around the fragment packs) to realign
// them - 230 VALU for 32 elements where ~120 do.  Round-2 A/B on an MI355X (NH = 48, NC = 282): 2.368 -> 2.195 ms per scan;
// the scalar form was removed.  (Not bit-identical to it - the contraction of mul + add into fma differs - but the same
// distance from the fp64 oracle.)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gelu_fwd_tile_pk(f32x16& z) {
    const f32x2 k0 = {GELU_K0, GELU_K0}, k1 = {GELU_K1, GELU_K1}, one = {1.0f, 1.0f};
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
        f32x2 x = {z[r], z[r + 1]};
        const f32x2 x2 = x * x;
        const f32x2 a = x * __builtin_elementwise_fma(x2, k1, k0);
        const f32x2 e = {__builtin_amdgcn_exp2f(a[0]), __builtin_amdgcn_exp2f(a[1])};
        const f32x2 d = one + e;
        const f32x2 sg = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
        x = x * sg;
        z[r] = x[0];
        z[r + 1] = x[1];
    }
}

// ------------------------------------------------------------------------------------------------
// LDS geometry shared by the scan kernels
constexpr int NT = 256;
constexpr int TILE_ELEMS = 64 * TS;                          // one padded [64][64] bf16 tile

// write one wave's partial [f][t] tiles (rows=f, lane=t) to red[w][t][f]
__device__ __forceinline__ void write_partial(float* redw, const f32x16 (&P)[2][2], int h, int c) {
#pragma unroll
    for (int fj = 0; fj < 2; ++fj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = {P[fj][ti][4 * q], P[fj][ti][4 * q + 1], P[fj][ti][4 * q + 2], P[fj][ti][4 * q + 3]};
                *reinterpret_cast<f32x4*>(redw + (32 * ti + c) * PS + 32 * fj + 8 * q + 4 * h) = v;
            }
}

// owner lane (token t, 16 features f0..f0+15): z = sum of the four partials (+ bias[f] if given)
__device__ __forceinline__ void gather_partial(const float* red, const float* bias, int t, int f0, float (&z)[16]) {
#pragma unroll
    for (int j = 0; j < 16; ++j) z[j] = bias ? bias[f0 + j] : 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + ((size_t)w * 64 + t) * PS + f0 + 4 * q);
            z[4 * q] += v[0]; z[4 * q + 1] += v[1]; z[4 * q + 2] += v[2]; z[4 * q + 3] += v[3];
        }
}

// sum over the 4 lanes (l, l^16, l^32, l^48) that share an owner token
__device__ __forceinline__ float quad_add(float v) { return xor_add(xor_add(v, 16), 32); }

// LayerNorm statistics of a 64-wide row spread over 4 lanes, 16 values each
__device__ __forceinline__ void row_stats(const float (&z)[16], float eps, float& mu, float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += z[j];
    mu = quad_add(s) * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) { const float d = z[j] - mu; v += d * d; }
    rstd = 1.0f / sqrtf(quad_add(v) * (1.0f / 64.0f) + eps);
}

__device__ __forceinline__ void load16_bf16(const __bf16* p, float (&o)[16]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
    const bf16x8 b = *reinterpret_cast<const bf16x8*>(p + 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) { o[j] = (float)a[j]; o[8 + j] = (float)b[j]; }
}
__device__ __forceinline__ void store16_bf16(__bf16* p, const float (&v)[16]) {
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)v[j]; b[j] = (__bf16)v[8 + j]; }
    *reinterpret_cast<bf16x8*>(p) = a;
    *reinterpret_cast<bf16x8*>(p + 8) = b;
}
__device__ __forceinline__ void load16_f32(const float* p, float (&o)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p + 4 * q);
        o[4 * q] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
    }
}

// per-register row values: o[r] = src[base + row_of(r,h)]  (src fp32 in LDS, 16-B aligned groups)
__device__ __forceinline__ f32x16 rows_from_lds(const float* src, int base, int h) {
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + base + 8 * q + 4 * h);
        o[4 * q] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
    }
    return o;
}
__device__ __forceinline__ f32x16 unpack2(bf16x8 lo, bf16x8 hi) {   // inverse of pack(.,0), pack(.,1)
    f32x16 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[e] = (float)lo[e]; o[8 + e] = (float)hi[e]; }
    return o;
}

// ------------------------------------------------------------------------------------------------
// Pieces of the backward's per-step record shared by its kernels (the record itself: ttt_bwd4_dev.h).  A fragment image is what
// one wave holds of a packed 32 x 16 operand: 64 lanes x 16 bytes, lane-linear; fragment arrays of a hidden slice hold 8 of
// them, index fr_idx(a, b, s).  (Round 2's 570-KiB record of 16 such arrays per wave - per-step W1 / W2 / W2^T images and
// second-orientation copies - was replaced in round 3: ttt_bwd4_dev.h.)
constexpr size_t FRAG_BYTES = 64 * 16;
// owner data: three row-major fp32 [64 t][64 f] arrays (0 = x_hat of the inner LN, 1 = go = y - target, 2 = x_hat of
// the output LN) + per-token (rstd, rstd_out)
constexpr size_t SLOT_OWN_ARR = 64 * 64 * 4;
constexpr size_t SLOT_OWN = 3 * SLOT_OWN_ARR + 64 * 8;
constexpr size_t SLOT_G = 64 * 64 * 2;                                   // gZ2 tile, bf16 row-major

__device__ __forceinline__ int fr_idx(int a, int b, int s) { return (a * 2 + b) * 2 + s; }
// owner rows: `own` = the record's owner area; token ot, features of0 .. of0 + N - 1
template <int N>
__device__ __forceinline__ void st_own(char* own, int arr, int ot, int of0, const float (&v)[N]) {
    float* p = reinterpret_cast<float*>(own + (size_t)arr * SLOT_OWN_ARR) + ot * 64 + of0;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        f32x4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        *reinterpret_cast<f32x4*>(p + 4 * q) = x;
    }
}
template <int N>
__device__ __forceinline__ void ld_own(const char* own, int arr, int ot, int of0, float (&v)[N]) {
    const float* p = reinterpret_cast<const float*>(own + (size_t)arr * SLOT_OWN_ARR) + ot * 64 + of0;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p + 4 * q);
        v[4 * q] = x[0]; v[4 * q + 1] = x[1]; v[4 * q + 2] = x[2]; v[4 * q + 3] = x[3];
    }
}
__device__ __forceinline__ float* own_stats(char* own, int ot) {
    return reinterpret_cast<float*>(own + 3 * SLOT_OWN_ARR) + 2 * ot;
}

}  // namespace mf
}  // namespace ttt
