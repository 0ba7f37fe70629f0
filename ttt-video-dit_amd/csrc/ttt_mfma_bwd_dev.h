// Device helpers shared by the TTT-MLP backward kernels (recompute ttt_mfma_rc4.hip, cluster sweep and tail ttt_mfma_bwd4.hip).
#pragma once
#include "ttt_mfma_dev.h"

namespace ttt {
namespace mfma {
using namespace ttt::mf;
namespace b2 {


constexpr int NT2 = 512;
typedef __attribute__((address_space(3))) bf16x4 lds_b4;

// ---- helpers shared with the revision-2 forward (same idioms) --------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8(float v) {
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    return v;
}
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* img, int stride, int r0, int r1, int col0, int l) {
    const int i = l & 15, g1 = (l >> 4) & 1;
    const int off = (i >> 2) * stride + col0 + 16 * g1 + 4 * (i & 3);
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + r0 * stride + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + r1 * stride + off));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// operand fragment (outer = column in [col0, col0+32), contraction = rows of the 32-row block at row0, pi slot order s)
__device__ __forceinline__ bf16x8 tr_pi(const __bf16* img, int row0, int s, int col0, int l) {
    const int h = l >> 5;
    return tr_frag(img, TS, row0 + 16 * s + 4 * h, row0 + 16 * s + 8 + 4 * h, col0, l);
}
// operand fragment (outer = row 32 ti + c, contraction = columns col0 + pi slots of s)
__device__ __forceinline__ bf16x8 row_pi(const __bf16* img, int ti, int col0, int s, int l) {
    return pi_read(img + (32 * ti + (l & 31)) * TS, col0, s, l >> 5);
}
__device__ __forceinline__ void load8_bf16(const __bf16* p, float (&o)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)a[j];
}
__device__ __forceinline__ void store8_bf16(__bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = a;
}
__device__ __forceinline__ void load8_f32(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void add8_f32(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] += a[0]; o[1] += a[1]; o[2] += a[2]; o[3] += a[3]; o[4] += b[0]; o[5] += b[1]; o[6] += b[2]; o[7] += b[3];
}
// one wave's partial tile (rows = f in Fp, lane = t of tile ti) -> red[w][t][f]
__device__ __forceinline__ void write_partial2(float* redw, const f32x16& P, int ti, int p, int h, int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = {P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]};
        *reinterpret_cast<f32x4*>(redw + (32 * ti + c) * PS + 32 * p + 8 * q + 4 * h) = v;
    }
}
__device__ __forceinline__ float tile_colsum(const f32x16& t) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += t[r];
    return xor_add(s, 32);
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
// Slot / tensor accesses of the sweep go through buffer instructions: wave-uniform base (SRD) + wave-uniform byte offset
// in an SGPR + ONE per-lane offset register (lane * 16 or thread * 16/32).  With flat addressing hipcc materialises a
// 64-bit address pair for each of the ~60 distinct slot accesses of a step at the top of the iteration and spills them.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(bytes > 0x7fffffffull ? 0x7fffffffull : bytes), 0x00020000);
}
__device__ __forceinline__ bf16x8 bld8(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bst8(__amdgpu_buffer_rsrc_t r, int voff, int soff, bf16x8 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 0);
}
__device__ __forceinline__ f32x4 bld4f(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ void bld8f(__amdgpu_buffer_rsrc_t r, int voff, int soff, float (&o)[8]) {
    const f32x4 a = bld4f(r, voff, soff), b = bld4f(r, voff + 16, soff);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
// write-through / L1-bypassing forms for the inter-workgroup exchange (aux 16 = sc1)
__device__ __forceinline__ void bst4f_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff, f32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 16);
}
__device__ __forceinline__ f32x4 bld4f_sc1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 16));
}

// carry area per (b,h), floats: natural-layout dW1 [64][256], dW2 [256][64], db1 [256], db2 [64], then per-thread dgamma / dbeta
constexpr size_t C_DW1 = 0, C_DW2 = 64 * 256, C_DB1 = 2 * 64 * 256, C_DB2 = C_DB1 + 256, C_DG = C_DB2 + 64, C_DBT = C_DG + NT2 * 8,
                 CARRY_FLOATS2 = C_DBT + NT2 * 8;

struct SweepParams2 {
    const __bf16 *XQ, *XK, *dOut, *eta;
    const float* ln_w;
    const float *uW1, *ub1, *uW2, *ub2;
    char* slots; size_t slot_stride_bh;
    float* carry;
    __bf16 *dXV, *deta;
    float *dW1, *db1, *dW2, *db2, *dlnw, *dlnb;
    int NH, NC, chunk_lo, chunk_hi, first, last;
    unsigned long long* dbg;                // optional: per-stage cycle totals of workgroup 0 (entries 16..27)
    int bh0, nbh;                           // this launch sweeps (b,h) = bh0 .. bh0 + nbh - 1: grid = 4 nbh workgroups
    int fast_records;                       // 1: plain (L2-resident) records once same-XCD placement is proven; 0: always write-through
    char* xch;                              // exchange area, XCH_BH_BYTES per (b,h)
    unsigned* flags;                        // [B NH][4] hand-over flags, one 128-byte line each (zeroed before every launch)
    unsigned* err;                          // host-mapped error word of the process: 1 + (b,h) of a cluster whose hand-over poll gave up (sticky)
    int fault;                              // DEBUG fault injection: workgroup 3 of every cluster leaves before its first hand-over
};


// ---- cluster form: exchange records ------------------------------------------------------------------------------------
// exchange record of one workgroup and step parity: the partial tile [64][PS] fp32 + the d(eta) partials of its two waves
constexpr int XCH_PART_BYTES = 64 * PS * 4;
constexpr int XCH_REC_BYTES = XCH_PART_BYTES + 2 * 64 * 4;
constexpr size_t XCH_BH_BYTES = 2 * 4 * (size_t)XCH_REC_BYTES;
constexpr int FLAG_STRIDE = 32;             // unsigned words: one 128-byte line per flag
static_assert(XCH_REC_BYTES % 128 == 0, "exchange records are line aligned");

}  // namespace b2
}  // namespace mfma
}  // namespace ttt
