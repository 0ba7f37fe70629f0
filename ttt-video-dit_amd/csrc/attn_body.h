// Segment-attention forward and dQ kernels (revision 2) as workgroup bodies written against a "wave backend", like the
// TTT bodies of ttt_lin16_body.h / ttt_mlp16_body.h: attn_v2.hip instantiates them with the gfx950 instructions,
// tests/emul/attn_emul.cpp with the lane-level emulator, so `pytest -m "not gpu"` executes this very code against the fp64
// attention oracle.
//
// Same algorithm, tiling and layout algebra as revision 1 (attn_fwd.hip: 8 waves x 32 query rows, keys / values
// streamed through LDS in tiles of 64, scores computed transposed so that a lane owns one query row, probabilities re-used
// in place as the B operand of the second product).  What changed, and why (static instruction mix of the revision-1 loops,
// tools/isa_mix.py - they are VALU-issue-bound, 16 / 24 MFMAs against ~270 / ~290 vector instructions per wave and tile):
//   * the ragged-tail mask (keys >= S exist only in the LAST tile, and only when S % 64 != 0) was if-converted by the
//     compiler into an index add + compare + select per score in EVERY tile: 97 of the forward loop's 239 VALU instructions
//     and 130 of the dQ loop's 256.  Here the mask sits behind a wave-uniform branch that is kept a real branch
//     (TTT_PIN_IN_BRANCH), so full tiles - every tile at the training geometry, S = 18 048 = 282 x 64 - never execute it.
//     (Peeling the last tile into a second instantiation of the loop body does the same but made the register allocator
//     spill in the 128-register dQ kernel; this form keeps revision 1's loop shape.)
// Everything else - the arithmetic, its order, the rounding points - is revision 1's (in the dQ kernel a masked score is set
// to -1e30 before the exponential instead of zeroing the probability after it: exp2(-1e30 ...) == 0 exactly), so the two
// revisions agree bit for bit (tests/test_attention_gpu.py::test_attention_v2_equals_v1).
//
// Backend contract: lane(), wave(), thread(), barrier(), exp2(), log(); tile_t = element pointer into LDS with lds_base(),
// ld<T>(ptr), st(ptr, v), tr(ptr) = ds_read_b64_tr_b16; mma3216(a, b, c) = v_mfma_f32_32x32x16_bf16; xor_read(v, mask) = the value
// of lane l ^ mask; any(bool) = wave-wide OR.  Device: attn_v2.hip (AttnDeviceWave); emulator: tests/emul/attn_emul.cpp.
#pragma once
#include <math.h>

#include "attn_types.h"
#include "ttt_wave_types.h"

#ifndef TTT_BODY_FN
#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
#define TTT_BODY_FN __device__ __forceinline__
#else
#define TTT_BODY_FN inline
#endif
#endif

namespace ttt {
namespace attnb {
using namespace ttt::wv;
using attn::BwdParams;
using attn::FwdParams;

constexpr int QB = 256, KB = 64;
constexpr int AS = 72;                  // row stride (elements) of a [64][64] bf16 tile read as 16-byte row fragments
constexpr int VS = 96;                  // row stride of the forward's V tile, read only through transposed reads (no bank conflicts)
constexpr int KT_ELEMS = 64 * AS, VT_ELEMS = 64 * VS;
constexpr int FWD_BUF_ELEMS = KT_ELEMS + VT_ELEMS, LDS_FWD = 2 * FWD_BUF_ELEMS * 2;
constexpr int DQ_BUF_ELEMS = 2 * KT_ELEMS, LDS_DQ = 2 * DQ_BUF_ELEMS * 2;     // K and V tiles, both stride 72
constexpr float LOG2E = 1.4426950408889634f;

// Keeps a wave-uniform branch a branch: a value routed through a volatile asm inside it cannot be turned into a select
// outside it (the compiler does not speculate volatile asm).  No instruction is emitted.
#if defined(__HIP_DEVICE_COMPILE__)
#define TTT_PIN_IN_BRANCH(x) asm volatile("" : "+v"(x))
#else
#define TTT_PIN_IN_BRANCH(x) ((void)0)
#endif

// BK::kPrio (round 6): s_setprio(1) .. s_setprio(0) around ONE MFMA cluster per kernel - the dS -> dQ cluster of dq_wide and the
// S / dP cluster of dkdv_staged - so that a wave entering its MFMA cluster is issued ahead of the waves of its SIMD that are in their
// exp / pack / LDS phases (cdna_hip_programming.md T5).  Measured on one MI355X at 48 heads x 18 048 tokens, interleaved rounds in
// one process (profiles/r6h_*, ms per backward): none 13.25 - 13.33; dQ's second cluster 12.76; its first cluster alone 13.25;
// both 12.68; dK/dV's SECOND cluster alone 13.46 (worse); THIS PAIR 12.55 / 13.02 (two boxes: -5.3 % / -2.3 %); all four clusters
// 12.51 / 13.07; static priorities by wave age 13.29.  Same bits as without (priorities only re-time).  The forward kernel
// (attn_fwd.hip) gains nothing from either cluster (4.64 - 4.66 against 4.66 ms) and has none; packed-fp32 forms of the exp chains
// (v_pk_mul_f32 on accumulator register pairs) LOST: hipcc splits a packed multiply by a uniform value whatever the source says, and
// forced through inline asm it costs scheduling freedom - 12.87 against 12.62 ms, removed.
TTT_BODY_FN int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
TTT_BODY_FN f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
TTT_BODY_FN bf16x8 pack(const f32x16& t, int s) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)t[8 * s + e];
    return r;
}
TTT_BODY_FN bf16x8 zero_frag() {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)0.0f;
    return r;
}
// LDS tiles are addressed through the backend's element pointer `BK::tile_t` (device: __bf16* into LDS, so that the compiler
// sees plain pointer arithmetic and folds the constant parts into the ds instructions' offset fields; emulator: an element
// offset), read with bk.ld<T>(ptr) / bk.tr(ptr) and written with bk.st(ptr, v).
// operand fragment, outer index = row (lane c reads 8 contiguous elements of row row0 + c), contraction over columns
template <class BK>
TTT_BODY_FN bf16x8 row_frag(BK& bk, typename BK::tile_t img, int stride, int row0, int col0, int l) {
    return bk.template ld<bf16x8>(img + ((row0 + (l & 31)) * stride + col0 + 8 * (l >> 5)));
}
// operand fragment through ds_read_b64_tr_b16: outer index = column (32 columns from col0), contraction over the rows
// r0..r0+3 and r1..r1+3 of the row-major image
template <class BK>
TTT_BODY_FN bf16x8 tr_frag(BK& bk, typename BK::tile_t img, int stride, int r0, int r1, int col0, int l) {
    const int i = l & 15, g1 = (l >> 4) & 1;
    const int off = (i >> 2) * stride + col0 + 16 * g1 + 4 * (i & 3);
    const bf16x4 lo = bk.tr(img + (r0 * stride + off));
    const bf16x4 hi = bk.tr(img + (r1 * stride + off));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// same, in the k-slot order of in-place fragment s of the 32-row block at row0 (ttt_mfma_dev.h: pi order)
template <class BK>
TTT_BODY_FN bf16x8 tr_frag_pi(BK& bk, typename BK::tile_t img, int stride, int row0, int s, int col0, int l) {
    const int h = l >> 5;
    return tr_frag(bk, img, stride, row0 + 16 * s + 4 * h, row0 + 16 * s + 8 + 4 * h, col0, l);
}

// ---- XOR-swizzled [64][64] tile (no padding): conflict-free for BOTH the 16-byte row fragments and the transposed reads ----------
// Element (r, col) lives at  r * 64 + ((col >> 3) ^ g(r)) * 8 + (col & 7),  g(r) = ((r >> 1) & 1) << 2 | (r >> 2) & 3.
// Why this g (lane groups of MI355X_MICROARCH.md; the emulator's bank model checks it, tests/test_emul_attention_cpu.py): a row is
// 32 dwords = half of the 64 banks, so rows of equal parity share a half.  A ds_read_b64_tr_b16 group reads rows r..r+3 (r % 4 == 0),
// 4 consecutive chunks each: rows r and r + 2 must take different 16-bank quarters -> bit 2 of g = bit 1 of the row.  A ds_read_b128
// group reads ONE chunk of 16 rows {0-3, 12-15, 20-27} (or {4-11, 16-19, 28-31}): the 8 rows of equal parity among them must take 8
// different chunks -> with bit 2 fixed by bit 1 of the row, bits 0-1 of g = bits 2-3 of the row do it.  The stride-72 layout serves
// the row fragments only (every transposed read 2-way conflicted: 22 - 24 % of the backward kernels' LDS passes).
TTT_BODY_FN int swz_g(int r) { return (((r >> 1) & 1) << 2) | ((r >> 2) & 3); }
template <bool SWZ>
TTT_BODY_FN int tile_off(int r, int col) {
    return SWZ ? r * 64 + ((((col >> 3) ^ swz_g(r)) << 3) | (col & 7)) : r * AS + col;
}
template <bool SWZ, class BK>
TTT_BODY_FN bf16x8 row_frag_s(BK& bk, typename BK::tile_t img, int row0, int col0, int l) {
    return bk.template ld<bf16x8>(img + tile_off<SWZ>(row0 + (l & 31), col0 + 8 * (l >> 5)));
}
template <bool SWZ, class BK>
TTT_BODY_FN bf16x8 tr_frag_pi_s(BK& bk, typename BK::tile_t img, int row0, int s, int col0, int l) {
    const int h = l >> 5, i = l & 15, g1 = (l >> 4) & 1;
    const int r0 = row0 + 16 * s + 4 * h + (i >> 2), col = col0 + 16 * g1 + 4 * (i & 3);
    const bf16x4 lo = bk.tr(img + tile_off<SWZ>(r0, col));
    const bf16x4 hi = bk.tr(img + tile_off<SWZ>(r0 + 8, col));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

struct KVStage {
    u32x4 k, v;
};
TTT_BODY_FN u32x4 zero_u4() {
    u32x4 z = {0u, 0u, 0u, 0u};
    return z;
}
// this thread's 16 bytes of the next K and V tiles: global -> registers (issued before the tile's MFMAs) ...
TTT_BODY_FN void kv_issue(KVStage& st, const __bf16* Kp, const __bf16* Vp, long k_ss, long v_ss, int kv0, int S, int tid) {
    const int row = tid >> 3, col = (tid & 7) * 8;
    const int key = kv0 + row;
    if (key < S) {
        st.k = *reinterpret_cast<const u32x4*>(Kp + (long)key * k_ss + col);
        st.v = *reinterpret_cast<const u32x4*>(Vp + (long)key * v_ss + col);
    } else {
        st.k = zero_u4();
        st.v = zero_u4();
    }
}
// ... -> LDS (parked after them)
template <class BK>
TTT_BODY_FN void kv_park(BK& bk, const KVStage& st, typename BK::tile_t k_img, typename BK::tile_t v_img, int v_stride, int tid) {
    const int row = tid >> 3, col = (tid & 7) * 8;
    bk.st(k_img + (row * AS + col), st.k);
    bk.st(v_img + (row * v_stride + col), st.v);
}

template <bool SWZ, class BK>
TTT_BODY_FN void kv_park_s(BK& bk, const KVStage& st, typename BK::tile_t k_img, typename BK::tile_t v_img, int tid) {
    const int row = tid >> 3, col = (tid & 7) * 8;
    bk.st(k_img + tile_off<SWZ>(row, col), st.k);
    bk.st(v_img + tile_off<SWZ>(row, col), st.v);
}

// workgroup -> (batch*head, block of 256 rows): blocks b, b+8, b+16, ... share an XCD; give each XCD whole heads, so a head's
// K and V are fetched from HBM once and then served by that XCD's L2
TTT_BODY_FN void head_of_block(int b, int nblk, int nbh, int& bh, int& blk) {
    if ((nbh & 7) == 0) {
        const int xcd = b & 7, idx = b >> 3;
        bh = xcd + 8 * (idx / nblk);
        blk = idx % nblk;
    } else {
        bh = b / nblk;
        blk = b % nblk;
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// O = softmax(Q K^T * scale) V and LSE for the 256 query rows of block qb of head bh (reference dit.py:196-198)
template <class BK>
TTT_BODY_FN void forward(BK& bk, const FwdParams& p, int bh, int qb) {
    const int tid = bk.thread(), wv = bk.wave(), l = bk.lane(), h = l >> 5, c = l & 31;
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    __bf16* Op = p.O + (long)bb * p.o_sb + (long)hh * p.o_sh;

    const int qrow = qb * QB + 32 * wv + c;    // this lane's query row
    const bool qvalid = qrow < p.S;
    bf16x8 Qf[4];                              // B operand: lane = query, 8 contiguous d per k-slice; kept for the whole loop
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
        Qf[kk] = qvalid ? *reinterpret_cast<const bf16x8*>(Qp + (long)qrow * p.q_ss + 16 * kk + 8 * h) : zero_frag();

    f32x16 O[2] = {zero16(), zero16()};        // O^T tiles: rows = d (32 db + row_of(r,h)), lane = query
    float m = -1e30f, lsum = 0.f;              // running max (raw score units), this half-wave's partial row sum
    const float sc = p.scale * LOG2E;
    const int nt = (p.S + KB - 1) / KB;
    KVStage st;

    const typename BK::tile_t lds = bk.lds_base();
    kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, 0, p.S, tid);
    kv_park(bk, st, lds, lds + KT_ELEMS, VS, tid);
    bk.barrier();
    for (int j = 0; j < nt; ++j) {
        const typename BK::tile_t Kt = lds + (j & 1) * FWD_BUF_ELEMS, Vt = Kt + KT_ELEMS;
        const bool more = j + 1 < nt;
        if (more) kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, (j + 1) * KB, p.S, tid);
        // ---- S^T = K Q^T : two key blocks of 32 ----
        f32x16 Sc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 acc = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = bk.mma3216(row_frag(bk, Kt, AS, 32 * kb, 16 * kk, l), Qf[kk], acc);
            Sc[kb] = acc;
        }
        if (!more && (p.S & (KB - 1))) {        // ragged last tile: keys >= S are masked.  Wave-uniform, and kept a real branch
            const int kv0 = j * KB;             // (if-converted it costs an add + compare + select per score in EVERY tile)
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = Sc[kb][r];
                    if (kv0 + 32 * kb + row_of(r, h) >= p.S) v = -1e30f;
                    TTT_PIN_IN_BRANCH(v);
                    Sc[kb][r] = v;
                }
        }
        // ---- online softmax: one query row per lane; the partner half-wave holds the other 32 keys of the tile ----
        float mt = Sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, Sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, Sc[1][r]);
        mt = fmaxf(mt, bk.xor_read(mt, 32));
        // rescale only when some row of this wave saw a larger maximum (alpha == 1 exactly for every lane otherwise)
        if (bk.any(mt > m)) {
            const float mn = fmaxf(m, mt);
            const float alpha = bk.exp2((m - mn) * sc);
            m = mn;
            lsum *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[db][r] *= alpha;
        }
        const float msc = m * sc;
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = bk.exp2(__builtin_fmaf(Sc[kb][r], sc, -msc));
                Sc[kb][r] = e;
                ps += e;
            }
        lsum += ps;
        // ---- O^T += V^T P^T ----
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack(Sc[kb], s);
                O[0] = bk.mma3216(tr_frag_pi(bk, Vt, VS, 32 * kb, s, 0, l), pf, O[0]);
                O[1] = bk.mma3216(tr_frag_pi(bk, Vt, VS, 32 * kb, s, 32, l), pf, O[1]);
            }
        if (more) {
            const typename BK::tile_t Kn = lds + ((j + 1) & 1) * FWD_BUF_ELEMS;
            kv_park(bk, st, Kn, Kn + KT_ELEMS, VS, tid);
        }
        bk.barrier();
    }

    // ---- epilogue: normalise, store O[q][d] (4 consecutive d per register group) and the log-sum-exp ----
    const float ltot = lsum + bk.xor_read(lsum, 32);
    const float inv = 1.0f / ltot;
    if (qvalid) {
        __bf16* orow = Op + (long)qrow * p.o_ss;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__bf16)(O[db][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * db + 8 * g + 4 * h) = v;
            }
        if (h == 0 && p.LSE) p.LSE[(long)bh * p.S + qrow] = m * p.scale + bk.log(ltot);
    }
}

// (Round 4 also measured forward() with 64 query rows per wave - `forward_wide`, the dq_wide idea applied to the forward: bit-identical
// on the emulator and on the device, 18 % fewer issue cycles and half the LDS instructions per row, and 1.0 % SLOWER than revision 1
// (4.30 against 4.26 ms at 48 heads x 18 048 tokens, profiles/r4zb_attn_fwd_wide_ab.txt): the forward needs its four waves per SIMD.
// Removed; the commit "attention forward with 64 query rows per wave" has it.)

// --------------------------------------------------------------------------------------------------------------------- dQ
// dQ = scale * dS K,  dS = P * (dP - Delta),  P = exp(S*scale - LSE),  dP = dO V^T, for the 256 query rows of block qb
template <class BK>
TTT_BODY_FN void dq(BK& bk, const BwdParams& p, int bh, int qb) {
    const int tid = bk.thread(), wv = bk.wave(), l = bk.lane(), h = l >> 5, c = l & 31;
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    const __bf16* dOp = p.dO + (long)bb * p.do_sb + (long)hh * p.do_sh;

    const int qrow = qb * QB + 32 * wv + c;
    const bool qvalid = qrow < p.S;
    bf16x8 Qf[4], Df[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        Qf[kk] = qvalid ? *reinterpret_cast<const bf16x8*>(Qp + (long)qrow * p.q_ss + 16 * kk + 8 * h) : zero_frag();
        Df[kk] = qvalid ? *reinterpret_cast<const bf16x8*>(dOp + (long)qrow * p.do_ss + 16 * kk + 8 * h) : zero_frag();
    }
    const float lse2 = qvalid ? p.LSE[(long)bh * p.S + qrow] * LOG2E : 1e30f;
    const float delta = qvalid ? p.Delta[(long)bh * p.S + qrow] : 0.f;
    const float sc = p.scale * LOG2E;
    f32x16 dQ[2] = {zero16(), zero16()};       // dQ^T tiles (rows = d, lane = query)
    const int nt = (p.S + KB - 1) / KB;
    KVStage st;

    const typename BK::tile_t lds = bk.lds_base();
    kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, 0, p.S, tid);
    kv_park(bk, st, lds, lds + KT_ELEMS, AS, tid);
    bk.barrier();
    for (int j = 0; j < nt; ++j) {
        const typename BK::tile_t Kt = lds + (j & 1) * DQ_BUF_ELEMS, Vt = Kt + KT_ELEMS;
        const bool more = j + 1 < nt;
        if (more) kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, (j + 1) * KB, p.S, tid);
        const bool ragged = !more && (p.S & (KB - 1));
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 Sc = zero16(), dP = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                Sc = bk.mma3216(row_frag(bk, Kt, AS, 32 * kb, 16 * kk, l), Qf[kk], Sc);
                dP = bk.mma3216(row_frag(bk, Vt, AS, 32 * kb, 16 * kk, l), Df[kk], dP);
            }
            if (ragged) {                       // keys >= S (zero-filled rows of the last tile) contribute nothing: P = 0 there.
#pragma unroll                                  // Wave-uniform and kept a real branch: exp2(-1e30) == 0 exactly
                for (int r = 0; r < 16; ++r) {
                    float v = Sc[r];
                    if (j * KB + 32 * kb + row_of(r, h) >= p.S) v = -1e30f;
                    TTT_PIN_IN_BRANCH(v);
                    Sc[r] = v;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = bk.exp2(__builtin_fmaf(Sc[r], sc, -lse2));
                dP[r] = pr * (dP[r] - delta);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 df = pack(dP, s);
                dQ[0] = bk.mma3216(tr_frag_pi(bk, Kt, AS, 32 * kb, s, 0, l), df, dQ[0]);
                dQ[1] = bk.mma3216(tr_frag_pi(bk, Kt, AS, 32 * kb, s, 32, l), df, dQ[1]);
            }
        }
        if (more) {
            const typename BK::tile_t Kn = lds + ((j + 1) & 1) * DQ_BUF_ELEMS;
            kv_park(bk, st, Kn, Kn + KT_ELEMS, AS, tid);
        }
        bk.barrier();
    }

    if (qvalid) {
        __bf16* row = p.dQ + (long)bb * p.dq_sb + (long)hh * p.dq_sh + (long)qrow * p.dq_ss;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__bf16)(dQ[db][4 * g + e] * p.scale);
                *reinterpret_cast<bf16x4*>(row + 32 * db + 8 * g + 4 * h) = v;
            }
    }
}

// dq() with NSUB key tiles of 64 per LDS stage, i.e. one workgroup barrier per NSUB tiles: 2 halves the number of barriers and
// doubles the loads in flight per stage.  Same arithmetic in the same order as dq() - bit-identical (tests/test_emul_attention_cpu.py) -;
// measured on the device in round 4 (two tiles per stage: -5 %).  Since round 5 the device runs dq_wide() (below) for dQ and
// dkdv_staged<12, true, 2>() for dK / dV; dq(), dq_staged() and dkdv() are the reference forms that the emulator tests hold those two
// to, bit for bit (tests/test_emul_attention_cpu.py).
template <int NSUB, bool SWZ = false, class BK>
TTT_BODY_FN void dq_staged(BK& bk, const BwdParams& p, int bh, int qb) {
    const int tid = bk.thread(), wv = bk.wave(), l = bk.lane(), h = l >> 5, c = l & 31;
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    const __bf16* dOp = p.dO + (long)bb * p.do_sb + (long)hh * p.do_sh;

    const int qrow = qb * QB + 32 * wv + c;
    const bool qvalid = qrow < p.S;
    bf16x8 Qf[4], Df[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        Qf[kk] = qvalid ? *reinterpret_cast<const bf16x8*>(Qp + (long)qrow * p.q_ss + 16 * kk + 8 * h) : zero_frag();
        Df[kk] = qvalid ? *reinterpret_cast<const bf16x8*>(dOp + (long)qrow * p.do_ss + 16 * kk + 8 * h) : zero_frag();
    }
    const float lse2 = qvalid ? p.LSE[(long)bh * p.S + qrow] * LOG2E : 1e30f;
    const float delta = qvalid ? p.Delta[(long)bh * p.S + qrow] : 0.f;
    const float sc = p.scale * LOG2E;
    f32x16 dQ[2] = {zero16(), zero16()};       // dQ^T tiles (rows = d, lane = query)
    const int nt = (p.S + KB - 1) / KB;        // key tiles of 64
    const int ns = (nt + NSUB - 1) / NSUB;     // LDS stages of NSUB tiles
    constexpr int STAGE_ELEMS = NSUB * DQ_BUF_ELEMS;
    KVStage st;                                // ONE tile in flight, as in dq(): the loads of tile u of the next stage are issued in
                                               // front of this stage's tile u and parked behind it (the next stage's buffer is idle)

    const typename BK::tile_t lds = bk.lds_base();
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {           // (tiles past the end of the sequence are zero-filled and never computed on)
        kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, u * KB, p.S, tid);
        kv_park_s<SWZ>(bk, st, lds + u * DQ_BUF_ELEMS, lds + u * DQ_BUF_ELEMS + KT_ELEMS, tid);
    }
    bk.barrier();
    for (int j = 0; j < ns; ++j) {
        const typename BK::tile_t stage = lds + (j & 1) * STAGE_ELEMS;
        const typename BK::tile_t nxt = lds + ((j + 1) & 1) * STAGE_ELEMS;
        const bool more = j + 1 < ns;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int jt = j * NSUB + u;       // key tile
            if (more) kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, ((j + 1) * NSUB + u) * KB, p.S, tid);
            if (jt < nt) {                     // (workgroup-uniform: false only in the last stage of an odd tile count)
            const typename BK::tile_t Kt = stage + u * DQ_BUF_ELEMS, Vt = Kt + KT_ELEMS;
            const bool ragged = jt + 1 == nt && (p.S & (KB - 1));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 Sc = zero16(), dP = zero16();
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    Sc = bk.mma3216(row_frag_s<SWZ>(bk, Kt, 32 * kb, 16 * kk, l), Qf[kk], Sc);
                    dP = bk.mma3216(row_frag_s<SWZ>(bk, Vt, 32 * kb, 16 * kk, l), Df[kk], dP);
                }
                if (ragged) {                   // keys >= S (zero-filled rows of the last tile) contribute nothing: P = 0 there.
#pragma unroll                                  // Wave-uniform and kept a real branch: exp2(-1e30) == 0 exactly
                    for (int r = 0; r < 16; ++r) {
                        float v = Sc[r];
                        if (jt * KB + 32 * kb + row_of(r, h) >= p.S) v = -1e30f;
                        TTT_PIN_IN_BRANCH(v);
                        Sc[r] = v;
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = bk.exp2(__builtin_fmaf(Sc[r], sc, -lse2));
                    dP[r] = pr * (dP[r] - delta);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 df = pack(dP, s);
                    dQ[0] = bk.mma3216(tr_frag_pi_s<SWZ>(bk, Kt, 32 * kb, s, 0, l), df, dQ[0]);
                    dQ[1] = bk.mma3216(tr_frag_pi_s<SWZ>(bk, Kt, 32 * kb, s, 32, l), df, dQ[1]);
                }
            }
            }
            if (more) kv_park_s<SWZ>(bk, st, nxt + u * DQ_BUF_ELEMS, nxt + u * DQ_BUF_ELEMS + KT_ELEMS, tid);
        }
        bk.barrier();
    }

    if (qvalid) {
        __bf16* row = p.dQ + (long)bb * p.dq_sb + (long)hh * p.dq_sh + (long)qrow * p.dq_ss;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__bf16)(dQ[db][4 * g + e] * p.scale);
                *reinterpret_cast<bf16x4*>(row + 32 * db + 8 * g + 4 * h) = v;
            }
    }
}

// dq_staged() with NQ blocks of 32 query rows per wave (round 4).  Why: every attention loop here feeds each MFMA with ONE 1-KiB
// fragment read from LDS - 32 cycles of the CU's LDS bandwidth share per SIMD for 32 cycles of MFMA pipe -, so the LDS array and the
// MFMA pipe are loaded equally and neither gets past ~40 % (profiles/r3p_wait_lds_summary.txt: LDS busy 29 - 42 % of the CU cycles
// with 3 - 4 waves per SIMD, MFMA busy 37 - 40 %).  A wave that owns 64 query rows uses every K / V fragment (row and transposed)
// for TWO MFMAs: half the LDS bytes per MFMA, at 2 waves of <= 256 registers per SIMD instead of 4 of 128.  The arithmetic of a
// query row and its order over the keys are those of dq(): bit-identical (tests/test_emul_attention_cpu.py).
// Workgroup = 8 waves = 256 * NQ query rows (block qb).
template <int NSUB, int NQ, class BK>
TTT_BODY_FN void dq_wide(BK& bk, const BwdParams& p, int bh, int qb) {
    const int tid = bk.thread(), wv = bk.wave(), l = bk.lane(), h = l >> 5, c = l & 31;
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    const __bf16* dOp = p.dO + (long)bb * p.do_sb + (long)hh * p.do_sh;

    int qrow[NQ];
    bool qvalid[NQ];
    bf16x8 Qf[NQ][4], Df[NQ][4];
    float lse2[NQ], delta[NQ];
    f32x16 dQ[NQ][2];                           // dQ^T tiles (rows = d, lane = query)
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
        qrow[qi] = qb * (QB * NQ) + 32 * NQ * wv + 32 * qi + c;
        qvalid[qi] = qrow[qi] < p.S;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            Qf[qi][kk] = qvalid[qi] ? *reinterpret_cast<const bf16x8*>(Qp + (long)qrow[qi] * p.q_ss + 16 * kk + 8 * h) : zero_frag();
            Df[qi][kk] = qvalid[qi] ? *reinterpret_cast<const bf16x8*>(dOp + (long)qrow[qi] * p.do_ss + 16 * kk + 8 * h) : zero_frag();
        }
        lse2[qi] = qvalid[qi] ? p.LSE[(long)bh * p.S + qrow[qi]] * LOG2E : 1e30f;
        delta[qi] = qvalid[qi] ? p.Delta[(long)bh * p.S + qrow[qi]] : 0.f;
        dQ[qi][0] = zero16();
        dQ[qi][1] = zero16();
    }
    const float sc = p.scale * LOG2E;
    const int nt = (p.S + KB - 1) / KB;        // key tiles of 64
    const int ns = (nt + NSUB - 1) / NSUB;     // LDS stages of NSUB tiles
    constexpr int STAGE_ELEMS = NSUB * DQ_BUF_ELEMS;
    KVStage st;

    const typename BK::tile_t lds = bk.lds_base();
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {
        kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, u * KB, p.S, tid);
        kv_park_s<false>(bk, st, lds + u * DQ_BUF_ELEMS, lds + u * DQ_BUF_ELEMS + KT_ELEMS, tid);
    }
    bk.barrier();
    for (int j = 0; j < ns; ++j) {
        const typename BK::tile_t stage = lds + (j & 1) * STAGE_ELEMS;
        const typename BK::tile_t nxt = lds + ((j + 1) & 1) * STAGE_ELEMS;
        const bool more = j + 1 < ns;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            const int jt = j * NSUB + u;       // key tile
            if (more) kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, ((j + 1) * NSUB + u) * KB, p.S, tid);
            if (jt < nt) {
            const typename BK::tile_t Kt = stage + u * DQ_BUF_ELEMS, Vt = Kt + KT_ELEMS;
            const bool ragged = jt + 1 == nt && (p.S & (KB - 1));
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
                f32x16 Sc[NQ], dP[NQ];
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi) { Sc[qi] = zero16(); dP[qi] = zero16(); }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const bf16x8 kf = row_frag_s<false>(bk, Kt, 32 * kb, 16 * kk, l), vf = row_frag_s<false>(bk, Vt, 32 * kb, 16 * kk, l);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        Sc[qi] = bk.mma3216(kf, Qf[qi][kk], Sc[qi]);
                        dP[qi] = bk.mma3216(vf, Df[qi][kk], dP[qi]);
                    }
                }
                if (ragged) {
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            float v = Sc[qi][r];
                            if (jt * KB + 32 * kb + row_of(r, h) >= p.S) v = -1e30f;
                            TTT_PIN_IN_BRANCH(v);
                            Sc[qi][r] = v;
                        }
                }
#pragma unroll
                for (int qi = 0; qi < NQ; ++qi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const float pr = bk.exp2(__builtin_fmaf(Sc[qi][r], sc, -lse2[qi]));
                        dP[qi][r] = pr * (dP[qi][r] - delta[qi]);
                    }
                if constexpr (BK::kPrio) bk.setprio(1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 k0 = tr_frag_pi_s<false>(bk, Kt, 32 * kb, s, 0, l), k1 = tr_frag_pi_s<false>(bk, Kt, 32 * kb, s, 32, l);
#pragma unroll
                    for (int qi = 0; qi < NQ; ++qi) {
                        const bf16x8 df = pack(dP[qi], s);
                        dQ[qi][0] = bk.mma3216(k0, df, dQ[qi][0]);
                        dQ[qi][1] = bk.mma3216(k1, df, dQ[qi][1]);
                    }
                }
                if constexpr (BK::kPrio) bk.setprio(0);
            }
            }
            if (more) kv_park_s<false>(bk, st, nxt + u * DQ_BUF_ELEMS, nxt + u * DQ_BUF_ELEMS + KT_ELEMS, tid);
        }
        bk.barrier();
    }

#pragma unroll
    for (int qi = 0; qi < NQ; ++qi)
        if (qvalid[qi]) {
            __bf16* row = p.dQ + (long)bb * p.dq_sb + (long)hh * p.dq_sh + (long)qrow[qi] * p.dq_ss;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (__bf16)(dQ[qi][db][4 * g + e] * p.scale);
                    *reinterpret_cast<bf16x4*>(row + 32 * db + 8 * g + 4 * h) = v;
                }
        }
}

// ------------------------------------------------------------------------------------------------------------------ dK, dV
// dV = P^T dO, dK = scale * dS^T Q for the 32 * NW keys of block kvb: each wave keeps 32 key rows of K and V as register-resident
// B operands and its dK / dV accumulator tiles over the whole loop over query tiles of 64 (Q, dO and the per-query LSE / Delta
// staged through LDS, double-buffered).  S = Q K^T and dP = dO V^T come out as (rows = query, lane = key) tiles, so P and dS are,
// in place, the A operands of  dV += P^T dO  and  dK += dS^T Q  (contraction over the tile's row index), with dO / Q fragments
// from transposed LDS reads.  Revision 1 (attn_bwd.hip) is <8 waves, ACC_INIT = false>.
//
// ACC_INIT: in this orientation LSE and Delta are PER-ROW values, i.e. 16 registers each per lane, live next to both
// accumulators (189 VGPRs in revision 1: one 8-wave workgroup per CU, and the loop is latency-bound at 2.7x its issue
// bound).  With ACC_INIT the staging code stores -LSE / scale and -Delta, and the two accumulators START from those rows
// instead of zero:  S' = Q K^T - LSE / scale,  dP' = dO V^T - Delta,  so  P = exp2(S' * scale * log2 e)  and  dS = P * dP'
// need no row value at all: 32 registers and 32 VALU instructions per tile fewer, which is what lets a workgroup run
// 12 waves (3 per SIMD, <= 168 registers).  fp32 accumulation of a start value of magnitude ~ |LSE| / scale ~ 1e2 costs
// ~1e-5 absolute in raw-score units, far below the bf16 rounding of P.
constexpr int DKV_BUF_ELEMS = 2 * KT_ELEMS + 2 * 64 * 2;       // Q tile, dO tile, lse[64], delta[64] (fp32 = 2 elements each)
constexpr int LDS_DKV = 2 * DKV_BUF_ELEMS * 2;

struct QStage {
    u32x4 q, d;
    float lse, del;
};
template <bool ACC_INIT>
TTT_BODY_FN void qstage_issue(QStage& st, const BwdParams& p, const __bf16* Qp, const __bf16* dOp, const float* lse, const float* del,
                              float inv_scale, int q0, int tid) {
    if (tid < 512) {
        const int row = tid >> 3, col = (tid & 7) * 8;
        const int q = q0 + row;
        if (q < p.S) {
            st.q = *reinterpret_cast<const u32x4*>(Qp + (long)q * p.q_ss + col);
            st.d = *reinterpret_cast<const u32x4*>(dOp + (long)q * p.do_ss + col);
        } else {
            st.q = zero_u4();
            st.d = zero_u4();
        }
    }
    if (tid < 64) {
        const int qq = q0 + tid;
        if (ACC_INIT) {
            st.lse = qq < p.S ? -lse[qq] * inv_scale : -1e30f;   // invalid rows: P = exp2(-huge) = 0
            st.del = qq < p.S ? -del[qq] : 0.f;
        } else {
            st.lse = qq < p.S ? lse[qq] * LOG2E : 1e30f;
            st.del = qq < p.S ? del[qq] : 0.f;
        }
    }
}
template <class BK>
TTT_BODY_FN void qstage_park(BK& bk, const QStage& st, typename BK::tile_t buf, int tid) {
    if (tid < 512) {
        const int row = tid >> 3, col = (tid & 7) * 8;
        bk.st(buf + (row * AS + col), st.q);
        bk.st(buf + (KT_ELEMS + row * AS + col), st.d);
    }
    if (tid < 64) {
        bk.st(buf + (2 * KT_ELEMS + 2 * tid), st.lse);
        bk.st(buf + (2 * KT_ELEMS + 2 * (64 + tid)), st.del);
    }
}
template <bool SWZ, class BK>
TTT_BODY_FN void qstage_park_s(BK& bk, const QStage& st, typename BK::tile_t buf, int tid) {
    if (tid < 512) {
        const int row = tid >> 3, col = (tid & 7) * 8;
        bk.st(buf + tile_off<SWZ>(row, col), st.q);
        bk.st(buf + (KT_ELEMS + tile_off<SWZ>(row, col)), st.d);
    }
    if (tid < 64) {
        bk.st(buf + (2 * KT_ELEMS + 2 * tid), st.lse);
        bk.st(buf + (2 * KT_ELEMS + 2 * (64 + tid)), st.del);
    }
}
// per-register row values of a (rows = query) tile: o[r] = src[base + row_of(r, h)], src = fp32 array `which` (0 lse, 1 delta)
template <class BK>
TTT_BODY_FN f32x16 rows_from_lds(BK& bk, typename BK::tile_t buf, int which, int base, int h) {
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = bk.template ld<f32x4>(buf + (2 * KT_ELEMS + 2 * (64 * which + base + 8 * q + 4 * h)));
    typedef float f32x8 __attribute__((ext_vector_type(8)));
    const f32x8 lo = __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    const f32x8 hi = __builtin_shufflevector(v[2], v[3], 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
}

template <int NW, bool ACC_INIT, class BK>
TTT_BODY_FN void dkdv(BK& bk, const BwdParams& p, int bh, int kvb) {
    const int tid = bk.thread(), wv = bk.wave(), l = bk.lane(), h = l >> 5, c = l & 31;
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    const __bf16* dOp = p.dO + (long)bb * p.do_sb + (long)hh * p.do_sh;
    const float* lse = p.LSE + (long)bh * p.S;
    const float* del = p.Delta + (long)bh * p.S;

    const int key0 = kvb * (32 * NW) + 32 * wv;      // this wave's first key
    const int krow = key0 + c;
    bf16x8 Kf[4], Vf[4];                              // B operands: lane = key, 8 contiguous d per k-slice
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        Kf[kk] = krow < p.S ? *reinterpret_cast<const bf16x8*>(Kp + (long)krow * p.k_ss + 16 * kk + 8 * h) : zero_frag();
        Vf[kk] = krow < p.S ? *reinterpret_cast<const bf16x8*>(Vp + (long)krow * p.v_ss + 16 * kk + 8 * h) : zero_frag();
    }
    f32x16 dK[2] = {zero16(), zero16()}, dV[2] = {zero16(), zero16()};   // tiles (rows = key, lane = d in block db)
    const float sc = p.scale * LOG2E, inv_scale = 1.0f / p.scale;

    const int nt = (p.S + 63) / 64;
    const typename BK::tile_t lds = bk.lds_base();
    QStage st;
    qstage_issue<ACC_INIT>(st, p, Qp, dOp, lse, del, inv_scale, 0, tid);
    qstage_park(bk, st, lds, tid);
    bk.barrier();

    for (int j = 0; j < nt; ++j) {
        const typename BK::tile_t buf = lds + (j & 1) * DKV_BUF_ELEMS;
        const typename BK::tile_t Qt = buf, Dt = buf + KT_ELEMS;
        const bool more = j + 1 < nt;
        if (more) qstage_issue<ACC_INIT>(st, p, Qp, dOp, lse, del, inv_scale, (j + 1) * 64, tid);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 Sc, dP;
            if (ACC_INIT) {
                Sc = rows_from_lds(bk, buf, 0, 32 * qb, h);      // -LSE / scale
                dP = rows_from_lds(bk, buf, 1, 32 * qb, h);      // -Delta
            } else {
                Sc = zero16();
                dP = zero16();
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                Sc = bk.mma3216(row_frag(bk, Qt, AS, 32 * qb, 16 * kk, l), Kf[kk], Sc);
                dP = bk.mma3216(row_frag(bk, Dt, AS, 32 * qb, 16 * kk, l), Vf[kk], dP);
            }
            if (ACC_INIT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pr = bk.exp2(Sc[r] * sc);
                    TTT_PIN_IN_BRANCH(pr);           // opaque to the SLP vectorizer, which otherwise pairs P / dS elements across the
                    Sc[r] = pr;                      // two tiles and then needs 40 moves / 16-bit shuffles per tile to un-pair them
                    float ds = pr * dP[r];
                    TTT_PIN_IN_BRANCH(ds);
                    dP[r] = ds;
                }
            } else {
                const f32x16 lseR = rows_from_lds(bk, buf, 0, 32 * qb, h);
                const f32x16 delR = rows_from_lds(bk, buf, 1, 32 * qb, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = bk.exp2(__builtin_fmaf(Sc[r], sc, -lseR[r]));
                    Sc[r] = pr;
                    dP[r] = pr * (dP[r] - delR[r]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack(Sc, s), df = pack(dP, s);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dV[db] = bk.mma3216(pf, tr_frag_pi(bk, Dt, AS, 32 * qb, s, 32 * db, l), dV[db]);
                    dK[db] = bk.mma3216(df, tr_frag_pi(bk, Qt, AS, 32 * qb, s, 32 * db, l), dK[db]);
                }
            }
        }
        if (more) qstage_park(bk, st, lds + ((j + 1) & 1) * DKV_BUF_ELEMS, tid);
        bk.barrier();
    }

    // epilogue: lane (c,h) register r of tile db holds element [key = key0 + row_of(r,h)][d = 32 db + c]
    __bf16* dKp = p.dK + (long)bb * p.dk_sb + (long)hh * p.dk_sh;
    __bf16* dVp = p.dV + (long)bb * p.dv_sb + (long)hh * p.dv_sh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + row_of(r, h);
            if (key < p.S) {
                dKp[(long)key * p.dk_ss + 32 * db + c] = (__bf16)(dK[db][r] * p.scale);
                dVp[(long)key * p.dv_ss + 32 * db + c] = (__bf16)dV[db][r];
            }
        }
}

// dkdv() with NSUB query tiles of 64 per LDS stage (see dq_staged)
template <int NW, bool ACC_INIT, int NSUB, bool SWZ = false, class BK>
TTT_BODY_FN void dkdv_staged(BK& bk, const BwdParams& p, int bh, int kvb) {
    const int tid = bk.thread(), wv = bk.wave(), l = bk.lane(), h = l >> 5, c = l & 31;
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    const __bf16* dOp = p.dO + (long)bb * p.do_sb + (long)hh * p.do_sh;
    const float* lse = p.LSE + (long)bh * p.S;
    const float* del = p.Delta + (long)bh * p.S;

    const int key0 = kvb * (32 * NW) + 32 * wv;      // this wave's first key
    const int krow = key0 + c;
    bf16x8 Kf[4], Vf[4];                              // B operands: lane = key, 8 contiguous d per k-slice
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        Kf[kk] = krow < p.S ? *reinterpret_cast<const bf16x8*>(Kp + (long)krow * p.k_ss + 16 * kk + 8 * h) : zero_frag();
        Vf[kk] = krow < p.S ? *reinterpret_cast<const bf16x8*>(Vp + (long)krow * p.v_ss + 16 * kk + 8 * h) : zero_frag();
    }
    f32x16 dK[2] = {zero16(), zero16()}, dV[2] = {zero16(), zero16()};   // tiles (rows = key, lane = d in block db)
    const float sc = p.scale * LOG2E, inv_scale = 1.0f / p.scale;

    const int nt = (p.S + 63) / 64;            // query tiles of 64
    const int ns = (nt + NSUB - 1) / NSUB;     // LDS stages of NSUB tiles (NSUB = 2: half the barriers; same arithmetic and order)
    constexpr int STAGE_ELEMS = NSUB * DKV_BUF_ELEMS;
    const typename BK::tile_t lds = bk.lds_base();
    QStage st;                                 // ONE tile in flight, as in dkdv() (see dq_staged)
#pragma unroll
    for (int u = 0; u < NSUB; ++u) {           // (query tiles past the end are zero / masked rows and never computed on)
        qstage_issue<ACC_INIT>(st, p, Qp, dOp, lse, del, inv_scale, u * 64, tid);
        qstage_park_s<SWZ>(bk, st, lds + u * DKV_BUF_ELEMS, tid);
    }
    bk.barrier();

    for (int j = 0; j < ns; ++j) {
        const typename BK::tile_t stage = lds + (j & 1) * STAGE_ELEMS;
        const typename BK::tile_t nxt = lds + ((j + 1) & 1) * STAGE_ELEMS;
        const bool more = j + 1 < ns;
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
        if (more) qstage_issue<ACC_INIT>(st, p, Qp, dOp, lse, del, inv_scale, ((j + 1) * NSUB + u) * 64, tid);
        if (j * NSUB + u < nt) {                         // (workgroup-uniform: false only in the last stage of an odd tile count)
        const typename BK::tile_t buf = stage + u * DKV_BUF_ELEMS;
        const typename BK::tile_t Qt = buf, Dt = buf + KT_ELEMS;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 Sc, dP;
            if (ACC_INIT) {
                Sc = rows_from_lds(bk, buf, 0, 32 * qb, h);      // -LSE / scale
                dP = rows_from_lds(bk, buf, 1, 32 * qb, h);      // -Delta
            } else {
                Sc = zero16();
                dP = zero16();
            }
            if constexpr (BK::kPrio) bk.setprio(1);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                Sc = bk.mma3216(row_frag_s<SWZ>(bk, Qt, 32 * qb, 16 * kk, l), Kf[kk], Sc);
                dP = bk.mma3216(row_frag_s<SWZ>(bk, Dt, 32 * qb, 16 * kk, l), Vf[kk], dP);
            }
            if constexpr (BK::kPrio) bk.setprio(0);
            if (ACC_INIT) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float pr = bk.exp2(Sc[r] * sc);
                    TTT_PIN_IN_BRANCH(pr);           // opaque to the SLP vectorizer, which otherwise pairs P / dS elements across the
                    Sc[r] = pr;                      // two tiles and then needs 40 moves / 16-bit shuffles per tile to un-pair them
                    float ds = pr * dP[r];
                    TTT_PIN_IN_BRANCH(ds);
                    dP[r] = ds;
                }
            } else {
                const f32x16 lseR = rows_from_lds(bk, buf, 0, 32 * qb, h);
                const f32x16 delR = rows_from_lds(bk, buf, 1, 32 * qb, h);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pr = bk.exp2(__builtin_fmaf(Sc[r], sc, -lseR[r]));
                    Sc[r] = pr;
                    dP[r] = pr * (dP[r] - delR[r]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack(Sc, s), df = pack(dP, s);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dV[db] = bk.mma3216(pf, tr_frag_pi_s<SWZ>(bk, Dt, 32 * qb, s, 32 * db, l), dV[db]);
                    dK[db] = bk.mma3216(df, tr_frag_pi_s<SWZ>(bk, Qt, 32 * qb, s, 32 * db, l), dK[db]);
                }
            }

        }
        }
        if (more) qstage_park_s<SWZ>(bk, st, nxt + u * DKV_BUF_ELEMS, tid);
        }
        bk.barrier();
    }

    // epilogue: lane (c,h) register r of tile db holds element [key = key0 + row_of(r,h)][d = 32 db + c]
    __bf16* dKp = p.dK + (long)bb * p.dk_sb + (long)hh * p.dk_sh;
    __bf16* dVp = p.dV + (long)bb * p.dv_sb + (long)hh * p.dv_sh;
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = key0 + row_of(r, h);
            if (key < p.S) {
                dKp[(long)key * p.dk_ss + 32 * db + c] = (__bf16)(dK[db][r] * p.scale);
                dVp[(long)key * p.dv_ss + 32 * db + c] = (__bf16)dV[db][r];
            }
        }
}

}  // namespace attnb
}  // namespace ttt
