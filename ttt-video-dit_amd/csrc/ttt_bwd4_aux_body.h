// TTT-MLP backward, revision 4: the per-step work of a DERIVER wave of the cluster sweep (ttt_mfma_bwd4.hip), written
// against the wave backend (see ttt_lin16_body.h for the idea) so that the CPU suite executes the same code on the lane-level
// emulator (tests/emul/bwd4_emul.cpp, tests/test_emul_bwd4_cpu.py) - this is where revision 4's new index algebra lives.
//
// A deriver wave pp (0 / 1) of the workgroup that sweeps hidden slice q owns the 32 hidden units Hp = [64 q + 32 pp, +32): the
// fp32 state tiles W1[:, Hp] (two tiles rows = f, lane = n) and W2[Hp, :] (two tiles rows = n, lane = f).  Per step i it
//   (1) turns the stored pre-activation Z1_i (bf16 T fragments) into X2 = gelu, D1 = gelu', D2 = gelu'' (T fragments);
//   (2) REVERSES the forward's state update:  W2_i = W2_{i+1} + (eta X2_i)^T gZ2_i   (the forward did W2' = W2 - (eta X2)^T gZ2);
//   (3) gX2 = gZ2 W2_i^T (contraction over f: W2^T by MFMA transposes of the packed state), gZ1 = gX2 * D1, M = gX2 * D2;
//   (4) W1_i = W1_{i+1} + (eta K_i)^T gZ1_i;
//   (5) transposes X2, D1, gZ1 to the N orientation (two MFMAs against identity fragments per 32 x 32 tile: exact for bf16);
//   (6) writes the operand fragments the compute waves read, lane-linear (fragment f of an array at f * 1 KiB + lane * 16):
//         R1: GZ1T | D1N | XT (N)      R2: W2 (rows = n, lane = f)            (consumed in stages S1 / S2 of step i)
//       and returns the T fragments D1 | M | X2 of the step (R4, consumed in S4a; the caller stores them when that region is
//       free) - the layouts are those of round 2's slot arrays (ttt_mfma_dev.h), which is what the compute waves still read;
//   (7) stores gZ1 (N) and the packed W1_i for the parallel dK / dQ tail kernel.
// Layout algebra: ttt_mfma_dev.h (pi slot order of in-place C-tile operands).  LDS tiles are row-major [64][TS] bf16.
#pragma once
#include "ttt_wave_types.h"

#ifndef TTT_WV_FN
#define TTT_WV_FN inline
#endif

namespace ttt {
namespace bwd4 {
using namespace ttt::wv;

constexpr int TS = 72;                                   // LDS row stride of a [64][64] bf16 tile, elements
constexpr int FRAG = 1024;                               // one fragment image: 64 lanes x 16 bytes
constexpr float GELU_A = 0.79788456f, GELU_C = 0.044715f, GELU_3AC = 0.1070322243f;
constexpr float GELU_K0 = -2.0f * GELU_A * 1.4426950408889634f, GELU_K1 = GELU_K0 * GELU_C;

TTT_WV_FN int fr_idx(int a, int b, int s) { return (a * 2 + b) * 2 + s; }
TTT_WV_FN f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
TTT_WV_FN bf16x8 pack(const f32x16& t, int s) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)t[8 * s + e];
    return r;
}
TTT_WV_FN bf16x8 cat(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
// identity fragment (as B operand) for the pi slot order: I[slot(h,e)][j=c] = (pi_s(h,e) == c)
TTT_WV_FN bf16x8 ident_pi(int s, int h, int c) {
    bf16x8 r;
#pragma unroll
    for (int e = 0; e < 8; ++e) r[e] = (__bf16)((16 * s + 8 * (e >> 2) + 4 * h + (e & 3)) == c ? 1.0f : 0.0f);
    return r;
}
// X^T of a 32 x 32 tile given as its two packed fragments
template <class BK>
TTT_WV_FN f32x16 transpose_tile(BK& bk, bf16x8 a0, bf16x8 a1, bf16x8 i0, bf16x8 i1) {
    f32x16 d = zero16();
    d = bk.mma3216(a0, i0, d);
    d = bk.mma3216(a1, i1, d);
    return d;
}
// "pi read": operand fragment (outer = this lane's row `row`, contraction = columns col0 + pi slots of s) of a row-major tile
template <class BK>
TTT_WV_FN bf16x8 pi_row(BK& bk, int tile_off, int row, int col0, int s, int h) {
    const int o = tile_off + (row * TS + col0 + 16 * s + 4 * h) * 2;
    return cat(bk.template lds_load<bf16x4>(o), bk.template lds_load<bf16x4>(o + 16));
}
// transposed read: operand fragment (outer = column col0 + (lane & 31), contraction = rows row0 + pi slots of s)
template <class BK>
TTT_WV_FN bf16x8 tr_pi(BK& bk, int tile_off, int row0, int s, int col0) {
    const int l = bk.lane(), h = l >> 5, i = l & 15, g1 = (l >> 4) & 1;
    const int off = tile_off + (((i >> 2) + row0 + 16 * s + 4 * h) * TS + col0 + 16 * g1 + 4 * (i & 3)) * 2;
    return cat(bk.tr_read(off), bk.tr_read(off + 8 * TS * 2));
}
// per-register row values of a fp32 [64] LDS vector: o[r] = v[base + row_of(r, h)]
template <class BK>
TTT_WV_FN f32x16 rows_from_lds(BK& bk, int vec_off, int base, int h) {
    f32x16 o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = bk.template lds_load<f32x4>(vec_off + (base + 8 * q + 4 * h) * 4);
        o[4 * q] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
    }
    return o;
}
template <class BK>
TTT_WV_FN void st_frag(BK& bk, int region_off, int idx, bf16x8 v) {
    bk.template lds_store<bf16x8>(region_off + idx * FRAG + bk.lane() * 16, v);
}

// tanh-GELU with first and second derivative (same forms as ttt_mfma_dev.h:gelu_fwd_grad2)
template <class BK>
TTT_WV_FN void gelu3(BK& bk, float x, float& y, float& dy, float& d2y) {
    const float x2 = x * x;
    const float s = bk.rcp(1.0f + bk.exp2(x * (x2 * GELU_K1 + GELU_K0)));     // (1 + tanh u) / 2
    const float t = 2.0f * s - 1.0f;
    const float q = 4.0f * s * (1.0f - s);
    const float du = GELU_A + GELU_3AC * x2;
    const float d2u = 2.0f * GELU_3AC * x;
    y = x * s;
    dy = s + 0.5f * x * q * du;
    d2y = q * du + 0.5f * x * q * (d2u - 2.0f * t * du * du);
}
template <class BK>
TTT_WV_FN void gelu2(BK& bk, float x, float& y, float& dy) {
    const float x2 = x * x;
    const float s = bk.rcp(1.0f + bk.exp2(x * (x2 * GELU_K1 + GELU_K0)));
    y = x * s;
    dy = (y - y * s) * (x2 * (2.0f * GELU_3AC) + 2.0f * GELU_A) + s;
}

// the deriver wave's carried state: W2[Hp, :] as two tiles (rows = n, lane = f in 32 b ..), fp32 (rounds 3 - 5 also carried W1[:, Hp])
struct AuxState {
    f32x16 W2t[2];
};
// T fragments [ti][s] of one quantity of a step for this wave's 32 hidden units (rows = t in registers, lane = n)
struct Frags4 {
    bf16x8 f[2][2];
};

// Z1b fragments -> X2b, gelu'(Z1b) fragments, written to their staging arrays (FR_D1B | FR_X2B order of the round-2 sweep)
template <class BK>
TTT_WV_FN void derive_z1b(BK& bk, const Frags4& z1b, int pp, int off_d1b, int off_x2b) {
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 x, d;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float y, dy;
                gelu2(bk, (float)z1b.f[ti][s][e], y, dy);
                x[e] = (__bf16)y;
                d[e] = (__bf16)dy;
            }
            st_frag(bk, off_d1b, fr_idx(ti, pp, s), d);
            st_frag(bk, off_x2b, fr_idx(ti, pp, s), x);
        }
}
// W2^T fragments (rows = f in 32 b .., lane = n) of the current state -> staging array in FR_W2T order [fj][ni][s]
template <class BK>
TTT_WV_FN void stage_w2t(BK& bk, const AuxState& st, int pp, int off_w2t, Frags4* keep = nullptr) {
    const int l = bk.lane(), h = l >> 5, c = l & 31;
    const bf16x8 I0 = ident_pi(0, h, c), I1 = ident_pi(1, h, c);
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const f32x16 t = transpose_tile(bk, pack(st.W2t[b], 0), pack(st.W2t[b], 1), I0, I1);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 v = pack(t, s);
            st_frag(bk, off_w2t, fr_idx(b, pp, s), v);
            if (keep) keep->f[b][s] = v;
        }
    }
}

// One reverse step, ordered for SHORT LIVE RANGES: the deriver role shares the kernel's 256 registers per lane with
// 32 (rounds 3 - 5: 64) registers of fp32 state, and a first version that kept the step's T fragments in registers until their staging region
// was free made hipcc spill ~200 dwords per step, each reload a serialised round trip (measured: 52 k cycles per step).  So:
//   * the T fragments D1 | M | X2 (R4 material, wanted only after the NEXT barrier Bd) are parked in a small per-wave global
//     scratch (12 KiB, rewritten every step: it lives in L2) as soon as a token tile is done, and fetched by stage_r4() later;
//   * the sigmoid of Z1 is evaluated twice (X2 for the W2 update, then gelu' / gelu'' per token tile) instead of carrying 32
//     registers across the W2 update (`opaque8` keeps hipcc from merging the two evaluations);
//   * W2^T is re-derived per token tile (two MFMAs per 32 x 32 block) instead of living through the whole step.
// LDS inputs: K tile, gZ2 tile (row-major [t][TS] bf16), eta[64] fp32 of the step; Z1: the step's pre-activation fragments.
// On return `st` is the state ENTERING the step, R1 / R2 are written, `r4_park` holds D1 | M | X2 ([array][ti][s] fragments of
// this wave), and gZ1 (T) has been stored to the step's slice region `g_slice` (byte offset off_gz1) for the tail kernel.
template <class BK>
TTT_WV_FN float gelu1(BK& bk, float x) { return x * bk.rcp(1.0f + bk.exp2(x * (x * x * GELU_K1 + GELU_K0))); }

constexpr int PARK_BYTES = 12 * FRAG;                    // per deriver wave
TTT_WV_FN int park_off(int arr, int ti, int s) { return ((arr * 2 + ti) * 2 + s) * FRAG; }

// `mid` (round 6) is called once, behind the W2 update (2) / R2 and in front of the token tiles (1b) .. (5): nothing a reverse step
// writes is read before the NEXT iteration (R1 by S1, R2 by S2, the parked fragments by stage_r4), and the tiles it reads belong to
// the step below, so the sweep lets its workgroup barrier Bc fall there - the token tiles then run beside the compute waves' S4a
// instead of in front of it (stage stamps of profiles/r6a: the 11.1 k cycles of reverse_step WERE the Bb .. Bc phase, and the
// derivers idled from Bc to Bd).  Other positions measured on one MI355X at NC = 804 (profiles/r6d_*, interleaved pairs, ms per
// backward): behind the whole step 10.88 (rounds 3 - 5), in front of it 10.66, HERE 10.22 - 10.30, behind the first token tile's
// gelu' / gelu'' 10.41, behind its W1 update 10.42, between the token tiles 10.39 - 10.47.
struct NoMid {
    TTT_WV_FN void operator()() const {}
};
// No W1 here since round 6: the per-step W1 existed ONLY for the dK / dQ tail, and the group-sequential tail (`mlp_bwd_tail5_kernel`)
// rebuilds it from the group's anchor itself - the deriver carries no W1 tiles (32 registers), runs no W1 update (8 MFMAs, 8 transposed
// reads per step) and stores no packed W1; gZ1 goes to the tail in the T orientation it is born in ([ti][nj][s] fragments like dZ1),
// the N orientation only to R1.
template <class BK, class Mid = NoMid>
TTT_WV_FN void reverse_step(BK& bk, AuxState& st, int pp, int tile_k, int tile_g, int vec_eta, const Frags4& Z1, int off_r1, int off_r2,
                            char* g_slice, int off_gz1, char* r4_park, Mid mid = Mid()) {
    const int l = bk.lane(), h = l >> 5, c = l & 31;
    constexpr int FRK = 8 * FRAG;

    // (1a), (2)  X2 = gelu(Z1) ; W2_i = W2_{i+1} + (eta X2)^T gZ2 :  A = eta-scaled X2 T fragment in place (m = n lane, k = t),
    //            B = gZ2 (k = t, j = f) by transposed reads
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const f32x16 etaR = rows_from_lds(bk, vec_eta, 32 * ti, h);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 xs;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const __bf16 y = (__bf16)gelu1(bk, (float)Z1.f[ti][s][e]);
                xs[e] = (__bf16)((float)y * etaR[8 * s + e]);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b) st.W2t[b] = bk.mma3216(xs, tr_pi(bk, tile_g, 32 * ti, s, 32 * b), st.W2t[b]);
        }
    }
    bk.stamp(0);
    // R2: W2_i (rows = n, lane = f), FR_W2 order [ni][fj][s]
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int s = 0; s < 2; ++s) st_frag(bk, off_r2, fr_idx(pp, b, s), pack(st.W2t[b], s));
    bk.stamp(1);
    mid();                           // (behind the W2 update / R2, in front of both token tiles)
    // (1b), (3), (4), (5) per token tile
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const bf16x8 I0 = ident_pi(0, h, c), I1 = ident_pi(1, h, c);
        f32x16 gx = zero16();                                     // gX2 = gZ2 W2^T   (rows = t, lane = n)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x16 t = transpose_tile(bk, pack(st.W2t[b], 0), pack(st.W2t[b], 1), I0, I1);      // W2^T block (rows = f, lane = n)
#pragma unroll
            for (int s = 0; s < 2; ++s) gx = bk.mma3216(pi_row(bk, tile_g, 32 * ti + c, 32 * b, s, h), pack(t, s), gx);
        }
        bk.stamp(2);
        bf16x8 g1p[2], x2[2], d1[2];
        {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 z = bk.opaque8(Z1.f[ti][s]);
                bf16x8 m;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float y, dy, d2y;
                    gelu3(bk, (float)z[e], y, dy, d2y);
                    const __bf16 db = (__bf16)dy;
                    x2[s][e] = (__bf16)y;
                    d1[s][e] = db;
                    const float g1 = gx[8 * s + e] * (float)db;            // gZ1, from the rounded gelu' the compute waves multiply with too
                    g1p[s][e] = (__bf16)g1;
                    m[e] = (__bf16)(gx[8 * s + e] * d2y);
                }
                *reinterpret_cast<bf16x8*>(r4_park + park_off(0, ti, s) + l * 16) = d1[s];
                *reinterpret_cast<bf16x8*>(r4_park + park_off(1, ti, s) + l * 16) = m;
                *reinterpret_cast<bf16x8*>(r4_park + park_off(2, ti, s) + l * 16) = x2[s];
            }
        }
        bk.stamp(3);
#pragma unroll
        for (int s = 0; s < 2; ++s) bk.store_stream(g_slice + off_gz1 + fr_idx(ti, pp, s) * FRAG + l * 16, g1p[s]);      // gZ1 (T) for the tail
        bk.stamp(4);
        // (5) N orientation: gZ1^T | gelu'(Z1)^T | X2^T  -> R1 (FR_GZ1T | FR_D1N | FR_XT order [nj][ti][s]), one tile at a time
        {
            const f32x16 t = transpose_tile(bk, g1p[0], g1p[1], I0, I1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 v = pack(t, s);
                st_frag(bk, off_r1, fr_idx(pp, ti, s), v);
            }
        }
        {
            const f32x16 t = transpose_tile(bk, d1[0], d1[1], I0, I1);
#pragma unroll
            for (int s = 0; s < 2; ++s) st_frag(bk, off_r1 + FRK, fr_idx(pp, ti, s), pack(t, s));
        }
        {
            const f32x16 t = transpose_tile(bk, x2[0], x2[1], I0, I1);
#pragma unroll
            for (int s = 0; s < 2; ++s) st_frag(bk, off_r1 + 2 * FRK, fr_idx(pp, ti, s), pack(t, s));
        }
        bk.stamp(5);
    }
}

// (Round 6 also measured a STAGE-major form of reverse_step<false> - every stage written for both token tiles so that the compiler can
// interleave the two chains, W2^T transposed once for both: emulator-green, same spill count, and no faster, 10.98 against 10.96 ms per
// backward - profiles/r6p_*.  With the one-sigmoid variant and the W1-free deriver that is three forms of LESS or better-overlapped deriver
// work that moved nothing.  Removed.)

// R4: the step's T fragments D1 | M | X2 (FR_D1 | FR_GX2 | FR_X2 order [ti][nj][s]) from the wave's parking area
template <class BK>
TTT_WV_FN void stage_r4(BK& bk, int pp, int off_r4, const char* r4_park) {
    constexpr int FRK = 8 * FRAG;
    const int l = bk.lane();
    bf16x8 v[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) v[k] = *reinterpret_cast<const bf16x8*>(r4_park + k * FRAG + l * 16);
#pragma unroll
    for (int arr = 0; arr < 3; ++arr)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int s = 0; s < 2; ++s) st_frag(bk, off_r4 + arr * FRK, fr_idx(ti, pp, s), v[(arr * 2 + ti) * 2 + s]);
}

}  // namespace bwd4
}  // namespace ttt
