// TTT-Linear scans at mini-batches of 16 tokens (F = 64), written against a "wave backend" BK: one wavefront owns one
// (batch, head) scan - the whole state (W1 64x64 fp32 = 16 accumulator tiles, b1) fits its registers - so a step has no
// workgroup barrier at all.  The backend supplies the gfx950 primitives (ttt_mfma16.hip: DeviceWave) or, in the CPU tests,
// their lane-by-lane emulation (tests/emul/wave_emul.h), so the SAME body is what runs on the GPU and what the CPU suite
// checks against the oracle:
//   bk.lane()                       0..63
//   bk.mma32(a, b, c) / mma16       v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x16_bf16
//   bk.tr_read(byte_addr)           ds_read_b64_tr_b16
//   bk.lds<T>(byte_off)             reference into the wave's private LDS region ; bk.lds_ptr(byte_off) generic pointer to it
//   bk.lds_fence()                  ordering point between this wave's LDS writes and reads (free on the device)
//   bk.sum16(x)                     sum over the 16 lanes of a DPP row ; bk.xor_add(x, m) = x + x[lane ^ m]
//   bk.rsq(x), bk.opaque(i)
// Layout algebra (lane (g, i) = (l >> 4, l & 15)): a 16x16 fp32 tile holds D[4g + r][i], r = 0..3.  A tile X (rows = R,
// lane = C) is in place an mma16 operand contracting over R (k-slot e = row 4g + e) and, stacked with a second tile along
// R, an mma32 operand (k-slot (g, e) = row 4g + e of the first (e < 4) / second tile: "rho" order; the partner operand
// presents the same order from a row-major LDS tile with two 8-byte reads at columns c0 + 4g and c0 + 16 + 4g).  A
// contraction over the LANE index goes through an LDS image [lane index][row index] and transposed reads.
// Math: reference ops/ttt_linear.py:8-54 (forward, primal form), kernels/linear_backward.py:73-197 (backward structure),
// SURVEY.md Appendix A.
#pragma once
#include "ttt_wave_types.h"

#ifndef TTT_WV_FN
#define TTT_WV_FN inline
#endif

namespace ttt {
namespace lin16 {
using namespace ttt::wv;

constexpr int TS = 72;                       // row stride (elements) of a padded [16][64] bf16 tile
constexpr int TILE = 16 * TS;                // elements
constexpr int IS = 24;                       // row stride of an image [64 f][16 t] bf16
// wave-private LDS map (bytes)
constexpr int L_K = 0, L_V = L_K + 2 * TILE * 2, L_Q = L_V + 2 * TILE * 2, L_D = L_Q + 2 * TILE * 2;   // K, V, Q, dOut x 2 buffers
constexpr int L_IMG = L_D + 2 * TILE * 2;    // 2 images [64][IS] bf16
constexpr int IMG_BYTES = 64 * IS * 2;
constexpr int L_ETA = L_IMG + 2 * IMG_BYTES; // [2][16] fp32
constexpr int WAVE_LDS = L_ETA + 2 * 16 * 4;   // what forward() uses
// backward() only: a [64][TRS] bf16 image for transposing a 64 x 64 matrix of operand values, and the packed state that
// ends a checkpoint group (16 operand fragments x 64 lanes x 16 bytes)
constexpr int TRS = 72;
constexpr int L_TR = WAVE_LDS, L_WHI = L_TR + 64 * TRS * 2;
// (Keeping a group's per-step state slots in LDS instead of the caller's L2-resident scratch was measured in round 2 -
// 14.7 vs 13.9 ms per backward at the 3 s geometry - and removed.)
constexpr int WAVE_LDS_BWD = L_WHI + 16 * 1024;
static_assert(WAVE_LDS_BWD <= 160 * 1024, "LDS budget");
static_assert(WAVE_LDS % 16 == 0 && L_WHI % 16 == 0, "alignment");

TTT_WV_FN f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
TTT_WV_FN bf16x4 pack4(f32x4 v) {
    bf16x4 r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return r;
}
TTT_WV_FN bf16x8 stack(f32x4 a, f32x4 b) {
    bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return r;
}
TTT_WV_FN bf16x8 cat(bf16x4 lo, bf16x4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }

// rho-order operand (lane = this lane's row i of the tile, k = columns c0 .. c0 + 31) of a row-major [16][TS] LDS tile
template <class BK>
TTT_WV_FN bf16x8 rho_read(BK& bk, int tile_off, int c0) {
    const int g = bk.lane() >> 4, i = bk.lane() & 15;
    const int o = tile_off + (i * TS + c0 + 4 * g) * 2;
    return cat(bk.template lds<bf16x4>(o), bk.template lds<bf16x4>(o + 32));
}
// transposed read: lane (g, i) gets img[row0 + 4g + e][col0 + i], e = 0..3   (operand with outer = column, k = row)
template <class BK>
TTT_WV_FN bf16x4 tr4(BK& bk, int img_off, int stride, int row0, int col0) {
    const int g = bk.lane() >> 4, i = bk.lane() & 15;
    return bk.tr_read(img_off + ((row0 + 4 * g + (i >> 2)) * stride + col0 + 4 * (i & 3)) * 2);
}
// per-token-row reduction over the 64 features of tiles (rows = t, lane = f): in-lane over the 4 feature tiles, then the
// 16 lanes of the row
template <class BK>
TTT_WV_FN f32x4 rowsum64(BK& bk, const f32x4 (&v)[4]) {
    f32x4 s = v[0] + v[1] + v[2] + v[3];
#pragma unroll
    for (int r = 0; r < 4; ++r) s[r] = bk.sum16(s[r]);
    return s;
}
// column sums over the 16 tokens of a (rows = t, lane = f) tile, in fp32: the lane's four rows, then the four row groups (lanes
// l ^ 16, l ^ 32) - replicated over the row groups like the first row of a ones-MFMA product.  Round 4: the bias gradients
// db1 are summed from the fp32 tiles instead of by a ones-MFMA over their bf16 packs (the same finding as in the TTT-MLP sweep,
// tests/test_rounding_budget_cpu.py: db enters d(eta) of every token of every earlier step).
template <class BK>
TTT_WV_FN float colsum16(BK& bk, const f32x4& v) {
    return bk.xor_add(bk.xor_add((v[0] + v[1]) + (v[2] + v[3]), 16), 32);
}
// (rows = t, lane = f) tiles -> [16 t][64 f] bf16 in global memory, 8 bytes per lane and feature block, through the [f][t]
// image `img_off`
template <class BK>
TTT_WV_FN void store_rows(BK& bk, int img_off, const f32x4 (&v)[4], __bf16* dst) {
    const int g = bk.lane() >> 4, i = bk.lane() & 15;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) bk.template lds<bf16x4>(img_off + ((16 * fb + i) * IS + 4 * g) * 2) = pack4(v[fb]);
    bk.lds_fence();
#pragma unroll
    for (int fb = 0; fb < 4; ++fb)
        *reinterpret_cast<bf16x4*>(dst + (size_t)i * 64 + 16 * fb + 4 * g) = tr4(bk, img_off, IS, 16 * fb, 0);
}
// one 2-KiB tile: 2 x 16 bytes per lane (rows l >> 3 and 8 + (l >> 3))
struct Stage {
    u32x4 lo, hi;
};
template <class BK>
TTT_WV_FN void stage_request(BK& bk, Stage& st, const __bf16* src_tile) {
    const int l = bk.lane();
    const __bf16* s = src_tile + (size_t)(l >> 3) * 64 + (l & 7) * 8;
    st.lo = *reinterpret_cast<const u32x4*>(s);
    st.hi = *reinterpret_cast<const u32x4*>(s + 512);
}
template <class BK>
TTT_WV_FN void stage_park(BK& bk, const Stage& st, int tile_off) {
    const int l = bk.lane();
    const int o = tile_off + ((l >> 3) * TS + (l & 7) * 8) * 2;
    bk.template lds<u32x4>(o) = st.lo;
    bk.template lds<u32x4>(o + 8 * TS * 2) = st.hi;
}

// LayerNorm statistics of z (rows = t, lane = f): z <- x_hat, returns 1/std per token row
template <class BK>
TTT_WV_FN f32x4 normalize_rows(BK& bk, f32x4 (&z)[4], float eps) {
    const f32x4 mu = rowsum64(bk, z) * (1.0f / 64.0f);
    f32x4 d2[4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) { z[fb] -= mu; d2[fb] = z[fb] * z[fb]; }
    const f32x4 var = rowsum64(bk, d2) * (1.0f / 64.0f);
    f32x4 rstd;
#pragma unroll
    for (int r = 0; r < 4; ++r) rstd[r] = bk.rsq(var[r] + eps);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) z[fb] *= rstd;
    return rstd;
}


// Fused LayerNorm + L2-loss gradient of one mini-batch (ops/utils.py:17-45), tiles (rows = t, lane = f).
// in: z = Z1 (bias included), tg = V - K.  out: xh, go = gamma xh + beta - tg, gz = dl/dZ1, rstd, s2g = sum_f (go gamma) xh
struct InnerGrad {
    f32x4 xh[4], go[4], gz[4];
    f32x4 rstd, s2g;
};
template <class BK>
TTT_WV_FN void inner_grad(BK& bk, const f32x4 (&z)[4], const f32x4 (&tg)[4], const float (&gam)[4], const float (&bet)[4], float eps,
                          InnerGrad& o) {
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) o.xh[fb] = z[fb];
    o.rstd = normalize_rows(bk, o.xh, eps);
    f32x4 gxh[4], gxx[4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        o.go[fb] = gam[fb] * o.xh[fb] + bet[fb] - tg[fb];
        gxh[fb] = o.go[fb] * gam[fb];
        gxx[fb] = gxh[fb] * o.xh[fb];
    }
    const f32x4 s1 = rowsum64(bk, gxh);
    o.s2g = rowsum64(bk, gxx);
    const f32x4 sc = o.rstd * (1.0f / 64.0f);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) o.gz[fb] = (64.0f * gxh[fb] - s1 - o.xh[fb] * o.s2g) * sc;
}

// ===================================================================================================================
// forward scan of (b, h) = bh
template <class BK>
TTT_WV_FN void forward(BK& bk, const Lin16Params& p, int bh) {
    const int l0 = bk.lane();
    const int NC = p.NC, G = p.G, head = bh % p.NH;

    f32x4 W1t[4][4];     // [fa][fb]  W1[16fa + 4g + r][16fb + i]     (rows = f_in, lane = f_out)
    float b1v[4], gam[4], bet[4];
    {
        const int g = l0 >> 4, i = l0 & 15;
        const float* W1g = p.W1 + (size_t)bh * 64 * 64;
#pragma unroll
        for (int fa = 0; fa < 4; ++fa)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int r = 0; r < 4; ++r) W1t[fa][fb][r] = W1g[(size_t)(16 * fa + 4 * g + r) * 64 + 16 * fb + i];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            b1v[fb] = p.b1[(size_t)bh * 64 + 16 * fb + i];
            gam[fb] = p.ln_w[(size_t)head * 64 + 16 * fb + i];
            bet[fb] = p.ln_b[(size_t)head * 64 + 16 * fb + i];
        }
    }
    const bf16x4 ONES = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
    bf16x4 IDP, IDN;     // +-identity as mma16 A operand: lane = t = i, k-slot e = token 4g + e
    {
        const int g = l0 >> 4, i = l0 & 15;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            IDP[e] = (__bf16)((4 * g + e) == i ? 1.0f : 0.0f);
            IDN[e] = (__bf16)((4 * g + e) == i ? -1.0f : 0.0f);
        }
    }
    bf16x8 W1F[2][4];    // [ks][fb] packed entering state, carried across steps
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) W1F[ks][fb] = stack(W1t[2 * ks][fb], W1t[2 * ks + 1][fb]);

    const size_t tile0 = (size_t)bh * NC;
    Stage sk, sv, sq;
    unsigned short pe;
    stage_request(bk, sk, p.XK + tile0 * 1024);
    stage_request(bk, sv, p.XV + tile0 * 1024);
    stage_request(bk, sq, p.XQ + tile0 * 1024);
    pe = *reinterpret_cast<const unsigned short*>(p.eta + tile0 * 16 + (l0 & 15));
    stage_park(bk, sk, L_K); stage_park(bk, sv, L_V); stage_park(bk, sq, L_Q);
    if (l0 < 16) bk.template lds<float>(L_ETA + l0 * 4) = (float)*reinterpret_cast<const __bf16*>(&pe);
    bk.lds_fence();

    for (int it = 0; it < NC; ++it) {
        const size_t tile = tile0 + it;
        const int buf = it & 1;
        const int l = bk.opaque(l0), g = l >> 4, i = l & 15;
        const int Kt = L_K + buf * TILE * 2, Vt = L_V + buf * TILE * 2, Qt = L_Q + buf * TILE * 2;
        {   // next step's inputs, parked at the end of this step
            const size_t tn = tile0 + (it + 1 < NC ? it + 1 : it);
            stage_request(bk, sk, p.XK + tn * 1024);
            stage_request(bk, sv, p.XV + tn * 1024);
            stage_request(bk, sq, p.XQ + tn * 1024);
            pe = *reinterpret_cast<const unsigned short*>(p.eta + tn * 16 + (l & 15));
        }
        if (it % G == 0) {      // checkpoint: state entering step `it` (linear_triton.py:84-97)
            const size_t ck = (size_t)bh * p.K + it / G;
            float* W1g = p.W1c + ck * 64 * 64;
#pragma unroll
            for (int fa = 0; fa < 4; ++fa)
#pragma unroll
                for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) W1g[(size_t)(16 * fa + 4 * g + r) * 64 + 16 * fb + i] = W1t[fa][fb][r];
            if (g == 0)
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) p.b1c[ck * 64 + 16 * fb + i] = b1v[fb];
        }
        // ---- Z1 = K W1 + b1 ; target = V - K, both (rows = t, lane = f) -----------------------------------------------
        const bf16x8 kA0 = rho_read(bk, Kt, 0), kA1 = rho_read(bk, Kt, 32);
        const f32x4 eta4 = bk.template lds<f32x4>(L_ETA + (buf * 16 + 4 * g) * 4);          // eta of token rows 4g .. 4g+3
        bf16x4 kT[4];
        f32x4 z[4], tg[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            kT[fb] = tr4(bk, Kt, TS, 0, 16 * fb);                                            // lane = f, k = t
            f32x4 a = zero4();
            a = bk.mma32(kA0, W1F[0][fb], a);
            a = bk.mma32(kA1, W1F[1][fb], a);
            z[fb] = a + b1v[fb];
            tg[fb] = bk.mma16(IDN, kT[fb], bk.mma16(IDP, tr4(bk, Vt, TS, 0, 16 * fb), zero4()));   // exact V - K
        }
        // ---- fused LayerNorm + L2 backward per token row ; Gs = -eta gZ1 ------------------------------------------------
        bf16x4 gzp[4];
        {
            InnerGrad ig;
            inner_grad(bk, z, tg, gam, bet, p.eps, ig);
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) gzp[fb] = pack4(ig.gz[fb] * (-eta4));                       // lane = f, k = t
        }
        // ---- W1 += K^T Gs ; b1 += colsum Gs ; repack -----------------------------------------------------------------------
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            b1v[fb] += bk.mma16(ONES, gzp[fb], zero4())[0];
#pragma unroll
            for (int fa = 0; fa < 4; ++fa) W1t[fa][fb] = bk.mma16(kT[fa], gzp[fb], W1t[fa][fb]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) W1F[ks][fb] = stack(W1t[2 * ks][fb], W1t[2 * ks + 1][fb]);
        // ---- Z1b = Q W1' + b1' ; LayerNorm ; + Q -> XQW --------------------------------------------------------------------------
        {
            const bf16x8 qA0 = rho_read(bk, Qt, 0), qA1 = rho_read(bk, Qt, 32);
            f32x4 y[4], qc[4];
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                f32x4 a = zero4();
                a = bk.mma32(qA0, W1F[0][fb], a);
                a = bk.mma32(qA1, W1F[1][fb], a);
                y[fb] = a + b1v[fb];
                qc[fb] = bk.mma16(IDP, tr4(bk, Qt, TS, 0, 16 * fb), zero4());                // exact Q in the accumulator layout
            }
            normalize_rows(bk, y, p.eps);
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) y[fb] = qc[fb] + gam[fb] * y[fb] + bet[fb];
            store_rows(bk, L_IMG, y, p.out + tile * 1024);
        }
        const int nb = buf ^ 1;
        stage_park(bk, sk, L_K + nb * TILE * 2); stage_park(bk, sv, L_V + nb * TILE * 2); stage_park(bk, sq, L_Q + nb * TILE * 2);
        if (l < 16) bk.template lds<float>(L_ETA + (nb * 16 + l) * 4) = (float)*reinterpret_cast<const __bf16*>(&pe);
        bk.lds_fence();
    }
}

// ===================================================================================================================
// backward of the scan of (b, h) = bh.  Checkpoint groups are walked from the last to the first; a group is first
// re-run forward, parking the state entering every step as packed MFMA operands in BOTH orientations in the caller's
// scratch (16 KiB per step = the W1_init_group storage; the state that ends the group goes to LDS), then walked in reverse
// carrying dW1, db1, dgamma, dbeta.  Products that contract over the feature a tile keeps in its LANES (dZ W^T, gZ1 dW^T)
// take their token-side operand from an LDS image + transposed reads and their weight-side operand from a transposed
// pack made the same way (one fp32 accumulator copy per matrix; a second, transposed accumulator copy was 128 registers
// too many for one wave).  Math: oracle/ttt_oracle.py:_lin_step_bwd.
constexpr int SLOT_BYTES = 16 * 1024;           // [16 operand fragments][64 lanes][16 bytes]: 0..7 (rows = f_in), 8..15 transposed

template <class BK>
TTT_WV_FN void st_pack(BK& bk, char* slot, int idx, bf16x8 v) { *reinterpret_cast<bf16x8*>(slot + ((size_t)idx * 64 + bk.lane()) * 16) = v; }
template <class BK>
TTT_WV_FN bf16x8 ld_pack(BK& bk, const char* slot, int idx) { return *reinterpret_cast<const bf16x8*>(slot + ((size_t)idx * 64 + bk.lane()) * 16); }

// token-side operand of a contraction over the features of X (rows = t, lane = f): lane = t, k = features 32ks .. 32ks+31
// in rho order, through the [f][t] image at img_off.  The packed X goes in (it is what the update MFMAs use as well).
template <class BK>
TTT_WV_FN void image_of(BK& bk, int img_off, const bf16x4 (&xp)[4], bf16x8 (&a)[2]) {
    const int g = bk.lane() >> 4, i = bk.lane() & 15;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) bk.template lds<bf16x4>(img_off + ((16 * fb + i) * IS + 4 * g) * 2) = xp[fb];
    bk.lds_fence();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) a[ks] = cat(tr4(bk, img_off, IS, 32 * ks, 0), tr4(bk, img_off, IS, 32 * ks + 16, 0));
}
// weight-side operands of a contraction over the f_out index of M (tiles T[fa][fb]: rows = f_in, lane = f_out):
// out[ks][fa]: lane = f_in of block fa, k = f_out in [32ks, 32ks + 32) in rho order.  Image [f_out][f_in] at L_TR.
template <class BK>
TTT_WV_FN void transposed_packs(BK& bk, const f32x4 (&T)[4][4], bf16x8 (&out)[2][4]) {
    const int g = bk.lane() >> 4, i = bk.lane() & 15;
#pragma unroll
    for (int fa = 0; fa < 4; ++fa)
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) bk.template lds<bf16x4>(L_TR + ((16 * fb + i) * TRS + 16 * fa + 4 * g) * 2) = pack4(T[fa][fb]);
    bk.lds_fence();
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int fa = 0; fa < 4; ++fa) out[ks][fa] = cat(tr4(bk, L_TR, TRS, 32 * ks, 16 * fa), tr4(bk, L_TR, TRS, 32 * ks + 16, 16 * fa));
}

template <class BK>
TTT_WV_FN void backward(BK& bk, const Lin16Params& p, int bh) {
    const int l0 = bk.lane();
    const int NC = p.NC, G = p.G, K = p.K, head = bh % p.NH;
    const size_t tile0 = (size_t)bh * NC;
    char* scr_w = p.scratch_w + (size_t)bh * G * SLOT_BYTES;
    float* scr_b = p.scratch_b + (size_t)bh * G * 64;

    float gam[4], bet[4];
    f32x4 dWt[4][4];      // [fa][fb]  dW1[16fa + 4g + r][16fb + i]    (rows = f_in, lane = f_out)
    float db[4];          // db1[16fb + i]
    float dgam[4] = {0.f, 0.f, 0.f, 0.f}, dbet[4] = {0.f, 0.f, 0.f, 0.f};     // per-lane partial sums over this lane's token rows
    {
        const int g = l0 >> 4, i = l0 & 15;
        const float* dWl = p.dW1_last + (size_t)bh * 64 * 64;
#pragma unroll
        for (int fa = 0; fa < 4; ++fa)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dWt[fa][fb][r] = dWl[(size_t)(16 * fa + 4 * g + r) * 64 + 16 * fb + i];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            db[fb] = p.db1_last[(size_t)bh * 64 + 16 * fb + i];
            gam[fb] = p.ln_w[(size_t)head * 64 + 16 * fb + i];
            bet[fb] = p.ln_b[(size_t)head * 64 + 16 * fb + i];
        }
    }
    const bf16x4 ONES = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
    bf16x4 IDP, IDN;
    {
        const int g = l0 >> 4, i = l0 & 15;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            IDP[e] = (__bf16)((4 * g + e) == i ? 1.0f : 0.0f);
            IDN[e] = (__bf16)((4 * g + e) == i ? -1.0f : 0.0f);
        }
    }
    // tiles of step s live in buffer (s & 1) of their kind; eta likewise
    Stage sk, sv, sq, sd;
    unsigned short pe = 0;
    {   // first tiles of the last group's recompute pass
        const int s0 = (K - 1) * G;
        stage_request(bk, sk, p.XK + (tile0 + s0) * 1024);
        stage_request(bk, sv, p.XV + (tile0 + s0) * 1024);
        pe = *reinterpret_cast<const unsigned short*>(p.eta + (tile0 + s0) * 16 + (l0 & 15));
        stage_park(bk, sk, L_K + (s0 & 1) * TILE * 2);
        stage_park(bk, sv, L_V + (s0 & 1) * TILE * 2);
        if (l0 < 16) bk.template lds<float>(L_ETA + ((s0 & 1) * 16 + l0) * 4) = (float)*reinterpret_cast<const __bf16*>(&pe);
        bk.lds_fence();
    }

    for (int k = K - 1; k >= 0; --k) {
        const int lo = k * G, hi = (lo + G < NC) ? lo + G : NC;
        float b1hi[4];                       // bias that ends the group
        // ================= re-run the group forward, parking the state entering each step ================================
        {
            f32x4 W1t[4][4];                 // [fa][fb] (rows = f_in, lane = f_out)
            float b1v[4];
            {
                const int g = l0 >> 4, i = l0 & 15;
                const float* W1g = p.W1c + ((size_t)bh * K + k) * 64 * 64;
#pragma unroll
                for (int fa = 0; fa < 4; ++fa)
#pragma unroll
                    for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                        for (int r = 0; r < 4; ++r) W1t[fa][fb][r] = W1g[(size_t)(16 * fa + 4 * g + r) * 64 + 16 * fb + i];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) b1v[fb] = p.b1c[((size_t)bh * K + k) * 64 + 16 * fb + i];
            }
            for (int it = lo; it <= hi; ++it) {      // iteration hi only parks the state that ends the group
                const int buf = it & 1;
                const int l = bk.opaque(l0), g = l >> 4, i = l & 15;
                const bool fin = (it == hi), last = (it + 1 == hi);
                bf16x8 WF[2][4];
                {
                    char* slot = fin ? bk.lds_ptr(L_WHI) : scr_w + (size_t)(it - lo) * SLOT_BYTES;
                    bf16x8 WT[2][4];
                    transposed_packs(bk, W1t, WT);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                        for (int f = 0; f < 4; ++f) {
                            WF[ks][f] = stack(W1t[2 * ks][f], W1t[2 * ks + 1][f]);
                            st_pack(bk, slot, ks * 4 + f, WF[ks][f]);
                            st_pack(bk, slot, 8 + ks * 4 + f, WT[ks][f]);
                        }
                }
                if (fin) {
#pragma unroll
                    for (int fb = 0; fb < 4; ++fb) b1hi[fb] = b1v[fb];
                    bk.lds_fence();
                    break;
                }
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) scr_b[(size_t)(it - lo) * 64 + 16 * fb + i] = b1v[fb];       // every lane group: same value
                const int Kt = L_K + buf * TILE * 2, Vt = L_V + buf * TILE * 2;
                if (!last) {        // K, V, eta of the next step
                    stage_request(bk, sk, p.XK + (tile0 + it + 1) * 1024);
                    stage_request(bk, sv, p.XV + (tile0 + it + 1) * 1024);
                    pe = *reinterpret_cast<const unsigned short*>(p.eta + (tile0 + it + 1) * 16 + (l & 15));
                } else {            // Q, dOut of this step: the reverse pass starts here
                    stage_request(bk, sq, p.XQ + (tile0 + it) * 1024);
                    stage_request(bk, sd, p.dOut + (tile0 + it) * 1024);
                }
                // forward step (same arithmetic as forward())
                const bf16x8 kA0 = rho_read(bk, Kt, 0), kA1 = rho_read(bk, Kt, 32);
                const f32x4 eta4 = bk.template lds<f32x4>(L_ETA + (buf * 16 + 4 * g) * 4);
                bf16x4 kT[4];
                f32x4 z[4], tg[4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    kT[fb] = tr4(bk, Kt, TS, 0, 16 * fb);
                    f32x4 a = zero4();
                    a = bk.mma32(kA0, WF[0][fb], a);
                    a = bk.mma32(kA1, WF[1][fb], a);
                    z[fb] = a + b1v[fb];
                    tg[fb] = bk.mma16(IDN, kT[fb], bk.mma16(IDP, tr4(bk, Vt, TS, 0, 16 * fb), zero4()));
                }
                bf16x4 gzp[4];
                {
                    InnerGrad ig;
                    inner_grad(bk, z, tg, gam, bet, p.eps, ig);
#pragma unroll
                    for (int fb = 0; fb < 4; ++fb) gzp[fb] = pack4(ig.gz[fb] * (-eta4));
                }
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    b1v[fb] += bk.mma16(ONES, gzp[fb], zero4())[0];
#pragma unroll
                    for (int fa = 0; fa < 4; ++fa) W1t[fa][fb] = bk.mma16(kT[fa], gzp[fb], W1t[fa][fb]);
                }
                if (!last) {
                    const int nb = (it + 1) & 1;
                    stage_park(bk, sk, L_K + nb * TILE * 2);
                    stage_park(bk, sv, L_V + nb * TILE * 2);
                    if (l < 16) bk.template lds<float>(L_ETA + (nb * 16 + l) * 4) = (float)*reinterpret_cast<const __bf16*>(&pe);
                } else {
                    stage_park(bk, sq, L_Q + buf * TILE * 2);
                    stage_park(bk, sd, L_D + buf * TILE * 2);
                }
                bk.lds_fence();
            }
        }

        // ================= reverse pass over the group ======================================================================
        for (int it = hi - 1; it >= lo; --it) {
            const size_t tile = tile0 + it;
            const int buf = it & 1;
            const int l = bk.opaque(l0), g = l >> 4, i = l & 15;
            const int Kt = L_K + buf * TILE * 2, Vt = L_V + buf * TILE * 2, Qt = L_Q + buf * TILE * 2, Dt = L_D + buf * TILE * 2;
            const int nxt = (it > lo) ? it - 1 : lo - G;          // step whose tiles are requested now (< 0: nothing left)
            if (nxt >= 0) {
                stage_request(bk, sk, p.XK + (tile0 + nxt) * 1024);
                stage_request(bk, sv, p.XV + (tile0 + nxt) * 1024);
                pe = *reinterpret_cast<const unsigned short*>(p.eta + (tile0 + nxt) * 16 + (l & 15));
                if (it > lo) {
                    stage_request(bk, sq, p.XQ + (tile0 + nxt) * 1024);
                    stage_request(bk, sd, p.dOut + (tile0 + nxt) * 1024);
                }
            }
            const char* slot = scr_w + (size_t)(it - lo) * SLOT_BYTES;                                   // state entering / after the step
            const char* slot_n = (it + 1 < hi) ? slot + SLOT_BYTES : bk.lds_ptr(L_WHI);
            const f32x4 eta4 = bk.template lds<f32x4>(L_ETA + (buf * 16 + 4 * g) * 4);
            float b1v[4], b1n[4];
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                b1v[fb] = scr_b[(size_t)(it - lo) * 64 + 16 * fb + i];
                b1n[fb] = (it + 1 < hi) ? scr_b[(size_t)(it + 1 - lo) * 64 + 16 * fb + i] : b1hi[fb];
            }

            // ---- (2) outer LayerNorm backward: Z1b = Q W1n + b1n ; dZ1b -----------------------------------------------------------
            bf16x4 qT[4], dZbp[4];
            f32x4 dq[4];                     // starts as dOut (accumulator layout), becomes dQ
            {
                const bf16x8 qA0 = rho_read(bk, Qt, 0), qA1 = rho_read(bk, Qt, 32);
                f32x4 y[4], dxl[4], t2[4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    qT[fb] = tr4(bk, Qt, TS, 0, 16 * fb);
                    f32x4 a = zero4();
                    a = bk.mma32(qA0, ld_pack(bk, slot_n, fb), a);
                    a = bk.mma32(qA1, ld_pack(bk, slot_n, 4 + fb), a);
                    y[fb] = a + b1n[fb];
                    dq[fb] = bk.mma16(IDP, tr4(bk, Dt, TS, 0, 16 * fb), zero4());                     // exact dOut
                }
                const f32x4 rstdl = normalize_rows(bk, y, p.eps);                                     // y <- x_hat of the output LN
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    const f32x4 dx = dq[fb] * y[fb];
                    dgam[fb] += dx[0] + dx[1] + dx[2] + dx[3];
                    dbet[fb] += dq[fb][0] + dq[fb][1] + dq[fb][2] + dq[fb][3];
                    dxl[fb] = dq[fb] * gam[fb];
                    t2[fb] = dxl[fb] * y[fb];
                }
                const f32x4 u1 = rowsum64(bk, dxl), u2 = rowsum64(bk, t2);
                const f32x4 sc = rstdl * (1.0f / 64.0f);
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    const f32x4 dzb = (64.0f * dxl[fb] - u1 - y[fb] * u2) * sc;
                    dZbp[fb] = pack4(dzb);
                    db[fb] += colsum16(bk, dzb);                                                        // db1n += colsum dZ1b (fp32)
                }
            }
            // ---- (3) dW1n += Q^T dZ1b ; db1n += colsum dZ1b ------------------------------------------------------------------------------
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
                for (int fa = 0; fa < 4; ++fa) dWt[fa][fb] = bk.mma16(qT[fa], dZbp[fb], dWt[fa][fb]);
            }
            // ---- (4) dQ = dOut + dZ1b W1n^T ------------------------------------------------------------------------------------------------
            {
                bf16x8 aZ[2];
                image_of(bk, L_IMG, dZbp, aZ);
#pragma unroll
                for (int fa = 0; fa < 4; ++fa) {
                    dq[fa] = bk.mma32(aZ[0], ld_pack(bk, slot_n, 8 + fa), dq[fa]);
                    dq[fa] = bk.mma32(aZ[1], ld_pack(bk, slot_n, 12 + fa), dq[fa]);
                }
                store_rows(bk, L_IMG + IMG_BYTES, dq, p.dXQ + tile * 1024);
            }
            // The tiles requested at the top of the step are parked as early as their buffers allow: right here when they
            // belong to the other parity (idle during this step: the usual case), at the end of the step when the next group's
            // first step shares this step's parity (even group sizes).
            const bool park_early = nxt >= 0 && (nxt & 1) != buf;
            const bool park_late = nxt >= 0 && !park_early;
            if (park_early) {
                const int nb = nxt & 1;
                stage_park(bk, sk, L_K + nb * TILE * 2);
                stage_park(bk, sv, L_V + nb * TILE * 2);
                if (l < 16) bk.template lds<float>(L_ETA + (nb * 16 + l) * 4) = (float)*reinterpret_cast<const __bf16*>(&pe);
                if (it > lo) {
                    stage_park(bk, sq, L_Q + nb * TILE * 2);
                    stage_park(bk, sd, L_D + nb * TILE * 2);
                }
            }
            bk.lds_fence();
            // ---- (1) inner forward of the step: Z1 = K W + b, LN / L2 gradient ------------------------------------------------
            const bf16x8 kA0 = rho_read(bk, Kt, 0), kA1 = rho_read(bk, Kt, 32);
            bf16x4 kT[4];
            InnerGrad ig;
            {
                f32x4 z[4], tg[4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    kT[fb] = tr4(bk, Kt, TS, 0, 16 * fb);
                    f32x4 a = zero4();
                    a = bk.mma32(kA0, ld_pack(bk, slot, fb), a);
                    a = bk.mma32(kA1, ld_pack(bk, slot, 4 + fb), a);
                    z[fb] = a + b1v[fb];
                    tg[fb] = bk.mma16(IDN, kT[fb], bk.mma16(IDP, tr4(bk, Vt, TS, 0, 16 * fb), zero4()));   // exact V - K
                }
                inner_grad(bk, z, tg, gam, bet, p.eps, ig);
            }
            bk.lds_fence();
            // ---- (6) dgZ1 = -eta (K dW1n + db1n) ; (8) backward of the fused LN / L2 gradient -> dZ1, dt, dgamma, dbeta -----------------
            bf16x4 dZ1p[4];
            float dbz[4];                        // colsum dZ1 (fp32), added to db1 in (10) - d(eta) in (7) needs db1n
            f32x4 dk[4];                     // starts as -dt (dt = gradient w.r.t. the target V - K = dV)
            {
                f32x4 dgz[4], mGr[4], t2[4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    f32x4 a = zero4();
                    a = bk.mma32(kA0, stack(dWt[0][fb], dWt[1][fb]), a);
                    a = bk.mma32(kA1, stack(dWt[2][fb], dWt[3][fb]), a);
                    dgz[fb] = (a + db[fb]) * (-eta4);
                    mGr[fb] = dgz[fb] * (-ig.rstd);
                    t2[fb] = mGr[fb] * ig.xh[fb];
                }
                const f32x4 s1 = rowsum64(bk, mGr) * (1.0f / 64.0f), s2 = rowsum64(bk, t2) * (1.0f / 64.0f);
                const f32x4 c2 = ig.s2g * (1.0f / 64.0f);
                f32x4 dxh[4], dstd[4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    const f32x4 dgxh = dgz[fb] * ig.rstd + s1 + ig.xh[fb] * s2;
                    const f32x4 dy = dgxh * gam[fb];
                    const f32x4 dg = ig.go[fb] * dgxh + dy * ig.xh[fb];
                    dgam[fb] += dg[0] + dg[1] + dg[2] + dg[3];
                    dbet[fb] += dy[0] + dy[1] + dy[2] + dy[3];
                    dk[fb] = dy;                                                                       // = -dt
                    dxh[fb] = dy * gam[fb] + (ig.go[fb] * gam[fb]) * s2 + mGr[fb] * c2;
                    dstd[fb] = (dxh[fb] * ig.xh[fb] + dgz[fb] * ig.gz[fb]) * (-ig.rstd);
                }
                const f32x4 v1 = rowsum64(bk, dxh) * (1.0f / 64.0f), v2 = rowsum64(bk, dstd) * (1.0f / 64.0f);
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) {
                    const f32x4 dz1 = (dxh[fb] - v1) * ig.rstd + ig.xh[fb] * v2;
                    dZ1p[fb] = pack4(dz1);
                    dbz[fb] = colsum16(bk, dz1);
                }
                f32x4 dv[4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) dv[fb] = -dk[fb];
                store_rows(bk, L_IMG + IMG_BYTES, dv, p.dXV + tile * 1024);                              // dV = dt
            }
            bk.lds_fence();
            // ---- (5, 7, 9) A1 = gZ1 dW1n^T ; d eta ; dK = -eta A1 - dt + dZ1 W^T --------------------------------------------------------------
            {
                bf16x8 dWT[2][4];
                transposed_packs(bk, dWt, dWT);
                bf16x4 gzq[4];
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) gzq[fb] = pack4(ig.gz[fb]);
                bf16x8 aG[2];
                image_of(bk, L_IMG, gzq, aG);
                f32x4 acc[4];
#pragma unroll
                for (int fa = 0; fa < 4; ++fa) {
                    f32x4 a1 = zero4();
                    a1 = bk.mma32(aG[0], dWT[0][fa], a1);
                    a1 = bk.mma32(aG[1], dWT[1][fa], a1);
                    dk[fa] -= a1 * eta4;
                    const f32x4 kc = bk.mma16(IDP, kT[fa], zero4());                                  // exact K, accumulator layout
                    acc[fa] = kc * a1 + ig.gz[fa] * db[fa];
                }
                const f32x4 de = rowsum64(bk, acc);
                if (i == 0) *reinterpret_cast<bf16x4*>(p.deta + tile * 16 + 4 * g) = pack4(-de);
            }
            bk.lds_fence();
            {
                bf16x8 aD[2];
                image_of(bk, L_IMG, dZ1p, aD);
#pragma unroll
                for (int fa = 0; fa < 4; ++fa) {
                    dk[fa] = bk.mma32(aD[0], ld_pack(bk, slot, 8 + fa), dk[fa]);
                    dk[fa] = bk.mma32(aD[1], ld_pack(bk, slot, 12 + fa), dk[fa]);
                }
                store_rows(bk, L_IMG + IMG_BYTES, dk, p.dXK + tile * 1024);
            }
            // ---- (10) dW1 = dW1n + K^T dZ1 ; db1 = db1n + colsum dZ1 --------------------------------------------------------------------------------
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                db[fb] += dbz[fb];
#pragma unroll
                for (int fa = 0; fa < 4; ++fa) dWt[fa][fb] = bk.mma16(kT[fa], dZ1p[fb], dWt[fa][fb]);
            }
            if (park_late) {                 // only K, V, eta of the next group's first step (it == lo)
                const int nb = nxt & 1;
                stage_park(bk, sk, L_K + nb * TILE * 2);
                stage_park(bk, sv, L_V + nb * TILE * 2);
                if (l < 16) bk.template lds<float>(L_ETA + (nb * 16 + l) * 4) = (float)*reinterpret_cast<const __bf16*>(&pe);
            }
            bk.lds_fence();
        }
    }
    // ---- results ---------------------------------------------------------------------------------------------------------
    {
        const int g = l0 >> 4, i = l0 & 15;
        float* dWg = p.dW1 + (size_t)bh * 64 * 64;
#pragma unroll
        for (int fa = 0; fa < 4; ++fa)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                for (int r = 0; r < 4; ++r) dWg[(size_t)(16 * fa + 4 * g + r) * 64 + 16 * fb + i] = dWt[fa][fb][r];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            const float dg = bk.xor_add(bk.xor_add(dgam[fb], 16), 32), dbt = bk.xor_add(bk.xor_add(dbet[fb], 16), 32);
            if (g == 0) {
                p.db1[(size_t)bh * 64 + 16 * fb + i] = db[fb];
                p.dln_w[(size_t)bh * 64 + 16 * fb + i] = dg;
                p.dln_b[(size_t)bh * 64 + 16 * fb + i] = dbt;
            }
        }
    }
}

}  // namespace lin16
}  // namespace ttt
