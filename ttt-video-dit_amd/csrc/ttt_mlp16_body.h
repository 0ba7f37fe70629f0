// TTT-MLP forward scan at mini-batches of 16 tokens as a WORKGROUP body over the wave backend (see ttt_lin16_body.h for the
// backend idea): 8 waves per (b, h), two barriers per step.  It restates the schedule of mlp_scan16_kernel (ttt_mfma16.hip,
// where the step structure and the hazard analysis are documented) against the backend primitives plus
//   bk.wave()  wave index in the workgroup,  bk.thread() = 64 wave + lane,  bk.barrier()  workgroup barrier,
//   bk.exp2(x), bk.rcp(x),  bk.lds_load<T>(off) / bk.lds_store<T>(off, v)  (plain LDS accesses on the device; on the emulator
//   they feed its LDS race detector: two waves touching a word without a workgroup barrier in between is reported)
// so that the CPU suite can execute it on the multi-wave emulator (tests/emul).  STATUS: emulator-verified against the
// oracle; on the device it is the opt-in variant `scan16_body` of ttt_hip_debug_option until it has been timed against the
// hand-placed kernel on an MI355X.
#pragma once
#include "ttt_lin16_body.h"

namespace ttt {
namespace mlp16 {
using namespace ttt::wv;
using ttt::lin16::cat;
using ttt::lin16::pack4;
using ttt::lin16::stack;
using ttt::lin16::tr4;
using ttt::lin16::zero4;
using ttt::lin16::TS;

constexpr int CS = 16, TILE = CS * TS, IS = 24, PS = 68;
constexpr int L_K = 0, L_V = L_K + 2 * TILE * 2, L_Q = L_V + 2 * TILE * 2, L_G = L_Q + 3 * TILE * 2, L_IMG = L_G + TILE * 2;
constexpr int IMG_BYTES = 32 * IS * 2;
constexpr int L_REDA = L_IMG + 8 * IMG_BYTES, RED_BYTES = 8 * CS * PS * 4, L_REDB = L_REDA + RED_BYTES;
constexpr int L_ETA = L_REDB + RED_BYTES, L_B2 = L_ETA + 32 * 4, L_GAM = L_B2 + 64 * 4, L_BET = L_GAM + 64 * 4;
constexpr int GROUP_LDS = L_BET + 64 * 4;
static_assert(GROUP_LDS <= 160 * 1024 && L_IMG % 16 == 0 && L_REDA % 16 == 0 && L_ETA % 16 == 0, "LDS map");

// rho-order operand from this lane's row of a row-major [16][TS] tile (tracked LDS loads: the race detector sees them)
template <class BK>
TTT_WV_FN bf16x8 rho_read(BK& bk, int tile_off, int c0) {
    const int g = bk.lane() >> 4, i = bk.lane() & 15;
    const int o = tile_off + (i * TS + c0 + 4 * g) * 2;
    return cat(bk.template lds_load<bf16x4>(o), bk.template lds_load<bf16x4>(o + 32));
}

constexpr float GELU_A = 0.79788456f, GELU_C = 0.044715f, GELU_3AC = 0.1070322243f;
constexpr float GELU_K0 = -2.0f * GELU_A * 1.4426950408889634f, GELU_K1 = GELU_K0 * GELU_C;
// tanh-GELU and its derivative (reference ops/utils.py:47-54), sigmoid form: gelu(x) = x s, s = 1 / (1 + 2^(x (k0 + k1 x^2)))
template <class BK>
TTT_WV_FN void gelu_fwd_grad(BK& bk, float x, float& y, float& dy) {
    const float x2 = x * x;
    const float s = bk.rcp(1.0f + bk.exp2(x * (x2 * GELU_K1 + GELU_K0)));
    y = x * s;
    dy = (y - y * s) * (x2 * (2.0f * GELU_3AC) + 2.0f * GELU_A) + s;
}
template <class BK>
TTT_WV_FN float gelu_fwd(BK& bk, float x) { return x * bk.rcp(1.0f + bk.exp2(x * (x * x * GELU_K1 + GELU_K0))); }

// owner thread (token ot, features of0 .. of0 + 3): bias + sum of the 8 waves' partials, then LayerNorm statistics
template <class BK>
TTT_WV_FN f32x4 gather8(BK& bk, int red_off, int ot, int of0) {
    f32x4 z = bk.template lds_load<f32x4>(L_B2 + of0 * 4);
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) z += bk.template lds_load<f32x4>(red_off + ((ww * CS + ot) * PS + of0) * 4);
    return z;
}
template <class BK>
TTT_WV_FN void row_stats(BK& bk, f32x4 z, float eps, float& mu, float& rstd) {
    mu = bk.sum16(z[0] + z[1] + z[2] + z[3]) * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = z[j] - mu; v += d * d; }
    rstd = bk.rsq(bk.sum16(v) * (1.0f / 64.0f) + eps);
}

template <class BK>
TTT_WV_FN void forward(BK& bk, const Mlp16Params& p, int bh) {
    const int tid0 = bk.thread(), wv = bk.wave(), l0 = bk.lane();
    const int n0 = 32 * wv, img = L_IMG + wv * IMG_BYTES;
    const int NC = p.NC, G = p.G, head = bh % p.NH;

    f32x4 W1t[4][2], W2t[2][4], W2Tt[4][2];      // (rows = f, lane = n) ; (rows = n, lane = f) ; (rows = f, lane = n)
    float b1v[2], b2v[4];
    {
        const int g = l0 >> 4, i = l0 & 15;
        const float* W1g = p.W1 + (size_t)bh * 64 * 256;
        const float* W2g = p.W2 + (size_t)bh * 256 * 64;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    W1t[fb][nb][r] = W1g[(size_t)(16 * fb + 4 * g + r) * 256 + n0 + 16 * nb + i];
                    W2t[nb][fb][r] = W2g[(size_t)(n0 + 16 * nb + 4 * g + r) * 64 + 16 * fb + i];
                }
                W2Tt[fb][nb] = *reinterpret_cast<const f32x4*>(W2g + (size_t)(n0 + 16 * nb + i) * 64 + 16 * fb + 4 * g);
            }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) b1v[nb] = p.b1[(size_t)bh * 256 + n0 + 16 * nb + i];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) b2v[fb] = p.b2[(size_t)bh * 64 + 16 * fb + i];
        if (tid0 < 64) {
            bk.template lds_store<float>(L_B2 + tid0 * 4, p.b2[(size_t)bh * 64 + tid0]);
            bk.template lds_store<float>(L_GAM + tid0 * 4, p.ln_w[(size_t)head * 64 + tid0]);
            bk.template lds_store<float>(L_BET + tid0 * 4, p.ln_b[(size_t)head * 64 + tid0]);
        }
    }
    const bf16x4 ONES = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
    bf16x8 W1F[2][2], W2F[4];                    // packed operands of the current state, carried across steps
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) W1F[ks][nb] = stack(W1t[2 * ks][nb], W1t[2 * ks + 1][nb]);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) W2F[fb] = stack(W2t[0][fb], W2t[1][fb]);

    // input staging: thread groups of 128 move one 16-byte chunk of K / V / Q (waves 6, 7 mirror Q's loads and do not store)
    const size_t tile0 = (size_t)bh * NC;
    const int which = wv >> 1;
    const __bf16* src = which == 0 ? p.XK : which == 1 ? p.XV : p.XQ;
    const int dstb = which == 0 ? L_K : which == 1 ? L_V : L_Q;
    const int nbufs = which == 2 ? 3 : 2;
    const int lt0 = tid0 & 127, lofs = ((lt0 >> 3) * TS + (lt0 & 7) * 8) * 2;
    const size_t gofs = (size_t)(lt0 >> 3) * 64 + (lt0 & 7) * 8;
    u32x4 pfO;
    unsigned short pfEO;
    {
        const u32x4 t0 = *reinterpret_cast<const u32x4*>(src + tile0 * 1024 + gofs);
        if (which < 3) bk.template lds_store<u32x4>(dstb + lofs, t0);
        if (tid0 < 16) bk.template lds_store<float>(L_ETA + tid0 * 4, (float)p.eta[tile0 * 16 + tid0]);
        const size_t t1 = tile0 + (NC > 1 ? 1 : 0);
        pfO = *reinterpret_cast<const u32x4*>(src + t1 * 1024 + gofs);
        pfEO = *reinterpret_cast<const unsigned short*>(p.eta + t1 * 16 + (tid0 & 15));
    }
    bk.barrier();

    for (int it = 0; it <= NC; ++it) {      // iteration NC only drains the last P6
        const size_t tile = tile0 + it;
        const bool live = it < NC;
        const int buf = it & 1;
        const int l = bk.opaque(l0), g = l >> 4, i = l & 15;
        const int tid = bk.opaque(tid0);
        const int ot = (tid & 255) >> 4, of0 = 4 * (tid & 15);
        const int Kt = L_K + buf * TILE * 2, Vt = L_V + buf * TILE * 2, Qt = L_Q + (it % 3) * TILE * 2;

        const size_t tn = tile0 + (it + 2 < NC ? it + 2 : NC - 1);
        const u32x4 pfN = *reinterpret_cast<const u32x4*>(src + tn * 1024 + gofs);
        const unsigned short pfEN = *reinterpret_cast<const unsigned short*>(p.eta + tn * 16 + (tid & 15));

        f32x4 D1[2];
        bf16x4 X2p[2];
        if (live) {
            if (it % G == 0) {
                const size_t ck = (size_t)bh * p.K + it / G;
                float* W1g = p.W1c + ck * 64 * 256;
                float* W2g = p.W2c + ck * 256 * 64;
#pragma unroll
                for (int fb = 0; fb < 4; ++fb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) W1g[(size_t)(16 * fb + 4 * g + r) * 256 + n0 + 16 * nb + i] = W1t[fb][nb][r];
                        *reinterpret_cast<f32x4*>(W2g + (size_t)(n0 + 16 * nb + i) * 64 + 16 * fb + 4 * g) = W2Tt[fb][nb];
                    }
                if (g == 0) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) p.b1c[ck * 256 + n0 + 16 * nb + i] = b1v[nb];
                    if (wv == 0) {
#pragma unroll
                        for (int fb = 0; fb < 4; ++fb) p.b2c[ck * 64 + 16 * fb + i] = b2v[fb];
                    }
                }
            }
            // A1: Z1 = K W1 + b1 ; X2, D1 (rows = t, lane = n) ; X2 image [n][t]
            const bf16x8 kA0 = rho_read(bk, Kt, 0), kA1 = rho_read(bk, Kt, 32);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 Z = zero4();
                Z = bk.mma32(kA0, W1F[0][nb], Z);
                Z = bk.mma32(kA1, W1F[1][nb], Z);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y, dy;
                    gelu_fwd_grad(bk, Z[r] + b1v[nb], y, dy);
                    Z[r] = y;
                    D1[nb][r] = dy;
                }
                X2p[nb] = pack4(Z);
                bk.template lds_store<bf16x4>(img + ((16 * nb + i) * IS + 4 * g) * 2, X2p[nb]);
            }
            bk.lds_fence();
            // A2: partial Z2^T[f, t] over this wave's hidden slice
            const bf16x8 xB = cat(tr4(bk, img, IS, 0, 0), tr4(bk, img, IS, 16, 0));           // lane = t, k = n (rho)
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
                bk.template lds_store<f32x4>(L_REDA + ((wv * CS + i) * PS + 4 * g + 16 * fb) * 4, bk.mma32(W2F[fb], xB, zero4()));
        }
        bk.barrier();                 // B1

        if (wv < 4) {
            if (live) {               // P3: owners - reduce, fused LN / L2 backward -> Gs = -eta gZ2
                f32x4 z = gather8(bk, L_REDA, ot, of0);
                float mu, rstd;
                row_stats(bk, z, p.eps, mu, rstd);
                const bf16x4 kk = bk.template lds_load<bf16x4>(Kt + (ot * TS + of0) * 2);
                const bf16x4 vv = bk.template lds_load<bf16x4>(Vt + (ot * TS + of0) * 2);
                const f32x4 gm = bk.template lds_load<f32x4>(L_GAM + of0 * 4), bt = bk.template lds_load<f32x4>(L_BET + of0 * 4);
                float s1 = 0.f, s2 = 0.f, gx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (z[j] - mu) * rstd;
                    gx[j] = (gm[j] * xh + bt[j] - ((float)vv[j] - (float)kk[j])) * gm[j];
                    z[j] = xh;
                    s1 += gx[j]; s2 += gx[j] * xh;
                }
                s1 = bk.sum16(s1);
                s2 = bk.sum16(s2);
                const float sc = -bk.template lds_load<float>(L_ETA + (buf * 16 + ot) * 4) * rstd * (1.0f / 64.0f);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)((64.0f * gx[j] - s1 - z[j] * s2) * sc);
                bk.template lds_store<bf16x4>(L_G + (ot * TS + of0) * 2, o);
            }
        } else if (it > 0) {          // P6 (step it-1): owners - reduce, LayerNorm, residual -> XQW
            const f32x4 z = gather8(bk, L_REDB, ot, of0);
            float mu, rstd;
            row_stats(bk, z, p.eps, mu, rstd);
            const bf16x4 q = bk.template lds_load<bf16x4>(L_Q + ((it - 1) % 3) * TILE * 2 + (ot * TS + of0) * 2);
            const f32x4 gm = bk.template lds_load<f32x4>(L_GAM + of0 * 4), bt = bk.template lds_load<f32x4>(L_BET + of0 * 4);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)((float)q[j] + gm[j] * ((z[j] - mu) * rstd) + bt[j]);
            *reinterpret_cast<bf16x4*>(p.out + (tile - 1) * 1024 + (size_t)ot * 64 + of0) = o;
        }
        // park tile it+1 (requested one step ago)
        if (which < 3) bk.template lds_store<u32x4>(dstb + ((it + 1) % nbufs) * TILE * 2 + lofs, pfO);
        if (tid < 16) bk.template lds_store<float>(L_ETA + ((buf ^ 1) * 16 + tid) * 4, (float)*reinterpret_cast<const __bf16*>(&pfEO));
        pfO = pfN;
        pfEO = pfEN;
        if (!live) break;
        bk.barrier();                 // B2

        // C: state updates ; gX2 ; W1 update ; Z1b
        {
            bf16x8 W2TF[2][2];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) W2TF[ks][nb] = stack(W2Tt[2 * ks][nb], W2Tt[2 * ks + 1][nb]);
            const bf16x8 gA0 = rho_read(bk, L_G, 0), gA1 = rho_read(bk, L_G, 32);
            bf16x4 gzp[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 gx = zero4();
                gx = bk.mma32(gA0, W2TF[0][nb], gx);
                gx = bk.mma32(gA1, W2TF[1][nb], gx);
                gx *= D1[nb];
                gzp[nb] = pack4(gx);
                b1v[nb] += bk.mma16(ONES, gzp[nb], zero4())[0];
            }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const bf16x4 kT = tr4(bk, Kt, TS, 0, 16 * fb);
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) W1t[fb][nb] = bk.mma16(kT, gzp[nb], W1t[fb][nb]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) W1F[ks][nb] = stack(W1t[2 * ks][nb], W1t[2 * ks + 1][nb]);
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const bf16x4 gT = tr4(bk, L_G, TS, 0, 16 * fb);
                b2v[fb] += bk.mma16(ONES, gT, zero4())[0];
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    W2t[nb][fb] = bk.mma16(X2p[nb], gT, W2t[nb][fb]);
                    W2Tt[fb][nb] = bk.mma16(gT, X2p[nb], W2Tt[fb][nb]);
                }
                W2F[fb] = stack(W2t[0][fb], W2t[1][fb]);
            }
            const bf16x8 qA0 = rho_read(bk, Qt, 0), qA1 = rho_read(bk, Qt, 32);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 Z = zero4();
                Z = bk.mma32(qA0, W1F[0][nb], Z);
                Z = bk.mma32(qA1, W1F[1][nb], Z);
#pragma unroll
                for (int r = 0; r < 4; ++r) Z[r] = gelu_fwd(bk, Z[r] + b1v[nb]);
                bk.template lds_store<bf16x4>(img + ((16 * nb + i) * IS + 4 * g) * 2, pack4(Z));
            }
            bk.lds_fence();
        }
        // E: partial Z2b^T -> redB ; b2' -> LDS
        {
            const bf16x8 xB = cat(tr4(bk, img, IS, 0, 0), tr4(bk, img, IS, 16, 0));
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
                bk.template lds_store<f32x4>(L_REDB + ((wv * CS + i) * PS + 4 * g + 16 * fb) * 4, bk.mma32(W2F[fb], xB, zero4()));
        }
        if (wv == 0 && g == 0) {
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) bk.template lds_store<float>(L_B2 + (16 * fb + i) * 4, b2v[fb]);
        }
    }
}

}  // namespace mlp16
}  // namespace ttt
