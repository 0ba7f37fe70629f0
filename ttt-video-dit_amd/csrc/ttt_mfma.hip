// MFMA TTT-MLP scan kernels for gfx950: bf16 operands on v_mfma_f32_32x32x16_bf16, fp32 state and
// accumulation.  Geometry: CS = 64, F = 64 (CogVideoX-5B heads), bf16 activations.
//
// One workgroup = 4 waves (one per SIMD, up to 512 VGPRs each) per (batch, head); the scan over
// NC mini-batches is sequential (reference grid (B,NH), linear_triton.py:96).  Wave w owns the
// hidden slice H_w = [64w, 64w+64) of the TTT-MLP:  W1[:, H_w], b1[H_w], W2[H_w, :] live as fp32
// MFMA accumulator tiles for the whole scan, so the inner-loop SGD update  W -= (eta X)^T g  is
// an MFMA that accumulates straight into the state.  See ttt_mfma_dev.h for the layout algebra
// (in-place operand reuse of C tiles, pi reads, MFMA transposes).
//
// Per step (primal form) - 7 algorithmic GEMMs + tile transposes:
//   P1  Z1 = K W1 + b1 (rows=t, lane=n) ; X2 = gelu, D1 = gelu'
//   P2  X2^T (MFMA transpose) ; partial Z2^T_w = W2[H_w,:]^T X2[:,H_w]^T  -> LDS (fp32)
//   P3  owners (16 tokens per wave): sum the 4 partials + b2, fused LN/L2 backward -> gZ2 -> LDS
//   P4  gX2 = gZ2 W2^T ; gZ1 = gX2*D1 ; W1 -= (eta K)^T gZ1 ; W2 -= (eta X2)^T gZ2 ; b1, b2
//   P5  Z1b^T = W1'^T Q^T + b1' (rows=n, lane=t) ; X2b = gelu ; partial Z2b^T -> LDS
//   P6  owners: sum partials + b2', LayerNorm, + Q  -> XQW (bf16)
// Q/K/V tiles of step i+1 are fetched into registers during step i and parked in the other LDS
// buffer before the step's last barrier.
//
// The same body, instantiated with SAVE=true, is the backward's group-recompute kernel: one
// workgroup per (batch, head, checkpoint group) - K-fold more parallelism than the scan itself, so
// it runs on the CUs the 48-workgroup scan leaves idle - re-runs the group's forward from its
// checkpoint and stores every intermediate the reverse sweep needs (ttt_mfma_bwd.hip) as register
// images into per-step workspace slots.
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"

namespace ttt {
namespace mfma {
using namespace ttt::mf;

constexpr int LDS_TILES = 2 * 3 * TILE_ELEMS * 2;            // bytes: 2 buffers x (K,Q,V)
constexpr int LDS_RED = 4 * 64 * PS * 4;                     // bytes: 4 waves x [64][PS] fp32
constexpr int LDS_G1 = TILE_ELEMS * 2;
constexpr int LDS_SMALL = (2 * 64 + 4 * 64 + 3 * 64) * 4;    // eta[2][64], b1s[4][64], b2, gam, bet
constexpr int LDS_FWD = LDS_TILES + LDS_RED + LDS_G1 + LDS_SMALL;

static unsigned long long* g_dbg = nullptr;
void set_debug_timing(void* buf) { g_dbg = (unsigned long long*)buf; }
unsigned long long* get_debug_timing() { return g_dbg; }

#define TTT_STAMP(k)                                                         \
    if (p.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {           \
        const unsigned long long _t = __builtin_readcyclecounter();          \
        p.dbg[k] += _t - t_last;                                             \
        t_last = _t;                                                         \
    }

struct Prefetch {
    uint4 v[6];
    float eta;
};

__device__ __forceinline__ void prefetch_issue(Prefetch& pf, const ScanParams& p, size_t tile) {
    const size_t base = tile * 4096;   // 64*64 elements per tile
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = threadIdx.x + NT * j;              // 16-byte chunk id, 8 per 128-B row
        const size_t off = base + (size_t)(q >> 3) * 64 + (q & 7) * 8;
        pf.v[0 + j] = *reinterpret_cast<const uint4*>(p.XK + off);
        pf.v[2 + j] = *reinterpret_cast<const uint4*>(p.XQ + off);
        pf.v[4 + j] = *reinterpret_cast<const uint4*>(p.XV + off);
    }
    pf.eta = (threadIdx.x < 64) ? (float)p.eta[tile * 64 + threadIdx.x] : 0.f;
}
__device__ __forceinline__ void prefetch_park(const Prefetch& pf, __bf16* tiles, float* etaL) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = threadIdx.x + NT * j;
        const int o = (q >> 3) * TS + (q & 7) * 8;
        *reinterpret_cast<uint4*>(tiles + 0 * TILE_ELEMS + o) = pf.v[0 + j];
        *reinterpret_cast<uint4*>(tiles + 1 * TILE_ELEMS + o) = pf.v[2 + j];
        *reinterpret_cast<uint4*>(tiles + 2 * TILE_ELEMS + o) = pf.v[4 + j];
    }
    if (threadIdx.x < 64) etaL[threadIdx.x] = pf.eta;
}

template <bool SAVE>
__global__ __launch_bounds__(NT, 1) void mlp_scan_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* tiles = reinterpret_cast<__bf16*>(smem);
    float* red = reinterpret_cast<float*>(smem + LDS_TILES);
    __bf16* G1 = reinterpret_cast<__bf16*>(smem + LDS_TILES + LDS_RED);
    float* etaL = reinterpret_cast<float*>(smem + LDS_TILES + LDS_RED + LDS_G1);
    float* b1s = etaL + 2 * 64;
    float* b2L = b1s + 4 * 64;
    float* gamL = b2L + 64;
    float* betL = gamL + 64;

    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, h = l >> 5, c = l & 31;
    const int NC = p.NC, G = p.G;
    // forward: one workgroup per (b,h), all steps.  SAVE: one workgroup per (b,h,group of the chunk).
    const int bh = SAVE ? blockIdx.x / p.chunk_groups : blockIdx.x;
    const int grp = SAVE ? p.chunk_group0 + blockIdx.x % p.chunk_groups : 0;
    const int i_lo = SAVE ? grp * G : 0;
    const int i_hi = SAVE ? min(i_lo + G, NC) : NC;
    const int head = bh % p.NH;
    char* slots = SAVE ? p.slots + (size_t)bh * p.slot_stride_bh : nullptr;   // slot s <-> step chunk_lo + s

    // ---- state: W1[:, H_w] as tiles (rows=f, lane=n), W2[H_w, :] as tiles (rows=n, lane=f) ----------
    f32x16 W1t[2][2], W2t[2][2];
    float b1v[2];
    {
        const size_t sb = SAVE ? (size_t)bh * p.K + grp : (size_t)bh;     // SAVE starts from checkpoint `grp`
        const float* W1g = (SAVE ? p.W1c : p.W1) + sb * 64 * 256;
        const float* W2g = (SAVE ? p.W2c : p.W2) + sb * 256 * 64;
        const float* b1g = (SAVE ? p.b1c : p.b1) + sb * 256;
        const float* b2g = (SAVE ? p.b2c : p.b2) + sb * 64;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    W1t[a][b][r] = W1g[(size_t)(32 * a + row_of(r, h)) * 256 + 64 * w + 32 * b + c];
                    W2t[a][b][r] = W2g[(size_t)(64 * w + 32 * a + row_of(r, h)) * 64 + 32 * b + c];
                }
        b1v[0] = b1g[64 * w + c];
        b1v[1] = b1g[64 * w + 32 + c];
        if (threadIdx.x < 64) {
            b2L[threadIdx.x] = b2g[threadIdx.x];
            gamL[threadIdx.x] = p.ln_w[(size_t)head * 64 + threadIdx.x];
            betL[threadIdx.x] = p.ln_b[(size_t)head * 64 + threadIdx.x];
        }
    }
    const bf16x8 I0 = ident_pi(0, h, c), I1 = ident_pi(1, h, c);

    Prefetch pf;
    prefetch_issue(pf, p, (size_t)bh * NC + i_lo);
    prefetch_park(pf, tiles, etaL);
    __syncthreads();

    // owner-lane geometry (P3 / P6)
    const int ot = 16 * w + (l & 15), of0 = 16 * (l >> 4);

    unsigned long long t_last = __builtin_readcyclecounter();
    for (int i = i_lo; i < i_hi; ++i) {
        TTT_STAMP(7)
        const int cur = (i - i_lo) & 1;
        const __bf16* Kt = tiles + (cur * 3 + 0) * TILE_ELEMS;
        const __bf16* Qt = tiles + (cur * 3 + 1) * TILE_ELEMS;
        const __bf16* Vt = tiles + (cur * 3 + 2) * TILE_ELEMS;
        const float* etaC = etaL + cur * 64;
        const size_t tile = (size_t)bh * NC + i;
        char* slot = SAVE ? slots + (size_t)(i - p.chunk_lo) * SLOT_BYTES : nullptr;
        char* slot_w = SAVE ? slot + (size_t)w * SLOT_WAVE_FR : nullptr;
        char* own = SAVE ? slot + SLOT_FR : nullptr;

        if (!SAVE && i % G == 0) {   // checkpoint: state entering step i (mlp_tk.py:95-98)
            const size_t ck = (size_t)bh * p.K + i / G;
            float* W1g = p.W1c + ck * 64 * 256;
            float* W2g = p.W2c + ck * 256 * 64;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        W1g[(size_t)(32 * a + row_of(r, h)) * 256 + 64 * w + 32 * b + c] = W1t[a][b][r];
                        W2g[(size_t)(64 * w + 32 * a + row_of(r, h)) * 64 + 32 * b + c] = W2t[a][b][r];
                    }
            if (h == 0) {
                p.b1c[ck * 256 + 64 * w + c] = b1v[0];
                p.b1c[ck * 256 + 64 * w + 32 + c] = b1v[1];
            }
            if (threadIdx.x < 64) p.b2c[ck * 64 + threadIdx.x] = b2L[threadIdx.x];
        }
        const bool more = (i + 1 < i_hi);
        if (more) prefetch_issue(pf, p, tile + 1);

        // ================= P1: Z1 = K W1 + b1 ; X2, D1 =========================================
        bf16x8 X2F[2][2][2];          // [ti][nj][s]  X2 tile (rows=t, lane=n) packed
        f32x16 D1[2][2];              // gelu'(Z1), same layout
        {
            bf16x8 Kpi[2][2][2];      // [ti][fi][s]
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int s = 0; s < 2; ++s) Kpi[ti][fi][s] = pi_read(Kt + (32 * ti + c) * TS, 32 * fi, s, h);
            f32x16 Z[2][2];
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) Z[ti][nj] = zero16();
#pragma unroll
            for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 w0 = pack(W1t[fi][0], s), w1 = pack(W1t[fi][1], s);
                    if (SAVE) { st_frag(slot_w, FR_W1, fr_idx(fi, 0, s), w0, l); st_frag(slot_w, FR_W1, fr_idx(fi, 1, s), w1, l); }
#pragma unroll
                    for (int ti = 0; ti < 2; ++ti) {
                        Z[ti][0] = mma(Kpi[ti][fi][s], w0, Z[ti][0]);
                        Z[ti][1] = mma(Kpi[ti][fi][s], w1, Z[ti][1]);
                    }
                }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int nj = 0; nj < 2; ++nj) {
                    f32x16 d2;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y, dy, d2y = 0.f;
                        if (SAVE) gelu_fwd_grad2(Z[ti][nj][r] + b1v[nj], y, dy, d2y);
                        else gelu_fwd_grad(Z[ti][nj][r] + b1v[nj], y, dy);
                        Z[ti][nj][r] = y;
                        D1[ti][nj][r] = dy;
                        d2[r] = d2y;
                    }
                    X2F[ti][nj][0] = pack(Z[ti][nj], 0);
                    X2F[ti][nj][1] = pack(Z[ti][nj], 1);
                    if (SAVE) {
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            st_frag(slot_w, FR_X2, fr_idx(ti, nj, s), X2F[ti][nj][s], l);
                            st_frag(slot_w, FR_D1, fr_idx(ti, nj, s), pack(D1[ti][nj], s), l);
                            st_frag(slot_w, FR_D2, fr_idx(ti, nj, s), pack(d2, s), l);
                        }
                        {                     // gelu'(Z1) also in (rows=n, lane=t) orientation
                            const f32x16 dn = transpose_tile(pack(D1[ti][nj], 0), pack(D1[ti][nj], 1), I0, I1);
                            st_frag(slot_w, FR_D1N, fr_idx(nj, ti, 0), pack(dn, 0), l);
                            st_frag(slot_w, FR_D1N, fr_idx(nj, ti, 1), pack(dn, 1), l);
                        }
                    }
                }
        }

        TTT_STAMP(0)
        // ================= P2: X2^T, partial Z2^T, W2^T ============================================
        bf16x8 WTF[2][2][2];          // [fj][ni][s]  W2^T tile (rows=f, lane=n) packed
        {
            bf16x8 W2F[2][2][2];      // [ni][fj][s]
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        W2F[ni][fj][s] = pack(W2t[ni][fj], s);
                        if (SAVE) st_frag(slot_w, FR_W2, fr_idx(ni, fj, s), W2F[ni][fj][s], l);
                    }
            f32x16 P[2][2];           // [fj][ti]
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) P[a][b] = zero16();
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    const f32x16 xt = transpose_tile(X2F[ti][ni][0], X2F[ti][ni][1], I0, I1);   // (rows=n, lane=t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 xb = pack(xt, s);
                        if (SAVE) st_frag(slot_w, FR_XT, fr_idx(ni, ti, s), xb, l);
                        P[0][ti] = mma(W2F[ni][0][s], xb, P[0][ti]);
                        P[1][ti] = mma(W2F[ni][1][s], xb, P[1][ti]);
                    }
                }
            __syncthreads();          // B0: the previous step's P6 reads of `red` are complete
            write_partial(red + (size_t)w * 64 * PS, P, h, c);
#pragma unroll
            for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const f32x16 wt = transpose_tile(W2F[ni][fj][0], W2F[ni][fj][1], I0, I1);   // (rows=f, lane=n)
                    WTF[fj][ni][0] = pack(wt, 0);
                    WTF[fj][ni][1] = pack(wt, 1);
                    if (SAVE) {
                        st_frag(slot_w, FR_W2T, fr_idx(fj, ni, 0), WTF[fj][ni][0], l);
                        st_frag(slot_w, FR_W2T, fr_idx(fj, ni, 1), WTF[fj][ni][1], l);
                    }
                }
        }
        TTT_STAMP(1)
        __syncthreads();              // B1: all partials visible
        TTT_STAMP(8)

        // ================= P3: owners - reduce, fused LN / L2 backward -> gZ2 =====================
        {
            float z[16], kk[16], vv[16];
            gather_partial(red, b2L, ot, of0, z);
            float mu, rstd;
            row_stats(z, p.eps, mu, rstd);
            load16_bf16(Kt + ot * TS + of0, kk);
            load16_bf16(Vt + ot * TS + of0, vv);
            float s1 = 0.f, s2 = 0.f, gx[16], go[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float xh = (z[j] - mu) * rstd;
                const float g = gamL[of0 + j];
                go[j] = g * xh + betL[of0 + j] - (vv[j] - kk[j]);
                gx[j] = go[j] * g;
                z[j] = xh;
                s1 += gx[j]; s2 += gx[j] * xh;
            }
            s1 = quad_add(s1);
            s2 = quad_add(s2);
#pragma unroll
            for (int j = 0; j < 16; ++j) gx[j] = (64.0f * gx[j] - s1 - z[j] * s2) * rstd * (1.0f / 64.0f);
            store16_bf16(G1 + ot * TS + of0, gx);
            if (SAVE) {
                st_own<16>(own, 0, ot, of0, z);
                st_own<16>(own, 1, ot, of0, go);
                own_stats(own, ot)[0] = rstd;
                store16_bf16(reinterpret_cast<__bf16*>(slot + SLOT_FR + SLOT_OWN) + ot * 64 + of0, gx);
            }
        }
        TTT_STAMP(2)
        __syncthreads();              // B2: gZ2 visible
        TTT_STAMP(9)

        // ================= P4: gZ1, state updates ================================================
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const f32x16 etaR = rows_from_lds(etaC, 32 * ti, h);    // eta_t for the rows of this t-tile
            bf16x8 Gpi[2][2];         // [fj][s]   gZ2 (m=t, k=f) pi-read
            bf16x8 GcF[2][2];         // [fj][s]   -eta*gZ2 tile (rows=t, lane=f) packed
            bf16x8 KcF[2][2];         // [fi][s]   K tile (rows=t, lane=f) packed
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                Gpi[fj][0] = pi_read(G1 + (32 * ti + c) * TS, 32 * fj, 0, h);
                Gpi[fj][1] = pi_read(G1 + (32 * ti + c) * TS, 32 * fj, 1, h);
                f32x16 gc = transpose_tile(Gpi[fj][0], Gpi[fj][1], I0, I1);      // gZ2 (rows=t, lane=f)
                float s = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { gc[r] *= -etaR[r]; s += gc[r]; }
                s = xor_add(s, 32);
                if (w == 0 && h == 0) b2L[32 * fj + c] += s;                     // b2' = b2 - sum eta gZ2
                GcF[fj][0] = pack(gc, 0);
                GcF[fj][1] = pack(gc, 1);
                const bf16x8 k0 = pi_read(Kt + (32 * ti + c) * TS, 32 * fj, 0, h);
                const bf16x8 k1 = pi_read(Kt + (32 * ti + c) * TS, 32 * fj, 1, h);
                const f32x16 kc = transpose_tile(k0, k1, I0, I1);                // K (rows=t, lane=f)
                KcF[fj][0] = pack(kc, 0);
                KcF[fj][1] = pack(kc, 1);
            }
            bf16x8 GZ1F[2][2];        // [nj][s]   -eta*gZ1 tile (rows=t, lane=n) packed
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                f32x16 gx = zero16();
#pragma unroll
                for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                    for (int s = 0; s < 2; ++s) gx = mma(Gpi[fj][s], WTF[fj][nj][s], gx);   // gX2 (rows=t, lane=n)
                if (SAVE) {
                    f32x16 g1;
#pragma unroll
                    for (int r = 0; r < 16; ++r) g1[r] = gx[r] * D1[ti][nj][r];   // gZ1, unscaled
                    const bf16x8 g1a = pack(g1, 0), g1b = pack(g1, 1);
                    {                         // the sweep consumes the product M = gX2 * gelu''(Z1) only
                        const f32x16 d2 = unpack2(ld_frag(slot_w, FR_D2, fr_idx(ti, nj, 0), l), ld_frag(slot_w, FR_D2, fr_idx(ti, nj, 1), l));
                        f32x16 mm;
#pragma unroll
                        for (int r = 0; r < 16; ++r) mm[r] = gx[r] * d2[r];
                        st_frag(slot_w, FR_GX2, fr_idx(ti, nj, 0), pack(mm, 0), l);
                        st_frag(slot_w, FR_GX2, fr_idx(ti, nj, 1), pack(mm, 1), l);
                    }
                    const f32x16 g1t = transpose_tile(g1a, g1b, I0, I1);          // gZ1^T (rows=n, lane=t)
                    st_frag(slot_w, FR_GZ1T, fr_idx(nj, ti, 0), pack(g1t, 0), l);
                    st_frag(slot_w, FR_GZ1T, fr_idx(nj, ti, 1), pack(g1t, 1), l);
                }
                float sb = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) { gx[r] = -etaR[r] * gx[r] * D1[ti][nj][r]; sb += gx[r]; }
                sb = xor_add(sb, 32);
                b1v[nj] += sb;                                                   // b1' = b1 - sum eta gZ1
                GZ1F[nj][0] = pack(gx, 0);
                GZ1F[nj][1] = pack(gx, 1);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        W1t[a][b] = mma(KcF[a][s], GZ1F[b][s], W1t[a][b]);          // W1[f,n] -= (eta K)^T gZ1
                        W2t[a][b] = mma(X2F[ti][a][s], GcF[b][s], W2t[a][b]);       // W2[n,f] -= (eta X2)^T gZ2
                    }
        }
        if (h == 0) { b1s[w * 64 + c] = b1v[0]; b1s[w * 64 + 32 + c] = b1v[1]; }

        TTT_STAMP(3)
        // ================= P5: Z1b^T = W1'^T Q^T + b1' ; X2b ; partial Z2b^T =====================
        {
            f32x16 P[2][2];           // [fj][ti]
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) P[a][b] = zero16();
            bf16x8 Qpi[2][2][2];      // [ti][fi][s]
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int s = 0; s < 2; ++s) Qpi[ti][fi][s] = pi_read(Qt + (32 * ti + c) * TS, 32 * fi, s, h);
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                const f32x16 bias = rows_from_lds(b1s + w * 64, 32 * nj, h);
                f32x16 zb[2] = {bias, bias};                                     // [ti]  (rows=n, lane=t)
#pragma unroll
                for (int fi = 0; fi < 2; ++fi)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 wa = pack(W1t[fi][nj], s);
                        zb[0] = mma(wa, Qpi[0][fi][s], zb[0]);
                        zb[1] = mma(wa, Qpi[1][fi][s], zb[1]);
                    }
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    f32x16 db;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float y, dy = 0.f;
                        if (SAVE) gelu_fwd_grad(zb[ti][r], y, dy);
                        else y = gelu_fwd(zb[ti][r]);
                        zb[ti][r] = y;
                        db[r] = dy;
                    }
                    const bf16x8 xb0 = pack(zb[ti], 0), xb1 = pack(zb[ti], 1);
                    if (SAVE) {   // the reverse sweep wants X2b and gelu'(Z1b) in (rows=t, lane=n) orientation
                        const f32x16 xn = transpose_tile(xb0, xb1, I0, I1);
                        const f32x16 dn = transpose_tile(pack(db, 0), pack(db, 1), I0, I1);
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            st_frag(slot_w, FR_X2B, fr_idx(ti, nj, s), pack(xn, s), l);
                            st_frag(slot_w, FR_D1B, fr_idx(ti, nj, s), pack(dn, s), l);
                        }
                    }
                    P[0][ti] = mma(pack(W2t[nj][0], 0), xb0, P[0][ti]);
                    P[1][ti] = mma(pack(W2t[nj][1], 0), xb0, P[1][ti]);
                    P[0][ti] = mma(pack(W2t[nj][0], 1), xb1, P[0][ti]);
                    P[1][ti] = mma(pack(W2t[nj][1], 1), xb1, P[1][ti]);
                }
            }
            if (more) prefetch_park(pf, tiles + ((cur ^ 1) * 3) * TILE_ELEMS, etaL + (cur ^ 1) * 64);
            write_partial(red + (size_t)w * 64 * PS, P, h, c);   // P3's reads of `red` finished before B2
        }
        TTT_STAMP(4)
        __syncthreads();              // B3
        TTT_STAMP(10)

        // ================= P6: owners - reduce, LayerNorm, residual -> XQW ========================
        {
            float z[16], q[16];
            gather_partial(red, b2L, ot, of0, z);
            float mu, rstd;
            row_stats(z, p.eps, mu, rstd);
            if (SAVE) {
#pragma unroll
                for (int j = 0; j < 16; ++j) z[j] = (z[j] - mu) * rstd;
                st_own<16>(own, 2, ot, of0, z);
                own_stats(own, ot)[1] = rstd;
            } else {
                load16_bf16(Qt + ot * TS + of0, q);
#pragma unroll
                for (int j = 0; j < 16; ++j) z[j] = q[j] + gamL[of0 + j] * ((z[j] - mu) * rstd) + betL[of0 + j];
                store16_bf16(p.out + tile * 4096 + (size_t)ot * 64 + of0, z);
            }
        }
        TTT_STAMP(5)
    }
    if (SAVE && grp == p.chunk_group0 + p.chunk_groups - 1) {
        // state after the chunk's last step: the "post-update" operands of that step in the reverse sweep
        char* slot_w = slots + (size_t)(i_hi - p.chunk_lo) * SLOT_BYTES + (size_t)w * SLOT_WAVE_FR;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    st_frag(slot_w, FR_W1, fr_idx(a, b, s), pack(W1t[a][b], s), l);
                    st_frag(slot_w, FR_W2, fr_idx(a, b, s), pack(W2t[a][b], s), l);
                }
        {
#pragma unroll
            for (int fj = 0; fj < 2; ++fj)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const f32x16 wt = transpose_tile(pack(W2t[ni][fj], 0), pack(W2t[ni][fj], 1), I0, I1);
                    st_frag(slot_w, FR_W2T, fr_idx(fj, ni, 0), pack(wt, 0), l);
                    st_frag(slot_w, FR_W2T, fr_idx(fj, ni, 1), pack(wt, 1), l);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
static void set_lds_attr_once() {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)mlp_scan_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FWD);
        done = true;
    }
}

void launch_group_recompute(const ScanParams& p0, int n_bh, hipStream_t s) {
    ScanParams p = p0;
    p.dbg = nullptr;
    set_lds_attr_once();
    hipLaunchKernelGGL(mlp_scan_kernel<true>, dim3(n_bh * p.chunk_groups), dim3(NT), LDS_FWD, s, p);
}

bool supports(const ttt_dims* d, bool mlp, bool backward) {
    if (!(d->F == 64 && d->act_dtype == TTT_DTYPE_BF16)) return false;
    if (d->CS == 16) return !backward || !mlp;  // mini-batches of 16 (ttt_mfma16.hip): MLP forward, Linear forward + backward
    if (!mlp) return false;
    return d->CS == 64 && (!backward || bwd_available());
}

void mlp_forward(const ttt_dims* d, const ttt_mlp_fwd_args* a, void*, hipStream_t s) {
    ScanParams p = {};
    p.XQ = (const __bf16*)a->XQ; p.XK = (const __bf16*)a->XK; p.XV = (const __bf16*)a->XV; p.eta = (const __bf16*)a->last_eta;
    p.ln_w = a->ttt_norm_weight; p.ln_b = a->ttt_norm_bias;
    p.W1 = a->W1_init; p.b1 = a->b1_init; p.W2 = a->W2_init; p.b2 = a->b2_init;
    p.W1c = a->W1_checkpoints; p.b1c = a->b1_checkpoints; p.W2c = a->W2_checkpoints; p.b2c = a->b2_checkpoints;
    p.out = (__bf16*)a->XQW;
    p.NH = d->NH; p.NC = d->NC; p.G = d->G; p.K = (d->NC + d->G - 1) / d->G; p.eps = d->eps;
    if (d->CS == 16) launch_scan_forward_cs16(p, d->B * d->NH, g_dbg, s);
    else launch_scan_forward_v2(p, d->B * d->NH, g_dbg, s);
}
void linear_forward(const ttt_dims* d, const ttt_linear_fwd_args* a, void*, hipStream_t s) {
    wv::Lin16Params p = {};
    p.XQ = (const __bf16*)a->XQ; p.XK = (const __bf16*)a->XK; p.XV = (const __bf16*)a->XV; p.eta = (const __bf16*)a->last_eta;
    p.ln_w = a->ttt_norm_weight; p.ln_b = a->ttt_norm_bias;
    p.W1 = a->W1_init; p.b1 = a->b1_init;
    p.W1c = a->W1_checkpoints; p.b1c = a->b1_checkpoints;
    p.out = (__bf16*)a->XQW;
    p.NH = d->NH; p.NC = d->NC; p.G = d->G; p.K = (d->NC + d->G - 1) / d->G; p.eps = d->eps;
    launch_linear_forward_cs16(p, d->B * d->NH, s);
}
void linear_backward(const ttt_dims* d, const ttt_linear_bwd_args* a, void*, hipStream_t s) {
    wv::Lin16Params p = {};
    p.XQ = (const __bf16*)a->XQ; p.XK = (const __bf16*)a->XK; p.XV = (const __bf16*)a->XV; p.eta = (const __bf16*)a->last_eta;
    p.ln_w = a->ttt_norm_weight; p.ln_b = a->ttt_norm_bias;
    p.W1c = const_cast<float*>(a->W1_checkpoints); p.b1c = const_cast<float*>(a->b1_checkpoints);      // read only here
    p.dOut = (const __bf16*)a->grad_L_XQW;
    p.dW1_last = a->grad_L_W1_last; p.db1_last = a->grad_L_b1_last;
    p.scratch_w = (char*)a->W1_init_group; p.scratch_b = a->b1_init_group;      // G x 16 KiB and G x 64 floats per (b,h)
    p.dln_w = a->grad_L_ttt_norm_weight; p.dln_b = a->grad_L_ttt_norm_bias;
    p.dW1 = a->grad_L_W1_init; p.db1 = a->grad_L_b1_init;
    p.deta = (__bf16*)a->grad_L_last_eta; p.dXQ = (__bf16*)a->grad_L_XQ; p.dXK = (__bf16*)a->grad_L_XK; p.dXV = (__bf16*)a->grad_L_XV;
    p.NH = d->NH; p.NC = d->NC; p.G = d->G; p.K = (d->NC + d->G - 1) / d->G; p.eps = d->eps;
    launch_linear_backward_cs16(p, d->B * d->NH, s);
}

}  // namespace mfma
}  // namespace ttt
