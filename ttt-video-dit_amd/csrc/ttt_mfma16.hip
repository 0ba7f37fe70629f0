// MFMA TTT-MLP forward scan for mini-batches of 16 tokens (gfx950): the evaluation / sampling geometry of the reference
// (configs/eval/ttt-mlp/*.toml: mini_batch_size = 16, no scan checkpoints; ttt_layer.py:429-473 -> mlp_tk.py forward).
//
// Same primal-form step as the CS = 64 kernel (ttt_mfma2.hip) but a different machine mapping: with
// 16 tokens the products that carry a token dimension are 16 wide, so everything runs on the 16x16 MFMA shapes
//   mma32 = v_mfma_f32_16x16x32_bf16  (contractions over the 64 features / the hidden units),
//   mma16 = v_mfma_f32_16x16x16_bf16  (contractions over the 16 tokens: the state updates),
// and the state lives in 16x16 fp32 accumulator tiles (lane (g, i) = (l >> 4, l & 15) holds D[4g + r][i], r = 0..3).
// Layout algebra: a 16x16 tile X (rows = R, lane = C) is, in place,
//   * an mma16 operand contracting over R: lane's k-slot e carries row 4g + e            (identity order),
//   * half of an mma32 operand contracting over R: two tiles stacked along R give k-slot (g, e) = row 4g + e of the first
//     (e < 4) or of the second (e >= 4) tile ("rho" order); the partner operand presents the same order from a row-major
//     LDS tile with two 8-byte reads at columns c0 + 4g and c0 + 16 + 4g.
// Contraction over the LANE index goes through LDS: the wave writes its tile as an image [lane index][row index] (one
// 8-byte store per lane) and reads it back with ds_read_b64_tr_b16 (private region, no barrier).
//
// Work split: 8 waves; wave w owns hidden units Hw = [32w, 32w + 32): W1[:, Hw] (tiles rows = f, lane = n), W2[Hw, :]
// (rows = n, lane = f) and a second accumulator copy W2^T[:, Hw] (rows = f, lane = n) for the contraction over f in
// gX2 = gZ2 W2^T.  Layer 1 is local to the wave; the two contractions over the hidden units leave 8 fp32 partials per
// element in LDS, which "owner" threads (16 lanes x 4 features per token) reduce.  Per step i, B* = workgroup barriers:
//   A1  Z1 = K W1 + b1 -> X2 = gelu, D1 = gelu'   (rows = t, lane = n) ; X2 image
//   A2  partial Z2^T[f, t] = W2[Hw, :]^T X2[:, Hw]^T -> redA
//   B1
//   P3  waves 0-3: sum partials + b2, fused LayerNorm / L2 backward, Gs = -eta gZ2 -> LDS [t][f] bf16
//   P6  waves 4-7, for step i-1 (off the critical path): sum the redB partials + b2, LayerNorm, + Q -> XQW
//       all: park the inputs of step i+1 (loaded one step earlier)
//   B2
//   C   b2 += colsum Gs (ones MFMA) ; W2 += X2^T Gs ; W2^T += Gs^T X2 ; gX2s = Gs W2^T (entering W2) ; gZ1s = gX2s * D1 ;
//       W1 += K^T gZ1s ; b1 += colsum gZ1s ; Z1b = Q W1' + b1' ; X2b = gelu ; X2b image
//   E   partial Z2b^T = W2'[Hw, :]^T X2b^T -> redB
// Two barriers per step are enough because every shared buffer has one writer phase and one reader phase on opposite sides
// of a barrier: redA (A2 | P3), Gs (P3 | C), redB (E | P6 of the next step), b2 in LDS (C | P3, P6), K / V double-buffered
// and Q triple-buffered (parked between B1 and B2 of the step before they are used; Q of step i-1 is still being read by
// P6 at that time, hence the third buffer).
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"
#define TTT_WV_FN __device__ __forceinline__
#include "ttt_lin16_body.h"
#include "ttt_mlp16_body.h"

namespace ttt {
namespace mfma {
using namespace ttt::mf;

namespace v16 {

constexpr int NT16 = 512;
constexpr int CS16 = 16;
constexpr int TILE16 = CS16 * TS;                     // elements of a padded [16][64] bf16 tile
constexpr int IS = 24;                                // row stride of a wave's [32 n][16 t] image (4 rows on disjoint banks)
constexpr int L_K = 0;                                // K, V: 2 buffers each, Q: 3
constexpr int L_V = L_K + 2 * TILE16 * 2;
constexpr int L_Q = L_V + 2 * TILE16 * 2;
constexpr int L_G = L_Q + 3 * TILE16 * 2;
constexpr int L_IMG = L_G + TILE16 * 2;
constexpr int IMG_BYTES = 32 * IS * 2;
constexpr int L_REDA = L_IMG + 8 * IMG_BYTES;
constexpr int RED16_BYTES = 8 * CS16 * PS * 4;        // [8 waves][16 t][PS] fp32
constexpr int L_REDB = L_REDA + RED16_BYTES;
constexpr int L_SMALL = L_REDB + RED16_BYTES;         // eta[2][16], b2[64], gamma[64], beta[64]
constexpr int LDS_V16 = L_SMALL + (32 + 64 + 64 + 64) * 4;
static_assert(LDS_V16 <= 160 * 1024, "LDS budget");
static_assert(L_IMG % 16 == 0 && L_REDA % 16 == 0 && L_SMALL % 16 == 0, "alignment");

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ bf16x4 pack4(f32x4 v) {
    bf16x4 r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return r;
}
// two tiles stacked along their row index -> one K = 32 operand in rho order
__device__ __forceinline__ bf16x8 stack(f32x4 a, f32x4 b) {
    bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return r;
}
// rho-order operand from this lane's row of a row-major tile: columns c0 + 4g .. +3 and c0 + 16 + 4g .. +3
__device__ __forceinline__ bf16x8 rho_read(const __bf16* rowp, int c0, int g) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(rowp + c0 + 4 * g);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(rowp + c0 + 16 + 4 * g);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// transposed read: lane (g, i) gets img[row0 + 4g + e][col0 + i], e = 0..3  (operand with outer = column, k = row)
__device__ __forceinline__ bf16x4 tr4(const __bf16* img, int stride, int row0, int col0, int l) {
    const int g = l >> 4, i = l & 15;
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + (row0 + 4 * g + (i >> 2)) * stride + col0 + 4 * (i & 3)));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror
    return v;
}

// cycle stamps of workgroup 0 / thread 0, accumulated in registers (a read-modify-write of global memory per stamp would
// put an L2 round trip - and the wait for every prefetch in flight - into each measured stage) and written once at the end
#define TTT_STAMP16(k)                                                       \
    if (DBG && stamp_on) {                                                   \
        const unsigned long long _t = __builtin_readcyclecounter();          \
        dbg_acc[k] += _t - t_last;                                           \
        t_last = _t;                                                         \
    }

// owner thread (token ot, features of0 .. of0 + 3): z = bias + sum of the 8 waves' partials
__device__ __forceinline__ f32x4 gather8(const float* red, const float* bias, int ot, int of0) {
    f32x4 z = *reinterpret_cast<const f32x4*>(bias + of0);
#pragma unroll
    for (int ww = 0; ww < 8; ++ww) z += *reinterpret_cast<const f32x4*>(red + ((size_t)ww * CS16 + ot) * PS + of0);
    return z;
}
__device__ __forceinline__ void row_stats16(f32x4 z, float eps, float& mu, float& rstd) {
    mu = sum16(z[0] + z[1] + z[2] + z[3]) * (1.0f / 64.0f);
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float d = z[j] - mu; v += d * d; }
    rstd = __builtin_amdgcn_rsqf(sum16(v) * (1.0f / 64.0f) + eps);
}

template <bool DBG>
__global__ __launch_bounds__(NT16) void mlp_scan16_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Kb = reinterpret_cast<__bf16*>(smem + L_K);
    __bf16* Vb = reinterpret_cast<__bf16*>(smem + L_V);
    __bf16* Qb = reinterpret_cast<__bf16*>(smem + L_Q);
    __bf16* Gs = reinterpret_cast<__bf16*>(smem + L_G);
    float* redA = reinterpret_cast<float*>(smem + L_REDA);
    float* redB = reinterpret_cast<float*>(smem + L_REDB);
    float* etaL = reinterpret_cast<float*>(smem + L_SMALL);     // [2][16]
    float* b2L = etaL + 32;
    float* gamL = b2L + 64;
    float* betL = gamL + 64;

    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = 32 * wv;
    __bf16* img = reinterpret_cast<__bf16*>(smem + L_IMG + wv * IMG_BYTES);
    const int NC = p.NC, G = p.G;
    const int bh = blockIdx.x, head = bh % p.NH;

    // ---- state -------------------------------------------------------------------------------------------------------
    f32x4 W1t[4][2];     // [fb][nb]  W1[16fb + 4g + r][n0 + 16nb + i]           (rows = f, lane = n)
    f32x4 W2t[2][4];     // [nb][fb]  W2[n0 + 16nb + 4g + r][16fb + i]           (rows = n, lane = f)
    f32x4 W2Tt[4][2];    // [fb][nb]  W2[n0 + 16nb + i][16fb + 4g + r]           (rows = f, lane = n)
    float b1v[2];        // b1[n0 + 16nb + i]
    float b2v[4];        // b2[16fb + i]      (every wave keeps the same copy)
    {
        const int l = tid & 63, g = l >> 4, i = l & 15;
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
        if (tid < 64) {
            b2L[tid] = p.b2[(size_t)bh * 64 + tid];
            gamL[tid] = p.ln_w[(size_t)head * 64 + tid];
            betL[tid] = p.ln_b[(size_t)head * 64 + tid];
        }
    }
    const bf16x4 ONES = {(__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f, (__bf16)1.0f};
    // packed operands of the current state, re-made right after each update and carried into the next step (the state is
    // packed once per step instead of once per use: -32 conversions per wave and step for 16 live registers)
    bf16x8 W1F[2][2];    // [ks][nb]  rows f = 32ks .. 32ks + 31 of W1[:, n-block nb]
    bf16x8 W2F[4];       // [fb]      rows n = Hw of W2[:, f-block fb]
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb) W1F[ks][nb] = stack(W1t[2 * ks][nb], W1t[2 * ks + 1][nb]);
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) W2F[fb] = stack(W2t[0][fb], W2t[1][fb]);

    // ---- input staging.  Thread groups of 128 move one 16-byte chunk of K / V / Q each (waves 6, 7 mirror Q's loads and
    // do not store), every thread carries one eta value.  The loads are unconditional and branch-free - tile indices are
    // clamped instead - so that hipcc can keep them in flight (a load inside a branch costs a vmcnt(0) at the join).
    // Tile i+2 is requested at the top of step i and parked between B1 and B2 of step i+1.
    const size_t tile0 = (size_t)bh * NC;
    const int which = wv >> 1;                                      // 0 K, 1 V, 2 Q, 3 none   (wave-uniform)
    const __bf16* src = which == 0 ? p.XK : which == 1 ? p.XV : p.XQ;
    __bf16* dstb = which == 0 ? Kb : which == 1 ? Vb : Qb;
    const int nbufs = which == 2 ? 3 : 2;
    const int lt0 = tid & 127, lofs = (lt0 >> 3) * TS + (lt0 & 7) * 8;       // chunk position inside a padded tile
    const size_t gofs = (size_t)(lt0 >> 3) * 64 + (lt0 & 7) * 8;
    uint4 pfO;
    unsigned short pfEO;
    {
        const uint4 t0 = *reinterpret_cast<const uint4*>(src + tile0 * 1024 + gofs);
        if (which < 3) *reinterpret_cast<uint4*>(dstb + lofs) = t0;
        if (tid < 16) etaL[tid] = (float)p.eta[tile0 * 16 + tid];
        const size_t t1 = tile0 + (NC > 1 ? 1 : 0);
        pfO = *reinterpret_cast<const uint4*>(src + t1 * 1024 + gofs);
        pfEO = *reinterpret_cast<const unsigned short*>(p.eta + t1 * 16 + (tid & 15));
    }
    __syncthreads();

    const bool stamp_on = DBG && p.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0;
    unsigned long long dbg_acc[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_last = __builtin_readcyclecounter();
    for (int it = 0; it <= NC; ++it) {      // iteration NC only drains the last P6
        const size_t tile = tile0 + it;
        const bool live = it < NC;
        const int buf = it & 1;
        // opaque per-iteration lane ids: keeps the (many) LDS / global addresses from being hoisted out of the loop and spilled
        int l_op = tid & 63, tid_op = tid;
        asm volatile("" : "+v"(l_op), "+v"(tid_op));
        const int l = l_op, g = l >> 4, i = l & 15;
        const int tid = tid_op;
        const int ot = (tid & 255) >> 4, of0 = 4 * (tid & 15);     // owner geometry: 16 lanes x 4 features per token
        const __bf16* Kt = Kb + buf * TILE16;
        const __bf16* Vt = Vb + buf * TILE16;
        const __bf16* Qt = Qb + (it % 3) * TILE16;

        // request tile it+2 (clamped: the tail re-reads the last tile and never parks it)
        const size_t tn = tile0 + (it + 2 < NC ? it + 2 : NC - 1);
        const uint4 pfN = *reinterpret_cast<const uint4*>(src + tn * 1024 + gofs);
        const unsigned short pfEN = *reinterpret_cast<const unsigned short*>(p.eta + tn * 16 + (tid & 15));

        f32x4 D1[2];
        bf16x4 X2p[2];
        if (live) {
            if (it % G == 0) {      // checkpoint: state entering step `it` (mlp_tk.py:95-98)
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
            // ================= A1: Z1 = K W1 + b1 ; X2, D1 (rows = t, lane = n) ; X2 image [n][t] =====================
            const bf16x8 kA0 = rho_read(Kt + i * TS, 0, g), kA1 = rho_read(Kt + i * TS, 32, g);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 Z = zero4();
                Z = mma32(kA0, W1F[0][nb], Z);
                Z = mma32(kA1, W1F[1][nb], Z);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float y, dy;
                    gelu_fwd_grad(Z[r] + b1v[nb], y, dy);
                    Z[r] = y;
                    D1[nb][r] = dy;
                }
                X2p[nb] = pack4(Z);
                *reinterpret_cast<bf16x4*>(img + (16 * nb + i) * IS + 4 * g) = X2p[nb];
            }
            // ================= A2: partial Z2^T[f, t] over Hw ============================================================
            const bf16x4 lo = tr4(img, IS, 0, 0, l), hi = tr4(img, IS, 16, 0, l);
            const bf16x8 xB = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);       // lane = t, k = n (rho)
            float* dst = redA + ((size_t)wv * CS16 + i) * PS + 4 * g;
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
                *reinterpret_cast<f32x4*>(dst + 16 * fb) = mma32(W2F[fb], xB, zero4());
        }
        TTT_STAMP16(0)
        __syncthreads();              // B1: redA, and redB / b2 of the previous step, visible
        TTT_STAMP16(4)

        // park tile it+1 (requested one step ago): K / V buffers were last read in C of step it-1, Q has 3 buffers.  The P3
        // waves park after their critical-path work, the others right away.
        auto park = [&]() {
            if (which < 3) *reinterpret_cast<uint4*>(dstb + ((it + 1) % nbufs) * TILE16 + lofs) = pfO;
            if (tid < 16) etaL[(buf ^ 1) * 16 + tid] = (float)__builtin_bit_cast(__bf16, pfEO);
            pfO = pfN;
            pfEO = pfEN;
        };
        if (wv < 4) {
            // ================= P3: owners - reduce, fused LN / L2 backward -> Gs = -eta gZ2 =============================
            // (s_setprio 3 around this block, against the P6 wave on the same SIMD, measured no gain: 2.67 vs 2.64 us/step)
            if (live) {
                f32x4 z = gather8(redA, b2L, ot, of0);
                float mu, rstd;
                row_stats16(z, p.eps, mu, rstd);
                const bf16x4 kk = *reinterpret_cast<const bf16x4*>(Kt + ot * TS + of0);
                const bf16x4 vv = *reinterpret_cast<const bf16x4*>(Vt + ot * TS + of0);
                const f32x4 gm = *reinterpret_cast<const f32x4*>(gamL + of0), bt = *reinterpret_cast<const f32x4*>(betL + of0);
                float s1 = 0.f, s2 = 0.f, gx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float xh = (z[j] - mu) * rstd;
                    gx[j] = (gm[j] * xh + bt[j] - ((float)vv[j] - (float)kk[j])) * gm[j];
                    z[j] = xh;
                    s1 += gx[j]; s2 += gx[j] * xh;
                }
                s1 = sum16(s1);
                s2 = sum16(s2);
                const float sc = -etaL[buf * 16 + ot] * rstd * (1.0f / 64.0f);
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (__bf16)((64.0f * gx[j] - s1 - z[j] * s2) * sc);
                *reinterpret_cast<bf16x4*>(Gs + ot * TS + of0) = o;
            }
            park();
        } else {
            park();
            if (it > 0) {
            // ================= P6 (step it-1): owners - reduce, LayerNorm, residual -> XQW ================================
            const f32x4 z = gather8(redB, b2L, ot, of0);
            float mu, rstd;
            row_stats16(z, p.eps, mu, rstd);
            const bf16x4 q = *reinterpret_cast<const bf16x4*>(Qb + ((it - 1) % 3) * TILE16 + ot * TS + of0);
            const f32x4 gm = *reinterpret_cast<const f32x4*>(gamL + of0), bt = *reinterpret_cast<const f32x4*>(betL + of0);
            bf16x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] = (__bf16)((float)q[j] + gm[j] * ((z[j] - mu) * rstd) + bt[j]);
            *reinterpret_cast<bf16x4*>(p.out + (tile - 1) * 1024 + (size_t)ot * 64 + of0) = o;
            }
        }
        if (!live) break;
        TTT_STAMP16(1)
        __syncthreads();              // B2: Gs and the parked tiles visible; redB / b2 in LDS free to be rewritten
        TTT_STAMP16(5)

        // ================= C: state updates ; gX2 ; W1 update ; Z1b ==========================================================
        {
            // operands of the ENTERING W2^T for gX2, packed before the accumulator copy is updated
            bf16x8 W2TF[2][2];        // [ks][nb]
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) W2TF[ks][nb] = stack(W2Tt[2 * ks][nb], W2Tt[2 * ks + 1][nb]);
            // critical path first: gX2s = Gs W2^T ; gZ1s = gX2s * D1 ; W1 += K^T gZ1s
            const bf16x8 gA0 = rho_read(Gs + i * TS, 0, g), gA1 = rho_read(Gs + i * TS, 32, g);
            bf16x4 gzp[2];
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 gx = zero4();
                gx = mma32(gA0, W2TF[0][nb], gx);
                gx = mma32(gA1, W2TF[1][nb], gx);
#pragma unroll
                for (int r = 0; r < 4; ++r) gx[r] *= D1[nb][r];
                gzp[nb] = pack4(gx);                                                          // lane = n, k = t
                b1v[nb] += mma16(ONES, gzp[nb], zero4())[0];                                  // b1' = b1 + colsum_t gZ1s
            }
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const bf16x4 kT = tr4(Kt, TS, 0, 16 * fb, l);                                 // lane = f, k = t
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) W1t[fb][nb] = mma16(kT, gzp[nb], W1t[fb][nb]);  // W1[f, n] += K^T gZ1s
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) W1F[ks][nb] = stack(W1t[2 * ks][nb], W1t[2 * ks + 1][nb]);
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                const bf16x4 gT = tr4(Gs, TS, 0, 16 * fb, l);                                 // lane = f, k = t
                b2v[fb] += mma16(ONES, gT, zero4())[0];                                       // column sums of Gs
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    W2t[nb][fb] = mma16(X2p[nb], gT, W2t[nb][fb]);                            // W2[n, f] += X2^T Gs
                    W2Tt[fb][nb] = mma16(gT, X2p[nb], W2Tt[fb][nb]);                          // W2^T[f, n] += Gs^T X2
                }
                W2F[fb] = stack(W2t[0][fb], W2t[1][fb]);
            }
            // Z1b = Q W1' + b1' ; X2b = gelu ; image
            const bf16x8 qA0 = rho_read(Qt + i * TS, 0, g), qA1 = rho_read(Qt + i * TS, 32, g);
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                f32x4 Z = zero4();
                Z = mma32(qA0, W1F[0][nb], Z);
                Z = mma32(qA1, W1F[1][nb], Z);
#pragma unroll
                for (int r = 0; r < 4; ++r) Z[r] = gelu_fwd(Z[r] + b1v[nb]);
                *reinterpret_cast<bf16x4*>(img + (16 * nb + i) * IS + 4 * g) = pack4(Z);
            }
        }
        TTT_STAMP16(2)
        // ================= E: partial Z2b^T -> redB ; b2' -> LDS =================================================================
        {
            const bf16x4 lo = tr4(img, IS, 0, 0, l), hi = tr4(img, IS, 16, 0, l);
            const bf16x8 xB = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            float* dst = redB + ((size_t)wv * CS16 + i) * PS + 4 * g;
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
                *reinterpret_cast<f32x4*>(dst + 16 * fb) = mma32(W2F[fb], xB, zero4());
        }
        if (wv == 0 && g == 0) {
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) b2L[16 * fb + i] = b2v[fb];   // b2' for the next step's P3 and for P6 of this step
        }
        TTT_STAMP16(3)
    }
    if (DBG && stamp_on) {
#pragma unroll
        for (int k = 0; k < 6; ++k) p.dbg[k] += dbg_acc[k];
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// TTT-Linear at mini-batches of 16 tokens (the reference trains and evaluates TTT-Linear at mini_batch_size 16:
// configs/train/ttt-linear/*.toml; replaces the Triton launches linear_triton.py:98-129, :203-246).  The kernel bodies live
// in ttt_lin16_body.h, written against a wave backend so that the CPU test-suite can execute the same code on a lane-level
// emulator (tests/emul); here is the device backend: the gfx950 instructions behind those primitives.  One wave per (b, h),
// 4 waves (4 heads) per workgroup only to fill the CU's four SIMDs; no workgroup barriers.
struct DeviceWave {
    char* base;                                                       // this wave's private LDS region
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    __device__ __forceinline__ int wave() const { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    __device__ __forceinline__ int thread() const { return threadIdx.x; }
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
    __device__ __forceinline__ float exp2(float x) const { return __builtin_amdgcn_exp2f(x); }
    __device__ __forceinline__ float rcp(float x) const { return __builtin_amdgcn_rcpf(x); }
    __device__ __forceinline__ int opaque(int v) const { asm volatile("" : "+v"(v)); return v; }
    __device__ __forceinline__ void lds_fence() const { asm volatile("" ::: "memory"); }    // LDS is in order within a wave
    template <class T> __device__ __forceinline__ T& lds(int byte_off) const { return *reinterpret_cast<T*>(base + byte_off); }
    __device__ __forceinline__ char* lds_ptr(int byte_off) const { return base + byte_off; }
    template <class T> __device__ __forceinline__ T lds_load(int byte_off) const { return *reinterpret_cast<const T*>(base + byte_off); }
    template <class T> __device__ __forceinline__ void lds_store(int byte_off, T v) const { *reinterpret_cast<T*>(base + byte_off) = v; }
    __device__ __forceinline__ float rsq(float x) const { return __builtin_amdgcn_rsqf(x); }
    __device__ __forceinline__ f32x4 mma32(bf16x8 a, bf16x8 b, f32x4 c) const { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ __forceinline__ f32x4 mma16(bf16x4 a, bf16x4 b, f32x4 c) const {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
    __device__ __forceinline__ bf16x4 tr_read(int byte_addr) const {
        typedef __attribute__((address_space(3))) bf16x4 lds_b4;
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(base + byte_addr));
    }
    __device__ __forceinline__ float sum16(float v) const { return v16::sum16(v); }
    __device__ __forceinline__ float xor_add(float v, int mask) const { return v + __shfl_xor(v, mask, 64); }
};

constexpr int LIN_WAVES = 4;
constexpr int LDS_LIN = LIN_WAVES * lin16::WAVE_LDS;

__global__ __launch_bounds__(64 * LIN_WAVES) void linear_scan16_kernel(wv::Lin16Params p, int n_bh) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x * LIN_WAVES + w;
    if (bh >= n_bh) return;                                          // whole wave; no barriers in this kernel
    DeviceWave bk{smem + w * lin16::WAVE_LDS};
    lin16::forward(bk, p, bh);
}

// TTT-MLP forward scan as the backend-templated workgroup body (ttt_mlp16_body.h): opt-in variant of mlp_scan16_kernel
// (debug option "scan16_body") until it has been timed against it on an MI355X
__global__ __launch_bounds__(NT16) void mlp_scan16_body_kernel(wv::Mlp16Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DeviceWave bk{smem};
    mlp16::forward(bk, p, blockIdx.x);
}

// backward: one wave per workgroup (48 .. 96 scans on 256 CUs: a CU of its own per scan; up to 512 registers per lane)
template <bool LDS_SLOTS>
__global__ __launch_bounds__(64) void linear_bwd16_kernel(wv::Lin16Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DeviceWave bk{smem};
    lin16::backward<LDS_SLOTS>(bk, p, blockIdx.x);
}

static void set_attr_once() {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)mlp_scan16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_V16);
        (void)hipFuncSetAttribute((const void*)mlp_scan16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_V16);
        done = true;
    }
}

}  // namespace v16

static void lin_attr_once() {
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void*)v16::linear_scan16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, v16::LDS_LIN);
        (void)hipFuncSetAttribute((const void*)v16::linear_bwd16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lin16::WAVE_LDS_BWD);
        (void)hipFuncSetAttribute((const void*)v16::linear_bwd16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lin16::WAVE_LDS_BWD);
        done = true;
    }
}
void launch_linear_forward_cs16(const wv::Lin16Params& p, int n_bh, hipStream_t s) {
    lin_attr_once();
    const int blocks = (n_bh + v16::LIN_WAVES - 1) / v16::LIN_WAVES;
    hipLaunchKernelGGL(v16::linear_scan16_kernel, dim3(blocks), dim3(64 * v16::LIN_WAVES), v16::LDS_LIN, s, p, n_bh);
}
void launch_linear_backward_cs16(const wv::Lin16Params& p, int n_bh, hipStream_t s) {
    lin_attr_once();
    // LDS: the fixed regions + the state slots kept in LDS (debug option "linear_bwd_lds_slots", default 0 = all in scratch)
    const int n_lds = p.lds_slots < lin16::MAX_LDS_SLOTS ? (p.lds_slots > 0 ? p.lds_slots : 0) : lin16::MAX_LDS_SLOTS;
    if (n_lds > 0) hipLaunchKernelGGL(v16::linear_bwd16_kernel<true>, dim3(n_bh), dim3(64), lin16::L_SLOTS + n_lds * lin16::SLOT_BYTES, s, p);
    else hipLaunchKernelGGL(v16::linear_bwd16_kernel<false>, dim3(n_bh), dim3(64), lin16::L_SLOTS, s, p);
}

void launch_scan_forward_cs16(const ScanParams& p0, int n_bh, unsigned long long* dbg, hipStream_t s) {
    ScanParams p = p0;
    p.dbg = dbg;
    v16::set_attr_once();
    if (get_debug_scan16_body()) {
        static bool done = false;
        if (!done) {
            (void)hipFuncSetAttribute((const void*)v16::mlp_scan16_body_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlp16::GROUP_LDS);
            done = true;
        }
        wv::Mlp16Params q = {};
        q.XQ = p.XQ; q.XK = p.XK; q.XV = p.XV; q.eta = p.eta; q.ln_w = p.ln_w; q.ln_b = p.ln_b;
        q.W1 = p.W1; q.b1 = p.b1; q.W2 = p.W2; q.b2 = p.b2; q.W1c = p.W1c; q.b1c = p.b1c; q.W2c = p.W2c; q.b2c = p.b2c;
        q.out = p.out; q.NH = p.NH; q.NC = p.NC; q.G = p.G; q.K = p.K; q.eps = p.eps;
        hipLaunchKernelGGL(v16::mlp_scan16_body_kernel, dim3(n_bh), dim3(v16::NT16), mlp16::GROUP_LDS, s, q);
        return;
    }
    if (p.dbg) hipLaunchKernelGGL(v16::mlp_scan16_kernel<true>, dim3(n_bh), dim3(v16::NT16), v16::LDS_V16, s, p);
    else hipLaunchKernelGGL(v16::mlp_scan16_kernel<false>, dim3(n_bh), dim3(v16::NT16), v16::LDS_V16, s, p);
}

}  // namespace mfma
}  // namespace ttt
