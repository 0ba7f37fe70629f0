// TTT-MLP backward, phase A of revision 4: group recompute in the 8-wave register-resident form of the forward scan
// (ttt_mfma2.hip: VGPR-form MFMAs, no spills, two waves per SIMD), writing the SLIM step record of ttt_bwd4_dev.h.
//
// One workgroup per (b, h, checkpoint group of the chunk): it re-runs the group's forward steps from the group's checkpoint
// and stores, per step, Z1 and Z1b (bf16 T-fragment images, 32 KiB each), the gZ2 tile (bf16, 8 KiB), the owner rows of the two
// LayerNorms (fp32) - 120.5 KiB where round 2's 4-wave kernel stored 440 KiB through one CU's store path with 84 spilled
// registers and ~1 400 accumulator copies per step (0.54 ms per chunk launch however many workgroups ran).  Nothing else is
// kept: activations, their derivatives, the second orientations and the per-step W1 / W2 are re-derived by the sweep.
//
// Differences from the forward body, all in the second half of a step: Z1b is formed in the T orientation (rows = t, lane = n;
// the sweep's output path consumes it that way), so X2b reaches the layer-2 contraction through the X2 image + transposed
// reads exactly like X2 does in the first half (the forward keeps the N orientation and a fragment exchange instead: one LDS
// round trip shorter, which matters only for the latency-bound scan).  Same arithmetic per element, same GELU forms (scalar
// gelu_fwd_grad for X2 / gelu', packed for X2b): the state trajectory inside a group is the forward's, bit for bit up to the
// order in which b1' enters Z1b.
// Math: SURVEY.md Appendix A forward; reference recompute: ttt/models/ssm/mlp_tk.py:192-210 (the buffers it fills).
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"
#include "ttt_mfma_bwd_dev.h"
#include "ttt_bwd4_dev.h"
#include "once_per_device.h"

namespace ttt {
namespace mfma {
namespace s4 {
using namespace ttt::mfma::b2;            // tr_frag / tr_pi, sum8, load8_*, add8_f32, write_partial2

constexpr int NT8 = 512;
constexpr int L_K = 0;
constexpr int L_Q = L_K + TILE_ELEMS * 2;
constexpr int L_V = L_Q + TILE_ELEMS * 2;
constexpr int L_G = L_V + TILE_ELEMS * 2;
constexpr int L_X2 = L_G + TILE_ELEMS * 2;                    // [256][TS] bf16 image [n][t]: X2, later X2b
constexpr int X2IMG_BYTES = 256 * TS * 2;
constexpr int L_RED = L_X2 + X2IMG_BYTES;                     // [4][64][PS] fp32
constexpr int RED_BYTES = 4 * 64 * PS * 4;
constexpr int L_SMALL = L_RED + RED_BYTES;                    // eta[64], b2[64], gamma[64], beta[64]
constexpr int LDS_RC4 = L_SMALL + (64 + 64 + 64 + 64) * 4;
static_assert(LDS_RC4 <= 160 * 1024, "LDS budget");

__device__ __forceinline__ bf16x8 tr_frag_pi(const __bf16* img, int stride, int row0, int s, int col0, int l) {
    const int h = l >> 5;
    return tr_frag(img, stride, row0 + 16 * s + 4 * h, row0 + 16 * s + 8 + 4 * h, col0, l);
}
// fragment `idx` of array `arr` of a hidden slice region: one 16-byte store per lane, lane-linear (1 KiB per wave instruction)
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
template <bool NTS>
__device__ __forceinline__ void st16(void* p, u32x4v v) {
    if (NTS) __builtin_nontemporal_store(v, reinterpret_cast<u32x4v*>(p));
    else *reinterpret_cast<u32x4v*>(p) = v;
}
template <bool NTS>
__device__ __forceinline__ void st_frag4(char* slice, int arr, int idx, bf16x8 v, int lane) {
    st16<NTS>(slice + fro4(arr, idx) + lane * 16, __builtin_bit_cast(u32x4v, v));
}
template <bool NTS, int N>
__device__ __forceinline__ void st_rows(char* own, int arr, int ot, int of0, const float (&v)[N]) {
    float* p = reinterpret_cast<float*>(own + (size_t)arr * SLOT_OWN_ARR) + ot * 64 + of0;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
        const f32x4 x = {v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
        st16<NTS>(p + 4 * q, __builtin_bit_cast(u32x4v, x));
    }
}

template <bool NTS>
__global__ __launch_bounds__(NT8) void mlp_recompute8_kernel(RecomputeParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Kt = reinterpret_cast<__bf16*>(smem + L_K);
    __bf16* Qt = reinterpret_cast<__bf16*>(smem + L_Q);
    __bf16* Vt = reinterpret_cast<__bf16*>(smem + L_V);
    __bf16* Gs = reinterpret_cast<__bf16*>(smem + L_G);
    __bf16* X2img = reinterpret_cast<__bf16*>(smem + L_X2);
    float* red = reinterpret_cast<float*>(smem + L_RED);
    float* etaL = reinterpret_cast<float*>(smem + L_SMALL);
    float* b2L = etaL + 64;
    float* gamL = b2L + 64;
    float* betL = gamL + 64;

    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, h = l >> 5, c = l & 31;
    const int w = wv >> 1, pp = wv & 1;                       // hidden slice, half
    const int nO = 64 * w + 32 * pp, nX = 64 * w + 32 * (1 - pp);
    const int fO = 32 * pp, fX = 32 * (1 - pp);
    const int NC = p.NC, G = p.G;
    // Work item = (b, h, checkpoint group of the chunk).  Beside a cluster sweep the host covers the items with several launches
    // of at most as many workgroups as CUs are free (a workgroup needs a CU of its own: 145 KiB of LDS): item0 = first item.
    const int item = p.item0 + (int)blockIdx.x;
    const int bh = item / p.chunk_groups, grp = p.chunk_group0 + item % p.chunk_groups, head = bh % p.NH;
    const int i_lo = grp * G, i_hi = (i_lo + G < NC) ? i_lo + G : NC;
    char* slots = p.slots + (size_t)bh * p.slot_stride_bh;

    // ---- state entering the group: checkpoint `grp` --------------------------------------------------
    f32x16 W1t[2];      // [a]  W1[f in 32a.., n in Hp]                      (rows=f, lane=n)
    f32x16 W2t[2];      // [0] W2[n in Hp, f in Fp], [1] W2[n in partner's, f in Fp]   (rows=n, lane=f)
    f32x16 W2Tt[2];     // [0] W2[n in Hp, f in Fp]^T, [1] W2[n in Hp, f in partner's]^T (rows=f, lane=n)
    float b1v;          // b1[nO + c]
    float b2v = 0.f;    // b2[fO + c]   (kept by the waves with w == 0)
    {
        const size_t sb = (size_t)bh * p.K + grp;
        const float* W1g = p.W1c + sb * 64 * 256;
        const float* W2g = p.W2c + sb * 256 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = row_of(r, h);
            W1t[0][r] = W1g[(size_t)ro * 256 + nO + c];
            W1t[1][r] = W1g[(size_t)(32 + ro) * 256 + nO + c];
            W2t[0][r] = W2g[(size_t)(nO + ro) * 64 + fO + c];
            W2t[1][r] = W2g[(size_t)(nX + ro) * 64 + fO + c];
            W2Tt[0][r] = W2g[(size_t)(nO + c) * 64 + fO + ro];
            W2Tt[1][r] = W2g[(size_t)(nO + c) * 64 + fX + ro];
        }
        b1v = p.b1c[sb * 256 + nO + c];
        if (w == 0) b2v = p.b2c[sb * 64 + fO + c];
        if (tid < 64) {
            b2L[tid] = p.b2c[sb * 64 + tid];
            gamL[tid] = p.ln_w[(size_t)head * 64 + tid];
            betL[tid] = p.ln_b[(size_t)head * 64 + tid];
        }
    }
    bf16x8 ONES;
#pragma unroll
    for (int e = 0; e < 8; ++e) ONES[e] = (__bf16)1.0f;

    // ---- first tiles ------------------------------------------------------------------------------
    const size_t tile0 = (size_t)bh * NC;
    uint4 pfK, pfQ, pfV;
    unsigned short pfE = 0;      // eta row of the next step as raw bf16 bits, every wave its own copy: converted when it is parked -
                                 // a conversion at the load sits behind the K / V / Q loads issued with it and waits vmcnt(0) for all of them
    {
        const int prow = tid >> 3, pcol = (tid & 7) * 8;          // one 16-byte chunk per thread per tile
        const size_t off = (tile0 + i_lo) * 4096 + (size_t)prow * 64 + pcol;
        pfK = *reinterpret_cast<const uint4*>(p.XK + off);
        pfV = *reinterpret_cast<const uint4*>(p.XV + off);
        pfQ = *reinterpret_cast<const uint4*>(p.XQ + off);
        pfE = reinterpret_cast<const unsigned short*>(p.eta)[(tile0 + i_lo) * 64 + (tid & 63)];
        *reinterpret_cast<uint4*>(Kt + prow * TS + pcol) = pfK;
        *reinterpret_cast<uint4*>(Vt + prow * TS + pcol) = pfV;
        unsigned pfEu = pfE;
        asm volatile("" : "+v"(pfEu));     // every wave consumes its load HERE (left to the compiler the conversion sinks into the branch below,
                                          // the register stays pending in the other waves, and its pairing with b1v in A1 waits vmcnt(0) there)
        const float pfEf = __builtin_bit_cast(float, pfEu << 16);
        if (tid < 64) etaL[tid] = pfEf;
    }
    bf16x8 W1F[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int s = 0; s < 2; ++s) W1F[a][s] = pack(W1t[a], s);
    __syncthreads();

    for (int i = i_lo; i < i_hi; ++i) {
        const size_t tile = tile0 + i;
        const bool more = (i + 1 < i_hi);
        // per-iteration opaque lane / thread index (see ttt_mfma2.hip: keeps ~80 address registers from being hoisted and spilled)
        int l_op = tid & 63, tid_op = tid;
        asm volatile("" : "+v"(l_op), "+v"(tid_op));
        const int l = l_op, h = l >> 5, c = l & 31;
        const int tid = tid_op;
        const int ot = tid >> 3, of0 = 8 * (tid & 7);             // owner geometry: token ot, features of0 .. of0+7
        const int prow = tid >> 3, pcol = (tid & 7) * 8;
        char* slot = slots + (size_t)(i - p.chunk_lo) * SLOT4_BYTES;
        char* slice = slot + (size_t)w * SLICE_BYTES;
        char* own = slot + SLOT4_FR;

        // ================= A1: Z1 = K W1 + b1 (stored) ; X2, D1 ; X2 image ========================
        f32x16 D1[2];                 // gelu'(Z1)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            f32x16 Z = zero16();
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    Z = mma(pi_read(Kt + (32 * ti + c) * TS, 32 * a, s, h), W1F[a][s], Z);
#pragma unroll
            for (int r = 0; r < 16; ++r) Z[r] += b1v;
            st_frag4<NTS>(slice, A_Z1, fr_idx(ti, pp, 0), pack(Z, 0), l);
            st_frag4<NTS>(slice, A_Z1, fr_idx(ti, pp, 1), pack(Z, 1), l);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y, dy;
                gelu_fwd_grad(Z[r], y, dy);
                Z[r] = y;
                D1[ti][r] = dy;
            }
            asm volatile("" : "+v"(D1[ti]));
#pragma unroll
            for (int s = 0; s < 2; ++s) st_image(X2img + (nO + c) * TS, 32 * ti, s, h, pack(Z, s));
        }
        __syncthreads();              // B0: X2 image complete; every P6 read of step i-1 (red, b2L) is done

        // ================= A2: partial Z2^T[Fp, t] over the hidden slice ==========================
        {
            bf16x8 W2F[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W2F[a][s] = pack(W2t[a], s);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 P = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    P = mma(W2F[0][s], tr_frag_pi(X2img, TS, nO, s, 32 * ti, l), P);
                    P = mma(W2F[1][s], tr_frag_pi(X2img, TS, nX, s, 32 * ti, l), P);
                }
                write_partial2(red + (size_t)w * 64 * PS, P, ti, pp, h, c);
            }
        }
        *reinterpret_cast<uint4*>(Qt + prow * TS + pcol) = pfQ;   // Q of this step (read only after B2)
        __syncthreads();              // B1: partials visible

        // ================= P3: owners - reduce, fused LN / L2 backward -> gZ2 (stored), Gs = -eta gZ2 =======
        {
            float z[8], kk[8], vv[8];
            load8_f32(b2L + of0, z);
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) add8_f32(red + ((size_t)ww * 64 + ot) * PS + of0, z);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += z[j];
            const float mu = sum8(s) * (1.0f / 64.0f);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = z[j] - mu; v += d * d; }
            const float rstd = __builtin_amdgcn_rsqf(sum8(v) * (1.0f / 64.0f) + p.eps);
            load8_bf16(Kt + ot * TS + of0, kk);
            load8_bf16(Vt + ot * TS + of0, vv);
            float s1 = 0.f, s2 = 0.f, gx[8], go[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (z[j] - mu) * rstd;
                const float g = gamL[of0 + j];
                go[j] = g * xh + betL[of0 + j] - (vv[j] - kk[j]);
                gx[j] = go[j] * g;
                z[j] = xh;
                s1 += gx[j]; s2 += gx[j] * xh;
            }
            s1 = sum8(s1);
            s2 = sum8(s2);
            const float su = rstd * (1.0f / 64.0f), sc = -etaL[ot] * su;
            bf16x8 o, gu;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float t = 64.0f * gx[j] - s1 - z[j] * s2;
                o[j] = (__bf16)(t * sc);
                gu[j] = (__bf16)(t * su);
            }
            *reinterpret_cast<bf16x8*>(Gs + ot * TS + of0) = o;
            st16<NTS>(reinterpret_cast<__bf16*>(slot + SLOT4_FR + SLOT4_OWN) + ot * 64 + of0, __builtin_bit_cast(u32x4v, gu));
            if (p.own16) {      // round 4: these two arrays as bf16 (the sweep's owners read half the bytes; precision budget: tools/diag/
                bf16x8 zb, gb;  // lr_gate_full_emul_cpu.py points own_xh / own_go - no gradient moves; the OUTPUT LayerNorm's x_hat must stay fp32)
#pragma unroll
                for (int j = 0; j < 8; ++j) { zb[j] = (__bf16)z[j]; gb[j] = (__bf16)go[j]; }
                st16<NTS>(reinterpret_cast<__bf16*>(own) + ot * 64 + of0, __builtin_bit_cast(u32x4v, zb));
                st16<NTS>(reinterpret_cast<__bf16*>(own + SLOT_OWN_ARR) + ot * 64 + of0, __builtin_bit_cast(u32x4v, gb));
            } else {
                st_rows<NTS, 8>(own, 0, ot, of0, z);
                st_rows<NTS, 8>(own, 1, ot, of0, go);
            }
            if ((tid & 7) == 0) own_stats(own, ot)[0] = rstd;
        }
        __syncthreads();              // B2: Gs visible

        // ================= C: state updates, f3, f4 ===============================================
        {
            bf16x8 W2TF[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W2TF[a][s] = pack(W2Tt[a], s);
            if (w == 0) {
                f32x16 acc = zero16();
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc = mma(ONES, tr_frag_pi(Gs, TS, 32 * ti, s, fO, l), acc);
                b2v += acc[0];
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 gO = tr_frag_pi(Gs, TS, 32 * ti, s, fO, l);
                    const bf16x8 xO = pi_read(X2img + (nO + c) * TS, 32 * ti, s, h);
                    W2t[0] = mma(xO, gO, W2t[0]);
                    W2t[1] = mma(pi_read(X2img + (nX + c) * TS, 32 * ti, s, h), gO, W2t[1]);
                    W2Tt[0] = mma(gO, xO, W2Tt[0]);
                    W2Tt[1] = mma(tr_frag_pi(Gs, TS, 32 * ti, s, fX, l), xO, W2Tt[1]);
                }
            float sb = 0.f;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 gx = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    gx = mma(pi_read(Gs + (32 * ti + c) * TS, fO, s, h), W2TF[0][s], gx);
                    gx = mma(pi_read(Gs + (32 * ti + c) * TS, fX, s, h), W2TF[1][s], gx);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 g = f32x2{gx[r], gx[r + 1]} * f32x2{D1[ti][r], D1[ti][r + 1]};
                    gx[r] = g[0];
                    gx[r + 1] = g[1];
                    sb += g[0];
                    sb += g[1];
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 gz = pack(gx, s);
                    W1t[0] = mma(tr_frag_pi(Kt, TS, 32 * ti, s, 0, l), gz, W1t[0]);
                    W1t[1] = mma(tr_frag_pi(Kt, TS, 32 * ti, s, 32, l), gz, W1t[1]);
                }
            }
            b1v += xor_add(sb, 32);   // b1' = b1 - sum_t eta gZ1
        }
        // ---- f6, T form: Z1b = Q W1' + b1' (rows=t, lane=n; stored) ; X2b = gelu ---------------------
        bf16x8 X2bF[2][2];            // [ti][s]  T fragments of X2b
        {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W1F[a][s] = pack(W1t[a], s);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 zb = zero16();
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        zb = mma(pi_read(Qt + (32 * ti + c) * TS, 32 * a, s, h), W1F[a][s], zb);
#pragma unroll
                for (int r = 0; r < 16; ++r) zb[r] += b1v;
                st_frag4<NTS>(slice, A_Z1B, fr_idx(ti, pp, 0), pack(zb, 0), l);
                st_frag4<NTS>(slice, A_Z1B, fr_idx(ti, pp, 1), pack(zb, 1), l);
                gelu_fwd_tile_pk(zb);
                X2bF[ti][0] = pack(zb, 0);
                X2bF[ti][1] = pack(zb, 1);
            }
        }
        __syncthreads();              // B3: every read of the X2 image, of Kt and of Vt / etaL is done
        if (more) {
            const size_t off = (tile + 1) * 4096 + (size_t)prow * 64 + pcol;
            pfK = *reinterpret_cast<const uint4*>(p.XK + off);
            pfV = *reinterpret_cast<const uint4*>(p.XV + off);
            pfQ = *reinterpret_cast<const uint4*>(p.XQ + off);
            pfE = reinterpret_cast<const unsigned short*>(p.eta)[(tile + 1) * 64 + (tid & 63)];
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int s = 0; s < 2; ++s) st_image(X2img + (nO + c) * TS, 32 * ti, s, h, X2bF[ti][s]);      // X2b image [n][t]
        if (w == 0 && h == 0) b2L[fO + c] = b2v;
        __syncthreads();              // B4: X2b image, b2' visible

        // ================= E: partial Z2b^T[Fp, t] ================================================
        {
            bf16x8 W2F[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W2F[a][s] = pack(W2t[a], s);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 P = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    P = mma(W2F[0][s], tr_frag_pi(X2img, TS, nO, s, 32 * ti, l), P);
                    P = mma(W2F[1][s], tr_frag_pi(X2img, TS, nX, s, 32 * ti, l), P);
                }
                write_partial2(red + (size_t)w * 64 * PS, P, ti, pp, h, c);
            }
        }
        // the state after the LAST step of the sequence anchors the sweep of the topmost chunk (stored from inside the loop: a use
        // of the state tiles behind the loop makes hipcc spill 32 registers per step)
        if (i + 1 == NC) {
            float* W1g = p.wfinal + (size_t)bh * FINAL_FLOATS;
            float* W2g = W1g + 64 * 256;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = row_of(r, h);
                W1g[(size_t)ro * 256 + nO + c] = W1t[0][r];
                W1g[(size_t)(32 + ro) * 256 + nO + c] = W1t[1][r];
                W2g[(size_t)(nO + ro) * 64 + fO + c] = W2t[0][r];
                W2g[(size_t)(nX + ro) * 64 + fO + c] = W2t[1][r];
            }
        }
        if (more) {                   // next step's K, V, eta (their last readers finished before B3)
            *reinterpret_cast<uint4*>(Kt + prow * TS + pcol) = pfK;
            *reinterpret_cast<uint4*>(Vt + prow * TS + pcol) = pfV;
            unsigned pfEu = pfE;
            asm volatile("" : "+v"(pfEu));     // every wave consumes its load HERE (left to the compiler the conversion sinks into the branch below,
                                              // the register stays pending in the other waves, and its pairing with b1v in A1 waits vmcnt(0) there)
            const float pfEf = __builtin_bit_cast(float, pfEu << 16);
            if (tid < 64) etaL[tid] = pfEf;
        }
        __syncthreads();              // B5

        // ================= P6: owners - reduce, output LayerNorm statistics -> x_hat rows (stored) ==
        {
            float z[8];
            load8_f32(b2L + of0, z);
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) add8_f32(red + ((size_t)ww * 64 + ot) * PS + of0, z);
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += z[j];
            const float mu = sum8(s) * (1.0f / 64.0f);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = z[j] - mu; v += d * d; }
            const float rstd = __builtin_amdgcn_rsqf(sum8(v) * (1.0f / 64.0f) + p.eps);
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (z[j] - mu) * rstd;
            st_rows<NTS, 8>(own, 2, ot, of0, z);
            if ((tid & 7) == 0) own_stats(own, ot)[1] = rstd;
        }
    }
}

void launch_recompute4(const RecomputeParams& p, int n_bh, int max_workgroups, hipStream_t s) {
    static ttt::OncePerDevice done;
    done.run([&] {
        (void)hipFuncSetAttribute((const void*)mlp_recompute8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_RC4);
        (void)hipFuncSetAttribute((const void*)mlp_recompute8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_RC4);
    });
    RecomputeParams q = p;
    const int n_items = n_bh * p.chunk_groups;
    const int per = (max_workgroups > 0 && max_workgroups < n_items) ? max_workgroups : n_items;
    for (q.item0 = 0; q.item0 < n_items; q.item0 += per) {
        const int grid = n_items - q.item0 < per ? n_items - q.item0 : per;
        if (q.nt) hipLaunchKernelGGL(mlp_recompute8_kernel<true>, dim3(grid), dim3(NT8), LDS_RC4, s, q);
        else hipLaunchKernelGGL(mlp_recompute8_kernel<false>, dim3(grid), dim3(NT8), LDS_RC4, s, q);
    }
}

}  // namespace s4
}  // namespace mfma
}  // namespace ttt
