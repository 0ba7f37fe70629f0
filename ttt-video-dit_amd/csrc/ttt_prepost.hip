// Fused, HBM-bound pre- / post-processing kernels of the TTT layer for gfx950.
//
// They replace the chains of PyTorch elementwise / reduction / copy kernels around the scan:
//   pre  (reference ttt_layer.py:252-306, a4-a8): L2-normalise Q,K per head, 3-D RoPE on video tokens,
//        LayerNorm reconstruction target for V, [B,L,NH,F] -> [B,NH,NC,CS,F] re-layout, and the token
//        permutation (scene interleave, time reversal of the bidirectional pass) - ONE pass: 3 reads, 3 writes.
//   post (ttt_layer.py:327-334, a9): [B,NH,NC,CS,F] -> [B,L,D] re-layout + inverse permutation + post_norm.
//   gate (dit.py:219-222, a18): out = residual + tanh(alpha_text|video) * y.
// Each has a hand-derived backward with the same traffic.  All arithmetic is fp32 on bf16 data; a token's
// 64-feature head row lives in 8 lanes x 8 features (16 bytes per lane, coalesced 128-byte rows), row
// reductions are 3 DPP adds.  Parameter-gradient reductions over tokens are accumulated in registers by
// persistent threads and written as per-block partials [P, ...] (summed by the caller): deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ttt_hip.h"
#include "ttt_prepost.h"

namespace ttt {
namespace prepost {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8(float v) {   // sum over the 8 lanes of an aligned group
    v += dpp_f<0xB1>(v);
    v += dpp_f<0x4E>(v);
    v += dpp_f<0x141>(v);
    return v;
}
__device__ __forceinline__ void ld8(const __bf16* p, float (&o)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)a[j];
}
__device__ __forceinline__ bf16x8 ld8_raw(const __bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ bf16x8 zero8_raw() {
    bf16x8 z;
#pragma unroll
    for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
    return z;
}
__device__ __forceinline__ void cvt8(const bf16x8& a, float (&o)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)a[j];
}
__device__ __forceinline__ void st8(__bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)v[j];
    *reinterpret_cast<bf16x8*>(p) = a;
}
__device__ __forceinline__ void ldf8(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ float bf16_round(float x) { return (float)(__bf16)x; }

constexpr float NORM_EPS = 1e-12f;   // F.normalize default (ttt_layer.py:264-266)
constexpr float TGT_EPS = 1e-8f;     // ln_reconstruction_target (ttt_layer.py:229)

// l2-normalise (+ optional rotation by the (cos, sin) pairs of this lane's 4 feature pairs), rounded to bf16 twice
// exactly like the unfused path (F.normalize output is bf16, the rotation result is bf16)
__device__ __forceinline__ void norm_rope(const float (&x)[8], const float* cs, float (&y)[8], float& inv_n) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
    const float n = fmaxf(sqrtf(sum8(ss)), NORM_EPS);
    inv_n = 1.0f / n;
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = bf16_round(x[j] * inv_n);
    if (cs) {
        float c8[8];
        ldf8(cs, c8);      // (cos0, sin0, cos1, sin1, ...)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = y[2 * q], b = y[2 * q + 1], co = c8[2 * q], si = c8[2 * q + 1];
            y[2 * q] = bf16_round(a * co - b * si);
            y[2 * q + 1] = bf16_round(a * si + b * co);
        }
    }
}

// ------------------------------------------------------------------------------------------------ pre forward
__global__ __launch_bounds__(256) void pre_fwd_kernel(PreArgs a) {
    const int tn = a.tn ? a.tn : a.L;          // (a part of the sequence: positions [t0, t0 + tn))
    const long total = (long)a.B * tn * a.NH * 8;
    const int D = a.NH * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int o = idx & 7;
        const int h = (idx >> 3) % a.NH;
        const long bt = (idx >> 3) / a.NH;
        const int tp = a.t0 + bt % tn;         // position in scan order
        const int b = bt / tn;
        const int src = a.src ? a.src[tp] : tp;
        const int pos = a.pos ? a.pos[tp] : -1;
        const size_t in_off = ((size_t)b * a.L + src) * D + h * 64 + 8 * o;
        const size_t out_off = (((size_t)b * a.NH + h) * a.L + tp) * 64 + 8 * o;   // [B,NH,NC,CS,F] == [B,NH,L,F]
        const float* cs = pos >= 0 ? a.rope + ((size_t)pos * 32 + 4 * o) * 2 : nullptr;
        float q[8], k[8], v[8], y[8], inv;
        ld8(a.XQ_raw + in_off, q);
        ld8(a.XK_raw + in_off, k);
        ld8(a.XV_raw + in_off, v);
        norm_rope(q, cs, y, inv);
        st8(a.XQ + out_off, y);
        norm_rope(k, cs, y, inv);
        st8(a.XK + out_off, y);
        // V <- gamma_h * LN_unbiased(V - K) + beta_h + K      (K = the bf16 value just stored)
        float d[8], s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { d[j] = v[j] - y[j]; s += d[j]; }
        const float mean = sum8(s) * (1.0f / 64.0f);
        float vs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { d[j] -= mean; vs += d[j] * d[j]; }
        const float inv_s = 1.0f / (sqrtf(sum8(vs) * (1.0f / 63.0f)) + TGT_EPS);
        float g[8], be[8];
        ldf8(a.ln_w + h * 64 + 8 * o, g);
        ldf8(a.ln_b + h * 64 + 8 * o, be);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g[j] * (d[j] * inv_s) + be[j] + y[j];
        st8(a.XV + out_off, v);
    }
}

// ------------------------------------------------------------------------------------------------ pre backward
// gradient through y = rope(bf16(x / max(|x|, eps))) given g = dL/dy: rotate back, then the normalisation Jacobian
__device__ __forceinline__ void norm_rope_bwd(const float (&x)[8], const float* cs, float (&g)[8], float (&dx)[8]) {
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
    const float nrm = sqrtf(sum8(ss));
    const float inv_n = 1.0f / fmaxf(nrm, NORM_EPS);
    if (cs) {
        float c8[8];
        ldf8(cs, c8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float ga = g[2 * q], gb = g[2 * q + 1], co = c8[2 * q], si = c8[2 * q + 1];
            g[2 * q] = ga * co + gb * si;
            g[2 * q + 1] = -ga * si + gb * co;
        }
    }
    float dot = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) dot += g[j] * x[j];
    dot = sum8(dot) * inv_n * inv_n;        // (y . g) / n  with y = x / n
    const float keep = nrm > NORM_EPS ? 1.0f : 0.0f;   // clamp_min: no norm gradient below eps
#pragma unroll
    for (int j = 0; j < 8; ++j) dx[j] = (g[j] - keep * x[j] * dot) * inv_n;
}

__global__ __launch_bounds__(256) void pre_bwd_kernel(PreBwdArgs a) {
    // threads are persistent; total thread count is a multiple of NH*8 so a thread keeps its (head, octet)
    const long total = (long)a.B * a.L * a.NH * 8;
    const int D = a.NH * 64;
    const long tid0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;
    const int o = tid0 & 7, h = (tid0 >> 3) % a.NH;
    float dgam[8], dbet[8], g8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dgam[j] = 0.f; dbet[j] = 0.f; }
    ldf8(a.ln_w + h * 64 + 8 * o, g8);
    for (long idx = tid0; idx < total; idx += nthreads) {
        const long bt = (idx >> 3) / a.NH;
        const int tp = bt % a.L, b = bt / a.L;
        const int src = a.src ? a.src[tp] : tp;
        const int pos = a.pos ? a.pos[tp] : -1;
        const size_t in_off = ((size_t)b * a.L + src) * D + h * 64 + 8 * o;
        const size_t out_off = (((size_t)b * a.NH + h) * a.L + tp) * 64 + 8 * o;
        const float* cs = pos >= 0 ? a.rope + ((size_t)pos * 32 + 4 * o) * 2 : nullptr;
        float q[8], k[8], v[8], kb[8], inv, gq[8], gk[8], gv[8], dx[8];
        ld8(a.XQ_raw + in_off, q);
        ld8(a.XK_raw + in_off, k);
        ld8(a.XV_raw + in_off, v);
        ld8(a.dXQ + out_off, gq);
        ld8(a.dXK + out_off, gk);
        ld8(a.dXV + out_off, gv);
        // ---- V path: vout = gamma * dn + beta + kb, dn = (d - mean) / (std + eps), d = v - kb
        norm_rope(k, cs, kb, inv);
        float d[8], s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { d[j] = v[j] - kb[j]; s += d[j]; }
        const float mean = sum8(s) * (1.0f / 64.0f);
        float vs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { d[j] -= mean; vs += d[j] * d[j]; }
        const float sd = sqrtf(sum8(vs) * (1.0f / 63.0f));
        const float inv_s = 1.0f / (sd + TGT_EPS);
        float s1 = 0.f, s2 = 0.f, ddn[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dn = d[j] * inv_s;
            dgam[j] += gv[j] * dn;
            dbet[j] += gv[j];
            ddn[j] = gv[j] * g8[j];
            s1 += ddn[j];
            s2 += ddn[j] * d[j];
        }
        s1 = sum8(s1) * (1.0f / 64.0f);
        s2 = sum8(s2) * inv_s * inv_s / (63.0f * fmaxf(sd, 1e-30f));
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float dd = (ddn[j] - s1) * inv_s - d[j] * s2;
            dx[j] = dd;                     // dV_raw
            gk[j] += gv[j] - dd;            // kb receives the residual path of vout minus the (v - kb) path
        }
        const size_t raw_off = ((size_t)b * a.L + src) * a.ld_out + h * 64 + 8 * o;      // (ld_out = D: the inputs' own layout)
        st8(a.dXV_raw + raw_off, dx);
        norm_rope_bwd(k, cs, gk, dx);
        st8(a.dXK_raw + raw_off, dx);
        norm_rope_bwd(q, cs, gq, dx);
        st8(a.dXQ_raw + raw_off, dx);
    }
    // partials [P][NH*64], P = nthreads / (NH*8)
    const long prow = tid0 / ((long)a.NH * 8);
    float* pg = a.dlnw_part + prow * D + h * 64 + 8 * o;
    float* pb = a.dlnb_part + prow * D + h * 64 + 8 * o;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pg[j] = dgam[j]; pb[j] = dbet[j]; }
}

// ------------------------------------------------------------------------------------------------ post (LayerNorm over D)
// one block per token (blockDim = D/8 threads, each 8 features of one head); persistent over tokens
__device__ __forceinline__ float block_sum(float v, float* sh, int nw) {
    v = sum8(v);
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += sh[i];
    return t;
}

// ---- LayerNorm-over-D backward kernels: token teams ----------------------------------------------------------------------
// A block of 8 waves is split into TEAMS of wt waves (wt = 1, 2, 4: the smallest that covers D / 8 / LN_E threads); a team works
// on one token at a time and each of its threads holds LN_E chunks of 8 features (chunk c = e * team_threads + t, so every e is a
// coalesced sweep over the row).  Why: one block of D / 8 threads per token with 8 features per thread spends ~70 instructions per
// feature on reductions, index arithmetic and conversions and is VALU-issue-bound at 1.3 - 2 TB/s; 24 features per thread cut
// that threefold, the three row reductions of a token cost one barrier each (three LDS buffers, written in turn), and the next
// token's rows are in flight while the current one is reduced.  All teams of a block run the same number of iterations (tokens
// past the end are masked), so the barriers are block-wide.
constexpr int LN_E = 3;
constexpr int LN_BLOCK = 512;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// (a __syncthreads() carries a fence that waits for ALL outstanding memory operations - it would pull the prefetched rows of the
// next token into every reduction; these barriers order LDS accesses only)

// team totals of N per-thread values; buf = float[8 * N] of this reduction round
template <int N>
__device__ __forceinline__ void team_sum(float (&v)[N], float* buf, int wt, int team) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float x = sum8(v[k]);
        x += __shfl_xor(x, 8, 64);
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        v[k] = x;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) buf[wave * N + k] = v[k];
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float t = 0.f;
        for (int i = 0; i < wt; ++i) t += buf[(team * wt + i) * N + k];
        v[k] = t;
    }
}
// The rows in flight have landed - said to the COMPILER: the wait sits here, in front of the next token's loads, and the registers
// leave the asm as plain values.  (Left to itself hipcc hoists the next loads above the first use of this token's data and then
// has to wait for both: the prefetch never overlaps anything.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void settle(bf16x8 (&a)[LN_E], bf16x8 (&b)[LN_E], int& c) {
    static_assert(LN_E == 3, "operand list below");
    u32x4 r[2 * LN_E];
#pragma unroll
    for (int k = 0; k < LN_E; ++k) { r[k] = __builtin_bit_cast(u32x4, a[k]); r[LN_E + k] = __builtin_bit_cast(u32x4, b[k]); }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(c) :: "memory");
#pragma unroll
    for (int k = 0; k < LN_E; ++k) { a[k] = __builtin_bit_cast(bf16x8, r[k]); b[k] = __builtin_bit_cast(bf16x8, r[LN_E + k]); }
}
// block-level sum of a per-thread accumulator array over the teams -> global row `dst` [D]; dyn = float[nt * D]
__device__ __forceinline__ void teams_to_global(const float (&acc)[LN_E][8], const bool (&act)[LN_E], const int (&off)[LN_E], float* dyn,
                                                int team, int nt, int D, float* dst) {
    lds_barrier();                                   // the previous array's readers are done
#pragma unroll
    for (int e = 0; e < LN_E; ++e)
        if (act[e]) {
            f32x4 lo = {acc[e][0], acc[e][1], acc[e][2], acc[e][3]}, hi = {acc[e][4], acc[e][5], acc[e][6], acc[e][7]};
            *reinterpret_cast<f32x4*>(dyn + (size_t)team * D + off[e]) = lo;
            *reinterpret_cast<f32x4*>(dyn + (size_t)team * D + off[e] + 4) = hi;
        }
    lds_barrier();
    for (int c = threadIdx.x * 4; c < D; c += LN_BLOCK * 4) {
        f32x4 t = *reinterpret_cast<const f32x4*>(dyn + c);
        for (int j = 1; j < nt; ++j) {
            const f32x4 u = *reinterpret_cast<const f32x4*>(dyn + (size_t)j * D + c);
            t[0] += u[0]; t[1] += u[1]; t[2] += u[2]; t[3] += u[3];
        }
        *reinterpret_cast<f32x4*>(dst + c) = t;
    }
}
static int ln_team_waves(int D) {                    // host: waves per team
    const int need = (D / 8 + LN_E - 1) / LN_E;
    int wt = 1;
    while (wt * 64 < need) wt *= 2;
    return wt;                                       // D <= 4096 (checked by the C API): wt <= 4
}

__global__ void post_fwd_kernel(PostArgs a) {
    __shared__ float sh[16];
    const int D = a.NH * 64, nw = blockDim.x >> 6;     // blockDim = NH*8 rounded up to whole waves
    const int h = threadIdx.x >> 3, o = threadIdx.x & 7;
    const bool act = h < a.NH;
    float w8[8], b8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w8[j] = 0.f; b8[j] = 0.f; }
    if (act) { ldf8(a.w + h * 64 + 8 * o, w8); ldf8(a.b + h * 64 + 8 * o, b8); }
    const int tn = a.tn ? a.tn : a.L;          // (a part of the sequence: positions [t0, t0 + tn))
    for (long bt = blockIdx.x; bt < (long)a.B * tn; bt += gridDim.x) {
        const int tp = a.t0 + bt % tn, b = bt / tn;
        const int src = a.src ? a.src[tp] : tp;
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = 0.f;
        if (act) ld8(a.Y + (((size_t)b * a.NH + h) * a.L + tp) * 64 + 8 * o, y);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += y[j];
        const float mean = block_sum(s, sh, nw) / D;
        float vs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { y[j] = act ? y[j] - mean : 0.f; vs += y[j] * y[j]; }
        const float rstd = 1.0f / sqrtf(block_sum(vs, sh, nw) / D + a.eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = y[j] * rstd * w8[j] + b8[j];
        if (act) st8(a.out + ((size_t)b * a.L + src) * D + h * 64 + 8 * o, y);
    }
}

// Token teams (above).  Block blk works on tokens (it * gridDim + blk) * nt + team; the token map `src` is read one iteration
// further ahead than the rows (its latency must not sit in front of their loads).  Partials: one row per block, as before.
__global__ __launch_bounds__(LN_BLOCK) void post_bwd_kernel(PostBwdArgs a) {
    constexpr int E = LN_E;
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ float rbuf[3][16];
    const int D = a.NH * 64, chunks = D >> 3;
    const int wt = a.wt, nt = 8 / wt, tthreads = 64 * wt;
    const int team = threadIdx.x / tthreads, t = threadIdx.x % tthreads;
    const unsigned n_tok = (unsigned)a.B * a.L, uL = a.L, per_it = gridDim.x * nt;       // B * L < 2^31
    const unsigned n_it = (n_tok + per_it - 1) / per_it;
    bool act[E];
    int off[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const int c = e * tthreads + t; act[e] = c < chunks; off[e] = act[e] ? c * 8 : 0; }
    float w8[E][8], dw[E][8], db[E][8];
#pragma unroll
    for (int e = 0; e < E; ++e) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { w8[e][j] = 0.f; dw[e][j] = 0.f; db[e][j] = 0.f; }
        if (act[e]) ldf8(a.w + off[e], w8[e]);
    }
    auto token_of = [&](unsigned it) { return (it * gridDim.x + blockIdx.x) * nt + team; };
    bf16x8 yr[E], gr[E];
    int srow = 0;                                    // token (row of dOut within its batch) of the position AFTER the one in flight
    auto load_row = [&](unsigned it) {
        const unsigned bt = token_of(it);
        srow = 0;
        if (it < n_it && bt < n_tok) {               // (nothing is computed from the loaded value here: that would wait for it)
            const unsigned tp = bt % uL;
            if (a.src) srow = a.src[tp];
            else srow = (int)tp;
        }
    };
    auto load = [&](unsigned it) {
        const unsigned bt = token_of(it);
        const unsigned tp = bt % uL, b = bt / uL;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            yr[e] = zero8_raw();
            gr[e] = zero8_raw();
            if (act[e] && bt < n_tok) {
                const int h = off[e] >> 6, o8 = off[e] & 63;
                yr[e] = ld8_raw(a.Y + (((size_t)b * a.NH + h) * a.L + tp) * 64 + o8);
                gr[e] = ld8_raw(a.dOut + ((size_t)b * a.L + srow) * D + off[e]);
            }
        }
    };
    if (n_it > 0) {
        load_row(0);
        load(0);
        load_row(1);
    }
    for (unsigned it = 0; it < n_it; ++it) {
        const unsigned bt = token_of(it);
        const bool valid = bt < n_tok;
        float y[E][8], g[E][8];
        settle(yr, gr, srow);
#pragma unroll
        for (int e = 0; e < E; ++e) { cvt8(yr[e], y[e]); cvt8(gr[e], g[e]); }
        if (it + 1 < n_it) { load(it + 1); load_row(it + 2); }
        float r[1] = {0.f};
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[0] += y[e][j];
        team_sum<1>(r, rbuf[0], wt, team);
        const float mean = r[0] / D;
        r[0] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) { y[e][j] = act[e] ? y[e][j] - mean : 0.f; r[0] += y[e][j] * y[e][j]; }
        team_sum<1>(r, rbuf[1], wt, team);
        const float rstd = 1.0f / sqrtf(r[0] / D + a.eps);
        float r2[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                y[e][j] *= rstd;                     // x_hat
                if (valid) { dw[e][j] += g[e][j] * y[e][j]; db[e][j] += g[e][j]; }
                g[e][j] *= w8[e][j];
                r2[0] += g[e][j];
                r2[1] += g[e][j] * y[e][j];
            }
        team_sum<2>(r2, rbuf[2], wt, team);
        if (valid) {
            const float s1 = r2[0] / D, s2 = r2[1] / D;
            const unsigned tp = bt % uL, b = bt / uL;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (act[e]) {
                    float out[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) out[j] = (g[e][j] - s1 - y[e][j] * s2) * rstd;
                    const int h = off[e] >> 6, o8 = off[e] & 63;
                    st8(a.dY + (((size_t)b * a.NH + h) * a.L + tp) * 64 + o8, out);
                }
        }
    }
    teams_to_global(dw, act, off, dyn, team, nt, D, a.dw_part + (size_t)blockIdx.x * D);
    teams_to_global(db, act, off, dyn, team, nt, D, a.db_part + (size_t)blockIdx.x * D);
}

// ------------------------------------------------------------------------------------------------ gate
// out[b,t,:] = res[b,t,:] + tanh(alpha_sel)[:] * y[b,t,:],  alpha_sel = alpha_text for t < n_text else alpha_video
__global__ __launch_bounds__(256) void gate_fwd_kernel(GateArgs a) {
    const int D8 = a.D / 8;
    const long total = (long)a.B * a.L * D8;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = idx % D8;
        const int t = (idx / D8) % a.L;
        const float* al = (t < a.n_text ? a.tanh_text : a.tanh_video) + 8 * c;
        float r[8], y[8], g[8];
        ld8(a.res + idx * 8, r);
        ld8(a.y + idx * 8, y);
        ldf8(al, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += g[j] * y[j];
        st8(a.out + idx * 8, r);
    }
}

// dy = g * tanh(alpha_sel); dtanh partials [P][2][D] (text, video); dres = g is returned by the caller as-is
__global__ __launch_bounds__(256) void gate_bwd_kernel(GateBwdArgs a) {
    const int D8 = a.D / 8;
    const long total = (long)a.B * a.L * D8;
    const long tid0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;   // multiple of D8: a thread keeps its feature octet
    const int c = tid0 % D8;
    float tt[8], tv[8], at[8], av[8];
    ldf8(a.tanh_text + 8 * c, tt);
    ldf8(a.tanh_video + 8 * c, tv);
#pragma unroll
    for (int j = 0; j < 8; ++j) { at[j] = 0.f; av[j] = 0.f; }
    for (long idx = tid0; idx < total; idx += nthreads) {
        const int t = (idx / D8) % a.L;
        float g[8], y[8];
        ld8(a.g + idx * 8, g);
        ld8(a.y + idx * 8, y);
        if (t < a.n_text) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { at[j] += g[j] * y[j]; g[j] *= tt[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { av[j] += g[j] * y[j]; g[j] *= tv[j]; }
        }
        st8(a.dy + idx * 8, g);
    }
    const long prow = tid0 / D8;
    float* pt = a.dtanh_part + (prow * 2 + 0) * a.D + 8 * c;
    float* pv = a.dtanh_part + (prow * 2 + 1) * a.D + 8 * c;
#pragma unroll
    for (int j = 0; j < 8; ++j) { pt[j] = at[j]; pv[j] = av[j]; }
}

// ------------------------------------------------------------------------------------------------ AdaLN (TransformerLayer glue)
// out[b, :, :] = [ modulate(LN(text[b])) | modulate(LN(vid[b])) ],  modulate(y) = shift + y * scale1p  (scale1p = 1 + scale),
// LN over D with (w, b, eps); shift / scale1p are per (batch, group) vectors.  Replaces, per call, two LayerNorm kernels, two
// addcmul kernels and the concat copy of the reference's  modulate(layernorm(vid)), modulate(layernorm(text))  + torch.cat
// (cogvideo/dit.py:353-371) by one pass: one block per token (blockDim = D/8 rounded up to whole waves, 8 features per thread).
// Rounding mirrors the unfused bf16 path: the LayerNorm output is rounded to bf16 before the modulation.
__global__ void adaln_fwd_kernel(AdaLNArgs a) {
    __shared__ float sh[16];
    const int D = a.D, nw = blockDim.x >> 6, L = a.Lt + a.Lv;
    const int o8 = threadIdx.x * 8;
    const bool act = o8 < D;
    float w8[8], b8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w8[j] = 0.f; b8[j] = 0.f; }
    if (act) { ldf8(a.w + o8, w8); ldf8(a.b + o8, b8); }
    for (long bt = blockIdx.x; bt < (long)a.B * L; bt += gridDim.x) {
        const int t = bt % L, b = bt / L;
        const int g = t < a.Lt ? 0 : 1;                                   // 0 = text, 1 = video
        const __bf16* src = g == 0 ? a.text + ((size_t)b * a.Lt + t) * D : a.vid + ((size_t)b * a.Lv + (t - a.Lt)) * D;
        float x[8], sf[8], sc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = 0.f;
        if (act) ld8(src + o8, x);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j];
        const float mean = block_sum(s, sh, nw) / D;
        float vs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { x[j] = act ? x[j] - mean : 0.f; vs += x[j] * x[j]; }
        const float rstd = 1.0f / sqrtf(block_sum(vs, sh, nw) / D + a.eps);
        if (act) {
            ldf8(a.shift + ((size_t)b * 2 + g) * D + o8, sf);
            ldf8(a.scale1p + ((size_t)b * 2 + g) * D + o8, sc);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = sf[j] + bf16_round(x[j] * rstd * w8[j] + b8[j]) * sc[j];
            st8(a.out + ((size_t)b * L + t) * D + o8, x);
        }
    }
}

// backward: d_in = LN'(dOut * scale1p) ; parameter-gradient partials, one row per block:
//   part[blk][0] = dw, [1] = db (LayerNorm), [2] = d scale1p, [3] = d shift  for the (batch, group) the block works on
// (block blk handles batch blk / (2 P), group (blk / P) % 2; the caller reduces over the P blocks of a (batch, group)).
// Token teams (above).  Block pi of a (batch, group) works on tokens (it * P + pi) * nt + team of that group.
__global__ __launch_bounds__(LN_BLOCK) void adaln_bwd_kernel(AdaLNBwdArgs a) {
    constexpr int E = LN_E;
    extern __shared__ __attribute__((aligned(16))) float dyn[];
    __shared__ float rbuf[3][16];
    const int D = a.D, chunks = D >> 3, L = a.Lt + a.Lv, P = a.P;
    const int wt = a.wt, nt = 8 / wt, tthreads = 64 * wt;
    const int team = threadIdx.x / tthreads, t = threadIdx.x % tthreads;
    const int b = blockIdx.x / (2 * P), g = (blockIdx.x / P) % 2, pi = blockIdx.x % P;
    const int n_tok = g == 0 ? a.Lt : a.Lv, t0 = g == 0 ? 0 : a.Lt;
    const int n_it = (n_tok + P * nt - 1) / (P * nt);
    const __bf16* src0 = g == 0 ? a.text + (size_t)b * a.Lt * D : a.vid + (size_t)b * a.Lv * D;
    __bf16* dst0 = g == 0 ? a.dtext + (size_t)b * a.Lt * D : a.dvid + (size_t)b * a.Lv * D;
    const __bf16* dout0 = a.dout + ((size_t)b * L + t0) * D;
    bool act[E];
    int off[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { const int c = e * tthreads + t; act[e] = c < chunks; off[e] = act[e] ? c * 8 : 0; }
    // accumulated over the block's tokens: dsh = sum dOut, gxh = sum dOut * x_hat, dsc = sum dOut * LN(x).  The LayerNorm
    // parameter gradients follow at the end, scale1p being constant over the block: dw = scale1p * gxh, db = scale1p * dsh.
    float gxh[E][8], dsc[E][8], dsh[E][8];
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int j = 0; j < 8; ++j) { gxh[e][j] = 0.f; dsc[e][j] = 0.f; dsh[e][j] = 0.f; }
    // LayerNorm weight / bias and scale1p of this (batch, group): 3 x D floats in LDS for the token loop (72 registers per thread
    // otherwise, which spill); the area is re-used for the team reduction afterwards (the host sizes it for the larger of the two)
    float* cw = dyn;
    float* cb = dyn + D;
    float* cs = dyn + 2 * (size_t)D;
    for (int c = threadIdx.x * 4; c < D; c += LN_BLOCK * 4) {
        *reinterpret_cast<f32x4*>(cw + c) = *reinterpret_cast<const f32x4*>(a.w + c);
        *reinterpret_cast<f32x4*>(cb + c) = *reinterpret_cast<const f32x4*>(a.b + c);
        *reinterpret_cast<f32x4*>(cs + c) = *reinterpret_cast<const f32x4*>(a.scale1p + ((size_t)b * 2 + g) * D + c);
    }
    __syncthreads();
    auto token_of = [&](int it) { return (it * P + pi) * nt + team; };
    bf16x8 xr[E], gr[E];
    int unused = 0;
    auto load = [&](int it) {
        const int tt = token_of(it);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            xr[e] = zero8_raw();
            gr[e] = zero8_raw();
            if (act[e] && tt < n_tok) { xr[e] = ld8_raw(src0 + (size_t)tt * D + off[e]); gr[e] = ld8_raw(dout0 + (size_t)tt * D + off[e]); }
        }
    };
    if (n_it > 0) load(0);
    for (int it = 0; it < n_it; ++it) {
        const int tt = token_of(it);
        const bool valid = tt < n_tok;
        float x[E][8], gy[E][8];
        settle(xr, gr, unused);
#pragma unroll
        for (int e = 0; e < E; ++e) { cvt8(xr[e], x[e]); cvt8(gr[e], gy[e]); }
        if (it + 1 < n_it) load(it + 1);
        float r[1] = {0.f};
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) r[0] += x[e][j];
        team_sum<1>(r, rbuf[0], wt, team);
        const float mean = r[0] / D;
        r[0] = 0.f;
#pragma unroll
        for (int e = 0; e < E; ++e)
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[e][j] = act[e] ? x[e][j] - mean : 0.f; r[0] += x[e][j] * x[e][j]; }
        team_sum<1>(r, rbuf[1], wt, team);
        const float rstd = 1.0f / sqrtf(r[0] / D + a.eps);
        float r2[2] = {0.f, 0.f};
#pragma unroll
        for (int e = 0; e < E; ++e) {
            float w8[8], b8[8], sc[8];
            ldf8(cw + off[e], w8);
            ldf8(cb + off[e], b8);
            ldf8(cs + off[e], sc);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[e][j] *= rstd;                                                     // x_hat
                if (valid) {
                    dsh[e][j] += gy[e][j];
                    gxh[e][j] += gy[e][j] * x[e][j];
                    dsc[e][j] += gy[e][j] * bf16_round(x[e][j] * w8[j] + b8[j]);     // d(scale1p): dOut * LN output
                }
                gy[e][j] *= sc[j] * w8[j];                                           // gradient w.r.t. x_hat
                r2[0] += gy[e][j];
                r2[1] += gy[e][j] * x[e][j];
            }
        }
        team_sum<2>(r2, rbuf[2], wt, team);
        if (valid) {
            const float s1 = r2[0] / D, s2 = r2[1] / D;
#pragma unroll
            for (int e = 0; e < E; ++e)
                if (act[e]) {
                    float out[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) out[j] = (gy[e][j] - s1 - x[e][j] * s2) * rstd;
                    st8(dst0 + (size_t)tt * D + off[e], out);
                }
        }
    }
    float* pr = a.part + (size_t)blockIdx.x * 4 * D;
    float dsh_s[E][8];                               // db = scale1p * dsh, dw = scale1p * gxh (the constants are still in LDS here)
#pragma unroll
    for (int e = 0; e < E; ++e) {
        float sc[8];
        ldf8(cs + off[e], sc);
#pragma unroll
        for (int j = 0; j < 8; ++j) { gxh[e][j] *= sc[j]; dsh_s[e][j] = dsh[e][j] * sc[j]; }
    }
    teams_to_global(gxh, act, off, dyn, team, nt, D, pr);                    // (its first barrier: every thread has read `cs`)
    teams_to_global(dsh_s, act, off, dyn, team, nt, D, pr + D);
    teams_to_global(dsc, act, off, dyn, team, nt, D, pr + 2 * (size_t)D);
    teams_to_global(dsh, act, off, dyn, team, nt, D, pr + 3 * (size_t)D);
}

// ------------------------------------------------------------------------------------------------ gated residual (TransformerLayer glue)
// new_vid = vid + g_v * y[:, Lt:], new_text = text + g_t * y[:, :Lt]  (cogvideo/dit.py:358-359, 372-373); y is [text | video].
__global__ __launch_bounds__(256) void resgate_fwd_kernel(ResGateArgs a) {
    const int D8 = a.D / 8, L = a.Lt + a.Lv;
    const long total = (long)a.B * L * D8;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = idx % D8;
        const int t = (idx / D8) % L;
        const int b = idx / ((long)D8 * L);
        const int g = t < a.Lt ? 0 : 1;
        const size_t ro = g == 0 ? ((size_t)b * a.Lt + t) * a.D + 8 * c : ((size_t)b * a.Lv + (t - a.Lt)) * a.D + 8 * c;
        float r[8], y[8], gt[8];
        ld8((g == 0 ? a.text : a.vid) + ro, r);
        ld8(a.y + idx * 8, y);
        ldf8(a.gate + ((size_t)b * 2 + g) * a.D + 8 * c, gt);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += bf16_round(gt[j] * y[j]);
        st8((g == 0 ? a.otext : a.ovid) + ro, r);
    }
}

// backward: dy = gate * d_out (written as one [text | video] tensor), d gate partials [P][B][2][D]; the residual gradients are
// d_out itself (returned by the caller as-is).  Threads are persistent over tokens with a fixed feature octet.
__global__ __launch_bounds__(256) void resgate_bwd_kernel(ResGateBwdArgs a) {
    const int D8 = a.D / 8, L = a.Lt + a.Lv;
    const long tid0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long nthreads = (long)gridDim.x * blockDim.x;     // multiple of D8
    const int c = tid0 % D8;
    const long prow = tid0 / D8, nrows = nthreads / D8;
    for (int b = 0; b < a.B; ++b) {
        float gt[2][8], acc[2][8];
        ldf8(a.gate + ((size_t)b * 2 + 0) * a.D + 8 * c, gt[0]);
        ldf8(a.gate + ((size_t)b * 2 + 1) * a.D + 8 * c, gt[1]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { acc[0][j] = 0.f; acc[1][j] = 0.f; }
        for (long t = prow; t < L; t += nrows) {
            const int g = t < a.Lt ? 0 : 1;
            const size_t ro = g == 0 ? ((size_t)b * a.Lt + t) * a.D + 8 * c : ((size_t)b * a.Lv + (t - a.Lt)) * a.D + 8 * c;
            const size_t yo = ((size_t)b * L + t) * a.D + 8 * c;
            float d[8], y[8];
            ld8((g == 0 ? a.dtext : a.dvid) + ro, d);
            ld8(a.y + yo, y);
            if (g == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[0][j] += d[j] * y[j]; d[j] *= gt[0][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[1][j] += d[j] * y[j]; d[j] *= gt[1][j]; }
            }
            st8(a.dy + yo, d);
        }
        float* p0 = a.dgate_part + (((size_t)prow * a.B + b) * 2 + 0) * a.D + 8 * c;
        float* p1 = a.dgate_part + (((size_t)prow * a.B + b) * 2 + 1) * a.D + 8 * c;
#pragma unroll
        for (int j = 0; j < 8; ++j) { p0[j] = acc[0][j]; p1[j] = acc[1][j]; }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static int grid_for(long total_threads, int block, int cap_blocks) {
    long g = (total_threads + block - 1) / block;
    return (int)(g < cap_blocks ? g : cap_blocks);
}

void pre_forward(const PreArgs& a, hipStream_t s) {
    const long total = (long)a.B * (a.tn ? a.tn : a.L) * a.NH * 8;
    hipLaunchKernelGGL(pre_fwd_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, a);
}
int pre_backward_partials(int NH) {          // P for a launch of pre_backward
    const int per = NH * 8;                  // threads per (head, octet) period
    int blocks = 2048;
    while ((long)blocks * 256 % per) --blocks;
    return (int)((long)blocks * 256 / per);
}
void pre_backward(const PreBwdArgs& a, hipStream_t s) {
    const int per = a.NH * 8;
    int blocks = 2048;
    while ((long)blocks * 256 % per) --blocks;
    hipLaunchKernelGGL(pre_bwd_kernel, dim3(blocks), dim3(256), 0, s, a);
}
int post_blocks(int B, int L) {
    const long n = (long)B * L;
    return (int)(n < 1024 ? n : 1024);
}
void post_forward(const PostArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(post_fwd_kernel, dim3(post_blocks(a.B, a.tn ? a.tn : a.L)), dim3((a.NH * 8 + 63) / 64 * 64), 0, s, a);
}
void post_backward(const PostBwdArgs& a0, hipStream_t s) {
    PostBwdArgs a = a0;
    const int D = a.NH * 64;
    a.wt = ln_team_waves(D);
    hipLaunchKernelGGL(post_bwd_kernel, dim3(post_blocks(a.B, a.L)), dim3(LN_BLOCK), (size_t)(8 / a.wt) * D * sizeof(float), s, a);
}
void gate_forward(const GateArgs& a, hipStream_t s) {
    const long total = (long)a.B * a.L * (a.D / 8);
    hipLaunchKernelGGL(gate_fwd_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, a);
}
int gate_backward_partials(int D) {
    const int per = D / 8;
    int blocks = 2048;
    while ((long)blocks * 256 % per) --blocks;
    return (int)((long)blocks * 256 / per);
}
void gate_backward(const GateBwdArgs& a, hipStream_t s) {
    const int per = a.D / 8;
    int blocks = 2048;
    while ((long)blocks * 256 % per) --blocks;
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(blocks), dim3(256), 0, s, a);
}


int adaln_blocks(int B, int Lt, int Lv) {
    const long n = (long)B * (Lt + Lv);
    return (int)(n < 2048 ? n : 2048);
}
void adaln_forward(const AdaLNArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(adaln_fwd_kernel, dim3(adaln_blocks(a.B, a.Lt, a.Lv)), dim3((a.D / 8 + 63) / 64 * 64), 0, s, a);
}
int adaln_backward_partials() { return 256; }          // P blocks per (batch, group)
void adaln_backward(const AdaLNBwdArgs& a0, hipStream_t s) {
    AdaLNBwdArgs a = a0;
    a.P = adaln_backward_partials();
    a.wt = ln_team_waves(a.D);
    const int rows = 8 / a.wt > 3 ? 8 / a.wt : 3;          // team reduction [nt][D] / constants [3][D] share the area
    hipLaunchKernelGGL(adaln_bwd_kernel, dim3(a.B * 2 * a.P), dim3(LN_BLOCK), (size_t)rows * a.D * sizeof(float), s, a);
}
void resgate_forward(const ResGateArgs& a, hipStream_t s) {
    const long total = (long)a.B * (a.Lt + a.Lv) * (a.D / 8);
    hipLaunchKernelGGL(resgate_fwd_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, s, a);
}
int resgate_backward_partials(int D) { return gate_backward_partials(D); }
void resgate_backward(const ResGateBwdArgs& a, hipStream_t s) {
    const int per = a.D / 8;
    int blocks = 2048;
    while ((long)blocks * 256 % per) --blocks;
    hipLaunchKernelGGL(resgate_bwd_kernel, dim3(blocks), dim3(256), 0, s, a);
}

}  // namespace prepost
}  // namespace ttt
