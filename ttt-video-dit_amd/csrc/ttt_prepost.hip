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
// The loads of the group in flight have landed - said to the COMPILER: the wait sits here, in front of the next group's
// loads, and the registers leave the asm as plain values.  (Left to itself hipcc hoists the next group's loads above the
// first use of this group's data and then has to wait for both: the prefetch never overlaps anything.)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void settle4(bf16x8 (&a)[4], bf16x8 (&b)[4]) {
    u32x4 r[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) { r[k] = __builtin_bit_cast(u32x4, a[k]); r[4 + k] = __builtin_bit_cast(u32x4, b[k]); }
    asm volatile("s_waitcnt vmcnt(0)"
                 : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7])
                 :: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) { a[k] = __builtin_bit_cast(bf16x8, r[k]); b[k] = __builtin_bit_cast(bf16x8, r[4 + k]); }
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
    const long total = (long)a.B * a.L * a.NH * 8;
    const int D = a.NH * 64;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int o = idx & 7;
        const int h = (idx >> 3) % a.NH;
        const long bt = (idx >> 3) / a.NH;
        const int tp = bt % a.L;               // position in scan order
        const int b = bt / a.L;
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
        st8(a.dXV_raw + in_off, dx);
        norm_rope_bwd(k, cs, gk, dx);
        st8(a.dXK_raw + in_off, dx);
        norm_rope_bwd(q, cs, gq, dx);
        st8(a.dXQ_raw + in_off, dx);
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

// N sums at once (N values per thread, same summation order per value as block_sum): two barriers for all of them.
// `sh` holds N x 16 floats.  The barriers order LDS accesses only and are written as `s_waitcnt lgkmcnt(0); s_barrier`:
// a __syncthreads() carries a fence that waits for ALL outstanding memory operations (vmcnt(0)), which would pull the
// callers' prefetched global loads of the next token group into every reduction.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void block_sum_n(float (&v)[N], float* sh, int nw) {
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float x = sum8(v[k]);
        x += __shfl_xor(x, 8, 64);
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        v[k] = x;
    }
    const int w = threadIdx.x >> 6;
    lds_barrier();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < N; ++k) sh[k * 16 + w] = v[k];
    }
    lds_barrier();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        float t = 0.f;
        for (int i = 0; i < nw; ++i) t += sh[k * 16 + i];
        v[k] = t;
    }
}
// tokens a block of adaln_bwd_kernel works on at once (their loads are issued one group ahead)
constexpr int LN_BWD_T = 4;                  // (settle4 is written for four)
constexpr int LN_BWD_MAX_THREADS = 512;      // 8 features per thread: rows of up to 4096 features (two waves per SIMD, 256 VGPRs each)

__global__ void post_fwd_kernel(PostArgs a) {
    __shared__ float sh[16];
    const int D = a.NH * 64, nw = blockDim.x >> 6;     // blockDim = NH*8 rounded up to whole waves
    const int h = threadIdx.x >> 3, o = threadIdx.x & 7;
    const bool act = h < a.NH;
    float w8[8], b8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w8[j] = 0.f; b8[j] = 0.f; }
    if (act) { ldf8(a.w + h * 64 + 8 * o, w8); ldf8(a.b + h * 64 + 8 * o, b8); }
    for (long bt = blockIdx.x; bt < (long)a.B * a.L; bt += gridDim.x) {
        const int tp = bt % a.L, b = bt / a.L;
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

// One block per token, as the forward.  (The four-tokens-per-iteration form of adaln_bwd_kernel below was measured here too:
// 0.52 ms against 0.46 ms for this one at the 9 s geometry - 1024 blocks of 40 VGPRs hide the latency better than four
// 196-VGPR blocks per CU do, and the token map adds an integer division per token.)
__global__ void post_bwd_kernel(PostBwdArgs a) {
    __shared__ float sh[16];
    const int D = a.NH * 64, nw = blockDim.x >> 6;
    const int h = threadIdx.x >> 3, o = threadIdx.x & 7;
    const bool act = h < a.NH;
    float w8[8], dw[8], db[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dw[j] = 0.f; db[j] = 0.f; w8[j] = 0.f; }
    if (act) ldf8(a.w + h * 64 + 8 * o, w8);
    for (long bt = blockIdx.x; bt < (long)a.B * a.L; bt += gridDim.x) {
        const int tp = bt % a.L, b = bt / a.L;
        const int src = a.src ? a.src[tp] : tp;
        const size_t yoff = (((size_t)b * a.NH + h) * a.L + tp) * 64 + 8 * o;
        float y[8], g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { y[j] = 0.f; g[j] = 0.f; }
        if (act) { ld8(a.Y + yoff, y); ld8(a.dOut + ((size_t)b * a.L + src) * D + h * 64 + 8 * o, g); }
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += y[j];
        const float mean = block_sum(s, sh, nw) / D;
        float vs = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { y[j] = act ? y[j] - mean : 0.f; vs += y[j] * y[j]; }
        const float rstd = 1.0f / sqrtf(block_sum(vs, sh, nw) / D + a.eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            y[j] *= rstd;                    // x_hat
            dw[j] += g[j] * y[j];
            db[j] += g[j];
            g[j] *= w8[j];
            s1 += g[j];
            s2 += g[j] * y[j];
        }
        s1 = block_sum(s1, sh, nw) / D;
        s2 = block_sum(s2, sh, nw) / D;
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (g[j] - s1 - y[j] * s2) * rstd;
        if (act) st8(a.dY + yoff, g);
    }
    if (act) {
        float* pw = a.dw_part + (size_t)blockIdx.x * D + h * 64 + 8 * o;
        float* pb = a.db_part + (size_t)blockIdx.x * D + h * 64 + 8 * o;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pw[j] = dw[j]; pb[j] = db[j]; }
    }
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
// One block per token as in the forward, but LN_BWD_T tokens per iteration: their row reductions share two barriers each
// (three reductions: mean | variance | the two backward sums together), and the next group's loads are in flight while the
// current group is reduced.  (The one-token form spent a token's time in 8 barriers and one exposed HBM round trip: 0.76 ms =
// 1.3 TB/s at the 9 s geometry; this form 0.52 ms.)  Per-token arithmetic and the order in which a block accumulates its
// parameter-gradient partials are those of the one-token form.
__global__ __launch_bounds__(LN_BWD_MAX_THREADS) void adaln_bwd_kernel(AdaLNBwdArgs a) {
    constexpr int T = LN_BWD_T;
    __shared__ float sh[2 * T * 16];
    const int D = a.D, nw = blockDim.x >> 6, L = a.Lt + a.Lv;
    const int o8 = threadIdx.x * 8;
    const bool act = o8 < D;
    const int P = a.P;
    const int b = blockIdx.x / (2 * P), g = (blockIdx.x / P) % 2, pi = blockIdx.x % P;
    const int n_tok = g == 0 ? a.Lt : a.Lv, t0 = g == 0 ? 0 : a.Lt;
    const __bf16* src0 = g == 0 ? a.text + (size_t)b * a.Lt * D : a.vid + (size_t)b * a.Lv * D;
    __bf16* dst0 = g == 0 ? a.dtext + (size_t)b * a.Lt * D : a.dvid + (size_t)b * a.Lv * D;
    const __bf16* dout0 = a.dout + ((size_t)b * L + t0) * D;
    float w8[8], b8[8], sc[8], dw[8], db[8], dsc[8], dsh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { w8[j] = 0.f; b8[j] = 0.f; sc[j] = 0.f; dw[j] = 0.f; db[j] = 0.f; dsc[j] = 0.f; dsh[j] = 0.f; }
    if (act) { ldf8(a.w + o8, w8); ldf8(a.b + o8, b8); ldf8(a.scale1p + ((size_t)b * 2 + g) * D + o8, sc); }
    bf16x8 xr[T], gr[T];                       // the group in flight, as loaded (converted when its turn comes)
    auto load = [&](int tt0) {
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const int tt = tt0 + k * P;
            xr[k] = zero8_raw();
            gr[k] = zero8_raw();
            if (act && tt < n_tok) { xr[k] = ld8_raw(src0 + (size_t)tt * D + o8); gr[k] = ld8_raw(dout0 + (size_t)tt * D + o8); }
        }
    };
    load(pi);
    for (int tt0 = pi; tt0 < n_tok; tt0 += T * P) {
        float x[T][8], gy[T][8];
        settle4(xr, gr);
#pragma unroll
        for (int k = 0; k < T; ++k) { cvt8(xr[k], x[k]); cvt8(gr[k], gy[k]); }
        const int nx = tt0 + T * P;
        if (nx < n_tok) load(nx);
        float r[T];
#pragma unroll
        for (int k = 0; k < T; ++k) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += x[k][j];
            r[k] = s;
        }
        block_sum_n<T>(r, sh, nw);
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const float mean = r[k] / D;
            float vs = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { x[k][j] = act ? x[k][j] - mean : 0.f; vs += x[k][j] * x[k][j]; }
            r[k] = vs;
        }
        block_sum_n<T>(r, sh, nw);
        float rstd[T], r2[2 * T];
#pragma unroll
        for (int k = 0; k < T; ++k) {
            rstd[k] = 1.0f / sqrtf(r[k] / D + a.eps);
            const bool valid = tt0 + k * P < n_tok;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                x[k][j] *= rstd[k];                                              // x_hat
                if (valid) {
                    dsh[j] += gy[k][j];
                    dsc[j] += gy[k][j] * bf16_round(x[k][j] * w8[j] + b8[j]);    // d(scale1p): dOut * LN output
                }
                gy[k][j] *= sc[j];                                               // gradient w.r.t. the LN output
                if (valid) { dw[j] += gy[k][j] * x[k][j]; db[j] += gy[k][j]; }
                gy[k][j] *= w8[j];
                s1 += gy[k][j];
                s2 += gy[k][j] * x[k][j];
            }
            r2[k] = s1;
            r2[T + k] = s2;
        }
        block_sum_n<2 * T>(r2, sh, nw);
#pragma unroll
        for (int k = 0; k < T; ++k) {
            const int tt = tt0 + k * P;
            if (act && tt < n_tok) {
                const float s1 = r2[k] / D, s2 = r2[T + k] / D;
                float out[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] = (gy[k][j] - s1 - x[k][j] * s2) * rstd[k];
                st8(dst0 + (size_t)tt * D + o8, out);
            }
        }
    }
    if (act) {
        float* pr = a.part + (size_t)blockIdx.x * 4 * D + o8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { pr[j] = dw[j]; pr[D + j] = db[j]; pr[2 * D + j] = dsc[j]; pr[3 * D + j] = dsh[j]; }
    }
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
    const long total = (long)a.B * a.L * a.NH * 8;
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
    hipLaunchKernelGGL(post_fwd_kernel, dim3(post_blocks(a.B, a.L)), dim3((a.NH * 8 + 63) / 64 * 64), 0, s, a);
}
void post_backward(const PostBwdArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(post_bwd_kernel, dim3(post_blocks(a.B, a.L)), dim3((a.NH * 8 + 63) / 64 * 64), 0, s, a);
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
    hipLaunchKernelGGL(adaln_bwd_kernel, dim3(a.B * 2 * a.P), dim3((a.D / 8 + 63) / 64 * 64), 0, s, a);
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
