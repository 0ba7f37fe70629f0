// Fused pre-processing of the local attention's queries and keys for gfx950 (HBM-bound, one pass):
// per-head LayerNorm(64, eps) of q and k followed by the real-valued 3-D RoPE on the video tokens of the segment
// (reference ttt/models/cogvideo/dit.py:184-195 with Rotary3DPositionEmbedding.forward, cogvideo/utils.py:424-437).
// The unfused path launches one LayerNorm block per 64-element row (866 k rows per tensor at S = 18 k: ~1.3 ms per call,
// 8 calls + their backward per layer) plus the slice / rotate / concat chain; here a row is 8 lanes x 8 features
// (16 bytes per lane, fully coalesced), row reductions are 3 DPP adds, and the [B, S, NH*64] layout of the projections
// is kept, so the attention kernels consume the result through strides.
// Rounding mirrors the unfused bf16 path: LayerNorm output, the cos / sin tables, both products and their sum are each
// rounded to bf16.  Parameter gradients are per-block partial sums [P, 4, 64] (q weight, q bias, k weight, k bias),
// reduced by the caller: deterministic.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ttt_hip.h"
#include "attn.h"

namespace ttt {
namespace attn {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
__device__ __forceinline__ void ld8(const __bf16* p, float (&o)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
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
__device__ __forceinline__ float rb(float x) { return (float)(__bf16)x; }

// y = bf16(LN(x) * w + b); then, for video tokens, out = bf16(bf16(y*cos) + bf16(rot(y)*sin)), rot = (-y[2i+1], y[2i])
__device__ __forceinline__ void ln_rope(const float (&x)[8], const float (&w)[8], const float (&b)[8], float eps,
                                        const float* cosr, const float* sinr, float (&out)[8], float (&xh)[8], float& rstd) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    const float mean = sum8(s) * (1.0f / 64.0f);
    float vs = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { xh[j] = x[j] - mean; vs += xh[j] * xh[j]; }
    rstd = 1.0f / sqrtf(sum8(vs) * (1.0f / 64.0f) + eps);
#pragma unroll
    for (int j = 0; j < 8; ++j) { xh[j] *= rstd; out[j] = rb(xh[j] * w[j] + b[j]); }
    if (cosr) {
        float c8[8], s8[8];
        ldf8(cosr, c8);
        ldf8(sinr, s8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float a = out[2 * q], bb = out[2 * q + 1];
            out[2 * q] = rb(rb(a * rb(c8[2 * q])) + rb(-bb * rb(s8[2 * q])));
            out[2 * q + 1] = rb(rb(bb * rb(c8[2 * q + 1])) + rb(a * rb(s8[2 * q + 1])));
        }
    }
}

__global__ __launch_bounds__(256) void attn_pre_fwd_kernel(PreParams a) {
    const long total = (long)a.B * a.S * a.NH * 8;
    const int o = threadIdx.x & 7;
    float wq[8], bq[8], wk[8], bk[8];
    ldf8(a.wq + 8 * o, wq); ldf8(a.bq + 8 * o, bq); ldf8(a.wk + 8 * o, wk); ldf8(a.bk + 8 * o, bk);
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int s = ((idx >> 3) / a.NH) % a.S;
        const int pos = s - a.n_text;
        const float* cosr = pos >= 0 ? a.cos + (long)pos * 64 + 8 * o : nullptr;
        const float* sinr = pos >= 0 ? a.sin + (long)pos * 64 + 8 * o : nullptr;
        float x[8], y[8], xh[8], rstd;
        ld8(a.q_raw + idx * 8, x);
        ln_rope(x, wq, bq, a.eps, cosr, sinr, y, xh, rstd);
        st8(a.q + idx * 8, y);
        ld8(a.k_raw + idx * 8, x);
        ln_rope(x, wk, bk, a.eps, cosr, sinr, y, xh, rstd);
        st8(a.k + idx * 8, y);
    }
}

// backward of one tensor: g = dL/d(out) -> dx; accumulates dw, db
__device__ __forceinline__ void ln_rope_bwd(const float (&x)[8], const float (&w)[8], float eps, const float* cosr, const float* sinr,
                                            float (&g)[8], float (&dx)[8], float (&dw)[8], float (&db)[8]) {
    if (cosr) {
        float c8[8], s8[8];
        ldf8(cosr, c8);
        ldf8(sinr, s8);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float ga = g[2 * q], gb = g[2 * q + 1];
            g[2 * q] = ga * rb(c8[2 * q]) + gb * rb(s8[2 * q + 1]);
            g[2 * q + 1] = gb * rb(c8[2 * q + 1]) - ga * rb(s8[2 * q]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x[j];
    const float mean = sum8(s) * (1.0f / 64.0f);
    float vs = 0.f, xh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { xh[j] = x[j] - mean; vs += xh[j] * xh[j]; }
    const float rstd = 1.0f / sqrtf(sum8(vs) * (1.0f / 64.0f) + eps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        xh[j] *= rstd;
        dw[j] += g[j] * xh[j];
        db[j] += g[j];
        g[j] *= w[j];
        s1 += g[j];
        s2 += g[j] * xh[j];
    }
    s1 = sum8(s1) * (1.0f / 64.0f);
    s2 = sum8(s2) * (1.0f / 64.0f);
#pragma unroll
    for (int j = 0; j < 8; ++j) dx[j] = (g[j] - s1 - xh[j] * s2) * rstd;
}

__global__ __launch_bounds__(256) void attn_pre_bwd_kernel(PreBwdParams a) {
    __shared__ float red[4][256][8 + 1];
    const long total = (long)a.B * a.S * a.NH * 8;
    const int o = threadIdx.x & 7;
    float wq[8], wk[8], dwq[8], dbq[8], dwk[8], dbk[8];
    ldf8(a.wq + 8 * o, wq); ldf8(a.wk + 8 * o, wk);
#pragma unroll
    for (int j = 0; j < 8; ++j) { dwq[j] = 0.f; dbq[j] = 0.f; dwk[j] = 0.f; dbk[j] = 0.f; }
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx >> 3;
        const int hd = row % a.NH;
        const long tok = row / a.NH;
        const int s = tok % a.S;
        const int b = tok / a.S;
        const int pos = s - a.n_text;
        const float* cosr = pos >= 0 ? a.cos + (long)pos * 64 + 8 * o : nullptr;
        const float* sinr = pos >= 0 ? a.sin + (long)pos * 64 + 8 * o : nullptr;
        float x[8], g[8], dx[8];
        ld8(a.q_raw + idx * 8, x);
        ld8(a.dq + (long)b * a.dq_sb + (long)hd * a.dq_sh + (long)s * a.dq_ss + 8 * o, g);
        ln_rope_bwd(x, wq, a.eps, cosr, sinr, g, dx, dwq, dbq);
        const long raw_off = tok * a.ld_out + hd * 64 + 8 * o;
        st8(a.dq_raw + raw_off, dx);
        ld8(a.k_raw + idx * 8, x);
        ld8(a.dk + (long)b * a.dk_sb + (long)hd * a.dk_sh + (long)s * a.dk_ss + 8 * o, g);
        ln_rope_bwd(x, wk, a.eps, cosr, sinr, g, dx, dwk, dbk);
        st8(a.dk_raw + raw_off, dx);
    }
    // block reduction over the 32 threads that share a feature octet -> partial [blockIdx][4][64]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[0][threadIdx.x][j] = dwq[j]; red[1][threadIdx.x][j] = dbq[j];
        red[2][threadIdx.x][j] = dwk[j]; red[3][threadIdx.x][j] = dbk[j];
    }
    __syncthreads();
    {
        const int which = threadIdx.x >> 6, f = threadIdx.x & 63;     // 4 x 64 outputs, one per thread
        const int oo = f >> 3, j = f & 7;
        float acc = 0.f;
        for (int t = oo; t < 256; t += 8) acc += red[which][t][j];
        a.part[((long)blockIdx.x * 4 + which) * 64 + f] = acc;
    }
}

int pre_blocks(long rows) {
    const long g = (rows * 8 + 255) / 256;
    return (int)(g < 2048 ? g : 2048);
}
void launch_pre_forward(const PreParams& a, hipStream_t s) {
    const long g = ((long)a.B * a.S * a.NH * 8 + 255) / 256;
    hipLaunchKernelGGL(attn_pre_fwd_kernel, dim3((int)(g < 8192 ? g : 8192)), dim3(256), 0, s, a);
}
void launch_pre_backward(const PreBwdParams& a, hipStream_t s) {
    hipLaunchKernelGGL(attn_pre_bwd_kernel, dim3(pre_blocks((long)a.B * a.S * a.NH)), dim3(256), 0, s, a);
}

}  // namespace attn
}  // namespace ttt
