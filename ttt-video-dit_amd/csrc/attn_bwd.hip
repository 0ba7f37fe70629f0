// Segment self-attention backward for gfx950 (head_dim 64, bf16 in / fp32 accumulate), three kernels:
//   delta  : Delta[q] = rowsum(dO * O)                                    (HBM-bound, one pass)
//   dk_dv  : one workgroup per (batch, head, 256 keys); each wave keeps 32 key rows of K and V as register-resident
//            B operands and its dK, dV accumulator tiles for the whole loop over query tiles of 64 (Q, dO, LSE,
//            Delta staged through LDS, double-buffered):
//              S = Q K^T, dP = dO V^T            tiles (rows = query, lane = key)
//              P = exp(S*scale - LSE), dS = P * (dP - Delta)             elementwise, row scalars from LDS
//              dV += P^T dO, dK += dS^T Q        P / dS re-used IN PLACE as A operands (contraction over the tile's
//                                                row index), dO / Q fragments by ds_read_b64_tr_b16
//   dq     : one workgroup per (batch, head, 256 queries), the forward kernel's structure: keys / values stream
//            through LDS, scores are computed transposed (rows = key, lane = query) so LSE / Delta are per-lane
//            scalars, dS^T is re-used in place as the B operand of dQ^T += K^T dS^T.
// LDS: the Q / dO / K tiles are read both as 16-byte row fragments and through ds_read_b64_tr_b16 from ONE stride-72 image.
// A second, stride-96 image for the transposed reads (as in the forward kernel's V tile) removes every bank conflict
// (SQ_LDS_BANK_CONFLICT 23 % -> 0) but measured 1-2.5 % slower here (one more ds_write per tile, larger footprint): not used.
// dk_dv stays at one 8-wave workgroup per CU (189 VGPRs): bounding it to 128 registers or splitting it into 4-wave
// workgroups (3 per CU) spills inside the loop and was 2.5x slower (measured).
// No atomics: dQ, dK, dV are each written by exactly one workgroup (deterministic); the price is that S and dP are
// computed in both kernels (7 GEMM units instead of 5), the usual trade at S ~ 18 k where a dQ atomic stream would be
// 15+ GB per call.  Math = autograd of the reference's F.scaled_dot_product_attention (cogvideo/dit.py:196-198).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "../../include/ttt_hip.h"
#include "attn.h"
#include "attn_dev.h"

namespace ttt {
namespace attn {

constexpr int NTB = 512;
constexpr float LOG2E = 1.4426950408889634f;

// ------------------------------------------------------------------------------------------------------- delta
__global__ __launch_bounds__(256) void attn_delta_kernel(BwdParams p) {
    const long rows = (long)p.B * p.NH * p.S;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < rows * 8; idx += (long)gridDim.x * blockDim.x) {
        const long row = idx >> 3;
        const int o = idx & 7;
        const int s = row % p.S;
        const int bh = row / p.S;
        const int b = bh / p.NH, h = bh % p.NH;
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(p.dO + (long)b * p.do_sb + (long)h * p.do_sh + (long)s * p.do_ss + 8 * o);
        const bf16x8 c = *reinterpret_cast<const bf16x8*>(p.O + (long)b * p.o_sb + (long)h * p.o_sh + (long)s * p.o_ss + 8 * o);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc += (float)a[e] * (float)c[e];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (o == 0) p.Delta[row] = acc;
    }
}

__device__ __forceinline__ void head_of_block(int b, int nblk, int nbh, int& bh, int& blk) {
    if ((nbh & 7) == 0) {                       // XCD-aware: blocks b, b+8, ... share an XCD; whole heads per XCD
        const int xcd = b & 7, idx = b >> 3;
        bh = xcd + 8 * (idx / nblk);
        blk = idx % nblk;
    } else {
        bh = b / nblk;
        blk = b % nblk;
    }
}

// ------------------------------------------------------------------------------------------------------- dK, dV
constexpr int DKV_BUF = 2 * ATILE * 2 + 2 * 64 * 4;      // Q tile, dO tile, lse[64], delta[64]
constexpr int LDS_DKV = 2 * DKV_BUF;

struct QStage {
    uint4 q, d;
    float lse, del;
};
__device__ __forceinline__ void qstage_issue(QStage& st, const BwdParams& p, const __bf16* Qp, const __bf16* dOp, const float* lse,
                                             const float* del, int q0, int tid) {
    const int row = tid >> 3, col = (tid & 7) * 8;
    const int q = q0 + row;
    if (q < p.S) {
        st.q = *reinterpret_cast<const uint4*>(Qp + (long)q * p.q_ss + col);
        st.d = *reinterpret_cast<const uint4*>(dOp + (long)q * p.do_ss + col);
    } else {
        st.q = make_uint4(0, 0, 0, 0);
        st.d = make_uint4(0, 0, 0, 0);
    }
    if (tid < 64) {
        const int qq = q0 + tid;
        st.lse = qq < p.S ? lse[qq] * LOG2E : 1e30f;       // invalid rows: P = exp2(-huge) = 0
        st.del = qq < p.S ? del[qq] : 0.f;
    }
}
__device__ __forceinline__ void qstage_park(const QStage& st, char* buf, int tid) {
    __bf16* Qt = reinterpret_cast<__bf16*>(buf);
    __bf16* Dt = Qt + ATILE;
    float* sm = reinterpret_cast<float*>(buf + 2 * ATILE * 2);
    const int row = tid >> 3, col = (tid & 7) * 8;
    *reinterpret_cast<uint4*>(Qt + row * AS + col) = st.q;
    *reinterpret_cast<uint4*>(Dt + row * AS + col) = st.d;
    if (tid < 64) { sm[tid] = st.lse; sm[64 + tid] = st.del; }
}

__global__ __launch_bounds__(NTB) void attn_dkdv_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, h = l >> 5, c = l & 31;
    const int nkb = (p.S + 255) / 256;
    int bh, kvb;
    head_of_block(blockIdx.x, nkb, p.B * p.NH, bh, kvb);
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    const __bf16* dOp = p.dO + (long)bb * p.do_sb + (long)hh * p.do_sh;
    const float* lse = p.LSE + (long)bh * p.S;
    const float* del = p.Delta + (long)bh * p.S;

    const int key0 = kvb * 256 + 32 * wv;      // this wave's first key
    const int krow = key0 + c;
    bf16x8 Kf[4], Vf[4];                        // B operands: lane = key, 8 contiguous d per k-slice
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (krow < p.S) {
            Kf[kk] = *reinterpret_cast<const bf16x8*>(Kp + (long)krow * p.k_ss + 16 * kk + 8 * h);
            Vf[kk] = *reinterpret_cast<const bf16x8*>(Vp + (long)krow * p.v_ss + 16 * kk + 8 * h);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { Kf[kk][e] = (__bf16)0.0f; Vf[kk][e] = (__bf16)0.0f; }
        }
    }
    f32x16 dK[2] = {zero16(), zero16()}, dV[2] = {zero16(), zero16()};   // tiles (rows = key, lane = d in block db)
    const float sc = p.scale * LOG2E;

    const int nt = (p.S + 63) / 64;
    QStage st;
    qstage_issue(st, p, Qp, dOp, lse, del, 0, tid);
    qstage_park(st, smem, tid);
    __syncthreads();

    for (int j = 0; j < nt; ++j) {
        const char* buf = smem + (j & 1) * DKV_BUF;
        const __bf16* Qt = reinterpret_cast<const __bf16*>(buf);
        const __bf16* Dt = Qt + ATILE;
        const float* lseL = reinterpret_cast<const float*>(buf + 2 * ATILE * 2);
        const float* delL = lseL + 64;
        const bool more = j + 1 < nt;
        if (more) qstage_issue(st, p, Qp, dOp, lse, del, (j + 1) * 64, tid);

#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            f32x16 Sc = zero16(), dP = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                Sc = mma(row_frag(Qt, AS, 32 * qb, 16 * kk, l), Kf[kk], Sc);
                dP = mma(row_frag(Dt, AS, 32 * qb, 16 * kk, l), Vf[kk], dP);
            }
            const f32x16 lseR = rows_from_lds(lseL, 32 * qb, h);
            const f32x16 delR = rows_from_lds(delL, 32 * qb, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(Sc[r], sc, -lseR[r]));
                Sc[r] = pr;
                dP[r] = pr * (dP[r] - delR[r]);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack(Sc, s), df = pack(dP, s);
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    dV[db] = mma(pf, tr_frag_pi(Dt, AS, 32 * qb, s, 32 * db, l), dV[db]);
                    dK[db] = mma(df, tr_frag_pi(Qt, AS, 32 * qb, s, 32 * db, l), dK[db]);
                }
            }
        }
        if (more) qstage_park(st, smem + ((j + 1) & 1) * DKV_BUF, tid);
        __syncthreads();
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

// ------------------------------------------------------------------------------------------------------- dQ
constexpr int LDS_DQ = 2 * 2 * ATILE * 2;

struct KVStage2 {
    uint4 k, v;
};
__device__ __forceinline__ void kv_issue(KVStage2& st, const __bf16* Kp, const __bf16* Vp, long k_ss, long v_ss, int kv0, int S, int tid) {
    const int row = tid >> 3, col = (tid & 7) * 8;
    const int key = kv0 + row;
    if (key < S) {
        st.k = *reinterpret_cast<const uint4*>(Kp + (long)key * k_ss + col);
        st.v = *reinterpret_cast<const uint4*>(Vp + (long)key * v_ss + col);
    } else {
        st.k = make_uint4(0, 0, 0, 0);
        st.v = make_uint4(0, 0, 0, 0);
    }
}
__device__ __forceinline__ void kv_park(const KVStage2& st, __bf16* Kt, __bf16* Vt, int tid) {
    const int row = tid >> 3, col = (tid & 7) * 8;
    *reinterpret_cast<uint4*>(Kt + row * AS + col) = st.k;
    *reinterpret_cast<uint4*>(Vt + row * AS + col) = st.v;
}

// W = waves per SIMD the register allocation is bounded for: 4 (two workgroups per CU, <= 128 VGPRs, no spills: measured
// 6.6 vs 7.5 ms at the 3 s segment) or 2 (one workgroup per CU, 166 VGPRs)
template <int W>
__global__ __launch_bounds__(NTB, W) void attn_dq_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);
    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, h = l >> 5, c = l & 31;
    const int nqb = (p.S + 255) / 256;
    int bh, qb;
    head_of_block(blockIdx.x, nqb, p.B * p.NH, bh, qb);
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    const __bf16* dOp = p.dO + (long)bb * p.do_sb + (long)hh * p.do_sh;

    const int qrow = qb * 256 + 32 * wv + c;
    const bool qvalid = qrow < p.S;
    bf16x8 Qf[4], Df[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (qvalid) {
            Qf[kk] = *reinterpret_cast<const bf16x8*>(Qp + (long)qrow * p.q_ss + 16 * kk + 8 * h);
            Df[kk] = *reinterpret_cast<const bf16x8*>(dOp + (long)qrow * p.do_ss + 16 * kk + 8 * h);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) { Qf[kk][e] = (__bf16)0.0f; Df[kk][e] = (__bf16)0.0f; }
        }
    }
    const float lse2 = qvalid ? p.LSE[(long)bh * p.S + qrow] * LOG2E : 1e30f;
    const float delta = qvalid ? p.Delta[(long)bh * p.S + qrow] : 0.f;
    const float sc = p.scale * LOG2E;
    f32x16 dQ[2] = {zero16(), zero16()};       // dQ^T tiles (rows = d, lane = query)

    const int nt = (p.S + 63) / 64;
    KVStage2 st;
    kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, 0, p.S, tid);
    kv_park(st, lds, lds + ATILE, tid);
    __syncthreads();

    for (int j = 0; j < nt; ++j) {
        const __bf16* Kt = lds + (j & 1) * 2 * ATILE;
        const __bf16* Vt = Kt + ATILE;
        const bool more = j + 1 < nt;
        if (more) kv_issue(st, Kp, Vp, p.k_ss, p.v_ss, (j + 1) * 64, p.S, tid);
        const bool ragged = !more && (p.S & 63);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 Sc = zero16(), dP = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                Sc = mma(row_frag(Kt, AS, 32 * kb, 16 * kk, l), Qf[kk], Sc);
                dP = mma(row_frag(Vt, AS, 32 * kb, 16 * kk, l), Df[kk], dP);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float pr = __builtin_amdgcn_exp2f(__builtin_fmaf(Sc[r], sc, -lse2));
                if (ragged && j * 64 + 32 * kb + row_of(r, h) >= p.S) pr = 0.f;
                dP[r] = pr * (dP[r] - delta);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 df = pack(dP, s);
                dQ[0] = mma(tr_frag_pi(Kt, AS, 32 * kb, s, 0, l), df, dQ[0]);
                dQ[1] = mma(tr_frag_pi(Kt, AS, 32 * kb, s, 32, l), df, dQ[1]);
            }
        }
        if (more) {
            __bf16* Kn = lds + ((j + 1) & 1) * 2 * ATILE;
            kv_park(st, Kn, Kn + ATILE, tid);
        }
        __syncthreads();
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

void launch_backward(const BwdParams& p, hipStream_t s) {
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)attn_dkdv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DKV);
        (void)hipFuncSetAttribute((const void*)attn_dq_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        (void)hipFuncSetAttribute((const void*)attn_dq_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DQ);
        attr = true;
    }
    const long rows = (long)p.B * p.NH * p.S;
    const int dgrid = (int)((rows * 8 + 255) / 256 < 65536 ? (rows * 8 + 255) / 256 : 65536);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(dgrid), dim3(256), 0, s, p);
    const int nb = (p.S + 255) / 256;
    if (get_dkdv_variant() != 1) launch_dkdv_v2(p, get_dkdv_variant(), s);
    else hipLaunchKernelGGL(attn_dkdv_kernel, dim3(p.B * p.NH * nb), dim3(NTB), LDS_DKV, s, p);
    static const int dq_occ = getenv("TTT_ATTN_DQ_OCC") ? atoi(getenv("TTT_ATTN_DQ_OCC")) : 4;     // DEBUG A/B knob
    if (get_attn_variant() == 2) return launch_dq_v2(p, dq_occ, s);
    if (dq_occ == 2) hipLaunchKernelGGL(attn_dq_kernel<2>, dim3(p.B * p.NH * nb), dim3(NTB), LDS_DQ, s, p);
    else hipLaunchKernelGGL(attn_dq_kernel<4>, dim3(p.B * p.NH * nb), dim3(NTB), LDS_DQ, s, p);
}

}  // namespace attn
}  // namespace ttt
