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
// computed in both kernels (7 GEMM units instead of 5).  Round 5 built the single-pass form (commit 1fb5d95: dS through an LDS
// image to the waves that contract it over the keys, the dQ blocks of the 73 key blocks added to an fp32 accumulator by
// global_atomic_add_f32) and measured it at 48 heads x S = 18 048 (profiles/r5a_attn_single_pass_ab.json): parity-green and
// 25.6 ms against 13.0 ms for this pair - the 32.5 GB of atomic traffic alone cost 14.4 ms (2.3 TB/s: gfx950 executes them at
// the memory side, the lines are dropped from L2), the kernel without them 11.1 ms, without its dQ phase 8.6 ms.  Removed.  Math = autograd of the reference's F.scaled_dot_product_attention (cogvideo/dit.py:196-198).
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

// dK / dV and dQ: the workgroup bodies of attn_body.h (emulator-checked against the fp64 oracle on the CPU), instantiated in
// attn_v2.hip.  Round-2 A/B on an MI355X (48 heads, S = 18 048): revision-1 kernels 14.97 ms per backward; dQ through the body
// (tail mask behind a wave-uniform branch) 14.83; + dK / dV with accumulator-initialised row scalars and 12 waves 14.27.  The
// revision-1 dK / dV and dQ kernels, and dK / dV variants 2 / 3, were removed.
void launch_backward(const BwdParams& p, hipStream_t s) {
    const long rows = (long)p.B * p.NH * p.S;
    const int dgrid = (int)((rows * 8 + 255) / 256 < 65536 ? (rows * 8 + 255) / 256 : 65536);
    hipLaunchKernelGGL(attn_delta_kernel, dim3(dgrid), dim3(256), 0, s, p);
    launch_dkdv_v2(p, s);
    launch_dq_v2(p, s);
}

}  // namespace attn
}  // namespace ttt
