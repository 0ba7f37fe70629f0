// Segment self-attention forward for gfx950 (non-causal flash attention, head_dim 64, bf16 in / fp32 accumulate).
//
// Replaces F.scaled_dot_product_attention of the reference's local attention (ttt/models/cogvideo/dit.py:196-198):
//   O = softmax(Q K^T / sqrt(64)) V   over one 3-second segment, S ~ 18 k tokens, 48 heads.
//
// Geometry.  A workgroup = 8 waves = 256 query rows of one (batch, head); a wave owns 32 query rows for the whole
// key loop.  Keys / values stream through LDS in tiles of 64 (double-buffered, register-staged: the loads of tile
// j+1 are issued before the MFMAs of tile j and parked after them, one barrier per tile).
// Layout algebra (ttt_mfma_dev.h): the scores are computed TRANSPOSED,  S^T[key, q] = K Q^T  (A = K rows from LDS,
// B = Q rows held in registers for the whole loop), so a lane owns ONE query row (its lane index) and 32 of the 64
// keys of the tile in registers: row max / row sum are register reductions plus one exchange between the two
// half-waves, the softmax never crosses lanes otherwise.  The packed probabilities are then, in place, the B
// operand of  O^T[d, q] = V^T P^T  (contraction over the tile's row index = key), with V^T fragments read by
// ds_read_b64_tr_b16 from the row-major V tile.  O^T keeps one query per lane too, so the online-softmax rescale is a
// per-lane scalar multiply.
// Workgroup -> (head, query block) mapping is XCD-aware: all query blocks of a head run on the same XCD (blockIdx % 8),
// so the head's K and V (2 x 2.3 MB at S = 18 k) are fetched from HBM once and then served by that XCD's L2.
#include <hip/hip_runtime.h>
#include "../../include/ttt_hip.h"
#include "attn.h"
#include "attn_dev.h"
#include "once_per_device.h"

namespace ttt {
namespace attn {

constexpr int QB = 256;          // query rows per workgroup
constexpr int KB = 64;           // keys per tile
constexpr int NTF = 512;
// K tile: rows read as 16-byte fragments -> stride 72 elements (36 dwords: conflict-free ds_read_b128).  V tile: read only
// through ds_read_b64_tr_b16, whose 32-lane groups cover 4 rows x 16 dwords -> stride 96 elements (48 dwords) puts the 4
// rows on disjoint bank quarters; with stride 72 rows r and r+2 overlap (measured 28 % of the LDS cycles were conflicts).
constexpr int VS = 96;
constexpr int KT_ELEMS = ATILE, VT_ELEMS = 64 * VS, BUF_ELEMS = KT_ELEMS + VT_ELEMS;
constexpr int LDS_FWD = 2 * BUF_ELEMS * 2;    // 2 buffers x (K [64][72], V [64][96]) bf16

struct KVStage {
    uint4 k, v;
};

__device__ __forceinline__ void stage_issue(KVStage& st, const __bf16* Kp, const __bf16* Vp, long k_ss, long v_ss, int kv0, int S, int tid) {
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
__device__ __forceinline__ void stage_park(const KVStage& st, __bf16* Kt, __bf16* Vt, int tid) {
    const int row = tid >> 3, col = (tid & 7) * 8;
    *reinterpret_cast<uint4*>(Kt + row * AS + col) = st.k;
    *reinterpret_cast<uint4*>(Vt + row * VS + col) = st.v;
}

__global__ __launch_bounds__(NTF) void attn_fwd_kernel(FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* lds = reinterpret_cast<__bf16*>(smem);

    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = tid & 63, h = l >> 5, c = l & 31;

    // XCD-aware mapping: blocks b, b+8, b+16, ... share an XCD; give each XCD whole heads.
    const int nqb = (p.S + QB - 1) / QB;
    int bh, qb;
    {
        const int nbh = p.B * p.NH;
        const int b = blockIdx.x;
        if ((nbh & 7) == 0) {
            const int xcd = b & 7, idx = b >> 3;
            bh = xcd + 8 * (idx / nqb);
            qb = idx % nqb;
        } else {
            bh = b / nqb;
            qb = b % nqb;
        }
    }
    const int bb = bh / p.NH, hh = bh % p.NH;
    const __bf16* Qp = p.Q + (long)bb * p.q_sb + (long)hh * p.q_sh;
    const __bf16* Kp = p.K + (long)bb * p.k_sb + (long)hh * p.k_sh;
    const __bf16* Vp = p.V + (long)bb * p.v_sb + (long)hh * p.v_sh;
    __bf16* Op = p.O + (long)bb * p.o_sb + (long)hh * p.o_sh;

    const int q0 = qb * QB + 32 * wv;          // this wave's first query row
    const int qrow = q0 + c;                   // this lane's query row
    const bool qvalid = qrow < p.S;

    // Q fragments (B operand: lane = query, 8 contiguous d per k-slice), kept for the whole loop
    bf16x8 Qf[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        if (qvalid) Qf[kk] = *reinterpret_cast<const bf16x8*>(Qp + (long)qrow * p.q_ss + 16 * kk + 8 * h);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) Qf[kk][e] = (__bf16)0.0f;
        }
    }

    f32x16 O[2] = {zero16(), zero16()};        // O^T tiles: rows = d (32 db + row_of(r,h)), lane = query
    float m = -1e30f, lsum = 0.f;              // running max (raw score units) and this half-wave's partial row sum
    const float sc = p.scale * 1.4426950408889634f;

    const int nt = (p.S + KB - 1) / KB;
    KVStage st;
    stage_issue(st, Kp, Vp, p.k_ss, p.v_ss, 0, p.S, tid);
    stage_park(st, lds, lds + KT_ELEMS, tid);
    __syncthreads();

    for (int j = 0; j < nt; ++j) {
        const __bf16* Kt = lds + (j & 1) * BUF_ELEMS;
        const __bf16* Vt = Kt + KT_ELEMS;
        const bool more = j + 1 < nt;
        if (more) stage_issue(st, Kp, Vp, p.k_ss, p.v_ss, (j + 1) * KB, p.S, tid);

        // ---- S^T = K Q^T : two key blocks of 32 ----------------------------------------------------------------
        f32x16 Sc[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            f32x16 acc = zero16();
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = mma(row_frag(Kt, AS, 32 * kb, 16 * kk, l), Qf[kk], acc);
            Sc[kb] = acc;
        }
        if (!more && (p.S & (KB - 1))) {        // ragged last tile: keys >= S are masked
            const int kv0 = j * KB;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + 32 * kb + row_of(r, h) >= p.S) Sc[kb][r] = -1e30f;
        }
        // ---- online softmax (one query row per lane; the partner half-wave holds the other 32 keys) --------------
        float mt = Sc[0][0];
#pragma unroll
        for (int r = 1; r < 16; ++r) mt = fmaxf(mt, Sc[0][r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, Sc[1][r]);
        mt = fmaxf(mt, half_swap(mt));
        // rescale only when some row of this wave saw a larger maximum (exact: alpha == 1 for every lane otherwise);
        // after the first few tiles that is the rare case, and it saves 33 multiplies + an exp per lane and tile
        if (__builtin_amdgcn_ballot_w64(mt > m) != 0) {
            const float mn = fmaxf(m, mt);
            const float alpha = __builtin_amdgcn_exp2f((m - mn) * sc);
            m = mn;
            lsum *= alpha;
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) O[db][r] *= alpha;
        }
        const float msc = m * sc;
        float ps = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(Sc[kb][r], sc, -msc));
                Sc[kb][r] = e;
                ps += e;
            }
        lsum += ps;
        // ---- O^T += V^T P^T ----------------------------------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 pf = pack(Sc[kb], s);
                O[0] = mma(tr_frag_pi(Vt, VS, 32 * kb, s, 0, l), pf, O[0]);
                O[1] = mma(tr_frag_pi(Vt, VS, 32 * kb, s, 32, l), pf, O[1]);
            }
        if (more) {
            __bf16* Kn = lds + ((j + 1) & 1) * BUF_ELEMS;
            stage_park(st, Kn, Kn + KT_ELEMS, tid);
        }
        __syncthreads();
    }

    // ---- epilogue: normalise, store O[q][d] (4 consecutive d per register group) and the log-sum-exp ----------------
    const float ltot = lsum + half_swap(lsum);
    const float inv = 1.0f / ltot;
    if (qvalid) {
        __bf16* orow = Op + (long)qrow * p.o_ss;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (__bf16)(O[db][4 * g + e] * inv);
                *reinterpret_cast<bf16x4*>(orow + 32 * db + 8 * g + 4 * h) = v;
            }
        if (h == 0 && p.LSE) p.LSE[(long)bh * p.S + qrow] = m * p.scale + __logf(ltot);
    }
}

void launch_forward(const FwdParams& p, hipStream_t s) {
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FWD);
    });
    const int nqb = (p.S + QB - 1) / QB;
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(p.B * p.NH * nqb), dim3(NTF), LDS_FWD, s, p);
}

}  // namespace attn
}  // namespace ttt
