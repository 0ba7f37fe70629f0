// Parameter blocks of the segment-attention kernels.  No HIP dependency: shared by the device code (attn_fwd.hip, attn_bwd.hip,
// attn_v2.hip), the backend-templated kernel bodies (attn_body.h) and the host-side wave emulator of the CPU tests (tests/emul).
#pragma once
#include <stdint.h>

namespace ttt {
namespace attn {

// element (b, h, s, d) of a tensor lives at base + b*sb + h*sh + s*ss + d  (strides in elements, d contiguous)
struct FwdParams {
    const __bf16 *Q, *K, *V;
    __bf16* O;
    float* LSE;                         // [B, NH, S] natural-log sum-exp of the scaled scores (may be null)
    long q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss;
    int B, NH, S;
    float scale;
};
struct BwdParams {
    const __bf16 *Q, *K, *V, *O, *dO;
    const float* LSE;                   // [B, NH, S]
    float* Delta;                       // [B, NH, S] workspace: rowsum(dO * O)
    __bf16 *dQ, *dK, *dV;
    long q_sb, q_sh, q_ss, k_sb, k_sh, k_ss, v_sb, v_sh, v_ss, o_sb, o_sh, o_ss, do_sb, do_sh, do_ss;
    long dq_sb, dq_sh, dq_ss, dk_sb, dk_sh, dk_ss, dv_sb, dv_sh, dv_ss;
    int B, NH, S;
    float scale;
};

}  // namespace attn
}  // namespace ttt
