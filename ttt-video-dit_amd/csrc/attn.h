// Host-side interface of the segment-attention kernels (attn_fwd.hip, attn_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "attn_types.h"

namespace ttt {
namespace attn {

// fused per-head LayerNorm(64) + RoPE of q and k (attn_pre.hip); q_raw / k_raw / q / k are contiguous [B, S, NH, 64]
struct PreParams {
    const __bf16 *q_raw, *k_raw;
    const float *wq, *bq, *wk, *bk;     // [64]
    const float *cos, *sin;             // [n_pos, 64] tables of Rotary3DPositionEmbedding (row = video token index in the segment)
    __bf16 *q, *k;
    int B, S, NH, n_text;
    float eps;
};
struct PreBwdParams {
    const __bf16 *q_raw, *k_raw, *dq, *dk;
    long dq_sb, dq_sh, dq_ss, dk_sb, dk_sh, dk_ss;   // dq / dk are [B, NH, S, 64] views
    const float *wq, *wk, *cos, *sin;
    __bf16 *dq_raw, *dk_raw;
    float* part;                        // [P, 4, 64]: dwq, dbq, dwk, dbk partial sums, P = pre_blocks(B*S*NH)
    int B, S, NH, n_text;
    float eps;
};
int  pre_blocks(long rows);
void launch_pre_forward(const PreParams& a, hipStream_t s);
void launch_pre_backward(const PreBwdParams& a, hipStream_t s);

void launch_forward(const FwdParams& p, hipStream_t s);
void launch_backward(const BwdParams& p, hipStream_t s);

// revision 2 of the forward and dQ kernels (attn_v2.hip, bodies in attn_body.h); debug option "attn_variant": 1 (default) =
// revision 1 everywhere, 2 = revision 2 for the forward and dQ kernels
void set_attn_variant(int v);
int get_attn_variant();
void launch_forward_v2(const FwdParams& p, hipStream_t s);
void launch_dq_v2(const BwdParams& p, int occ, hipStream_t s);
// dK / dV through the body of attn_body.h; debug option "attn_dkdv_variant": 1 (default) = attn_dkdv_kernel of attn_bwd.hip,
// 2 = the same arithmetic through the body, 3 = accumulator-initialised row scalars (8 waves), 4 = ... with 12 waves
void set_dkdv_variant(int v);
int get_dkdv_variant();
void launch_dkdv_v2(const BwdParams& p, int variant, hipStream_t s);

}  // namespace attn
}  // namespace ttt
