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
    __bf16 *dq_raw, *dk_raw;            // [B, S, NH*64], token rows `ld_out` elements apart (NH*64 = contiguous)
    long ld_out;
    float* part;                        // [P, 4, 64]: dwq, dbq, dwk, dbk partial sums, P = pre_blocks(B*S*NH)
    int B, S, NH, n_text;
    float eps;
};
int  pre_blocks(long rows);
void launch_pre_forward(const PreParams& a, hipStream_t s);
void launch_pre_backward(const PreBwdParams& a, hipStream_t s);

void launch_forward(const FwdParams& p, hipStream_t s);
void launch_backward(const BwdParams& p, hipStream_t s);

// dQ and dK / dV through the workgroup bodies of attn_body.h (attn_v2.hip)
void launch_dq_v2(const BwdParams& p, hipStream_t s);
void launch_dkdv_v2(const BwdParams& p, hipStream_t s);
void set_debug_attn_variant(int v);     // DEBUG A/B: 1 (default) s_setprio around one MFMA cluster per backward kernel, 0 without

}  // namespace attn
}  // namespace ttt
