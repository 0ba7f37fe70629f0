// Host-side interface of the segment-attention kernels (attn_fwd.hip, attn_bwd.hip).
#pragma once
#include <hip/hip_runtime.h>
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

}  // namespace attn
}  // namespace ttt
