/*
 * ttt_hip.h - C ABI of the MI355X (gfx950) TTT scan kernels: the drop-in boundary.
 *
 * This library replaces, for the reference test-time-training/ttt-video-dit:
 *   - the un-vendored CUDA extension `test_time_training` (ttt-tk):
 *       ttt_forward  called at ttt/models/ssm/mlp_tk.py:116-133   -> ttt_hip_mlp_forward
 *       ttt_backward called at ttt/models/ssm/mlp_tk.py:227-275   -> ttt_hip_mlp_backward
 *   - the Triton TTT-Linear kernels:
 *       ttt_linear_scan_forward  (ttt/models/ssm/kernels/linear_forward.py:5-148,
 *                                 launched at ttt/models/ssm/linear_triton.py:98-129)   -> ttt_hip_linear_forward
 *       ttt_linear_scan_backward (ttt/models/ssm/kernels/linear_backward.py:200-520,
 *                                 launched at ttt/models/ssm/linear_triton.py:203-246)  -> ttt_hip_linear_backward
 *
 * Conventions (identical to the reference call sites):
 *   - every buffer is allocated by the caller and is contiguous; the library allocates nothing
 *     and owns nothing; results are written through the given pointers;
 *   - all pointers are DEVICE pointers on the current device;
 *   - kernels are enqueued on `stream` (a hipStream_t, NULL = default stream) and the call
 *     returns without synchronising; ttt_hip_mlp_backward may run part of its work on an internal
 *     stream of the library, ordered after everything queued on `stream` before the call and joined
 *     into `stream` before the call returns (stream semantics for the caller are unchanged);
 *   - return value 0 = enqueued; negative = argument/launch error, message via ttt_hip_last_error().
 *
 * Tensor shapes use the reference's names: B batch, NH heads, NC mini-batches, CS mini-batch
 * size, F head dim, H = 4F (TTT-MLP hidden), G checkpoint_group_size, K = ceil(NC/G).
 * "act" tensors (XQ/XK/XV/eta/XQW and their gradients) are bf16 or fp32 according to
 * ttt_dims.act_dtype; state, checkpoints and LayerNorm parameters are always fp32
 * (mlp_tk.py:95-98,107-113).
 */
#ifndef TTT_HIP_H
#define TTT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 4): ttt_hip_mlp_forward / _backward return -3 (sticky: an earlier backward hand-over timed out; acknowledge with
 * ttt_hip_sweep_error_clear), -10 (fewer than 4 compute units visible), -11 (no host-mapped error word), -12 (a HIP event /
 * stream call of the backward's two-stream schedule failed); the round-1 exports ttt_hip_debug_variant / ttt_hip_debug_helpers
 * are gone.  1: rounds 1 - 3. */
/* 5 (round 6): ttt_hip_mlp_forward_workspace is non-zero for the MFMA scan at mini-batches of 64 - the forward runs as a PAIR of
 * workgroups per (b,h) (the state chain on one CU publishes the updated state per step into a ring of records in the workspace, a second
 * CU runs the output path from them: same bits, -25 % per scan) - and ttt_hip_mlp_forward_chunk USES its workspace arguments (NULL / too
 * small: the one-workgroup scan, as ABI 4 callers get it).  A pair whose second workgroup is never scheduled gives up after 2 s, poisons
 * its outputs with NaN and sets the sticky error that ttt_hip_sweep_error_clear() acknowledges (return code -3 of the next call), like
 * the backward's cluster.  Signatures unchanged.  4: ttt_hip_stream_create_masked / _destroy / ttt_hip_debug_placement_probe. */
#define TTT_HIP_ABI_VERSION 5

enum { TTT_DTYPE_BF16 = 0, TTT_DTYPE_F32 = 1 };
/* implementation selector: AUTO picks the MFMA kernels when the geometry is supported - bf16, F=64 and
 * TTT-MLP forward/backward at CS=64, TTT-MLP forward at CS=16, TTT-Linear forward/backward at CS=16 -
 * and the generic fp32-arithmetic kernels otherwise; TTT_IMPL_MFMA makes an unsupported geometry an error. */
enum { TTT_IMPL_AUTO = 0, TTT_IMPL_GENERIC = 1, TTT_IMPL_MFMA = 2 };

typedef struct ttt_dims {
    int32_t B, NH, NC, CS, F, G;
    int32_t act_dtype;   /* TTT_DTYPE_* of XQ/XK/XV/last_eta/XQW and their gradients          */
    int32_t impl;        /* TTT_IMPL_*                                                          */
    float   eps;         /* LayerNorm epsilon; 1e-8 = reference ops path (ops/utils.py:4,21)    */
} ttt_dims;

/* ---- TTT-MLP forward: the 15 tensors of mlp_tk.py:116-133, same order ------------------- */
typedef struct ttt_mlp_fwd_args {
    const void*  XQ;             /* [B,NH,NC,CS,F] act                                          */
    const void*  XK;
    const void*  XV;
    const void*  last_eta;       /* [B,NH,NC,CS,1] act  (last row of the eta tile, mlp_tk.py:105)*/
    const float* ttt_norm_weight;/* [1,NH,1,F] */
    const float* ttt_norm_bias;  /* [1,NH,1,F] */
    const float* W1_init;        /* [B,NH,F,H] */
    const float* b1_init;        /* [B,NH,1,H] */
    const float* W2_init;        /* [B,NH,H,F] */
    const float* b2_init;        /* [B,NH,1,F] */
    float* W1_checkpoints;       /* [B,NH,K,F,H]  out: state entering steps 0,G,2G,...          */
    float* b1_checkpoints;       /* [B,NH,K,1,H] */
    float* W2_checkpoints;       /* [B,NH,K,H,F] */
    float* b2_checkpoints;       /* [B,NH,K,1,F] */
    void*  XQW;                  /* [B,NH,NC,CS,F] act, out                                     */
} ttt_mlp_fwd_args;

/* ---- TTT-MLP backward: the 42 tensors of mlp_tk.py:227-275, same order ------------------ */
typedef struct ttt_mlp_bwd_args {
    const void*  XQ; const void* XK; const void* XV; const void* last_eta;
    const float* ttt_norm_weight; const float* ttt_norm_bias;
    const float* W1_checkpoints; const float* b1_checkpoints;
    const float* W2_checkpoints; const float* b2_checkpoints;
    const void*  XQW;            /* forward output (unused by the arithmetic; kept for ABI parity) */
    /* caller-allocated re-materialisation scratch, [B,NH,G,...] (mlp_tk.py:192-210).  The generic kernels keep their per-step
     * state in the four *_init_group buffers (required there); the MFMA backward works in `ws` only and accepts NULL for all
     * sixteen; the twelve below are never touched by this implementation. */
    float* W1_init_group; float* b1_init_group; float* W2_init_group; float* b2_init_group;
    void*  x_hat_ln_group;          /* bf16 [B,NH,G,CS,F]  */
    float* std_ln_group;            /* f32  [B,NH,G,CS,1]  */
    void*  X2_group;                /* bf16 [B,NH,G,CS,H]  */
    void*  Z1_group;                /* bf16 [B,NH,G,CS,H]  */
    void*  Z1_bar_group;            /* bf16 [B,NH,G,CS,H]  */
    void*  X2_bar_group;            /* bf16 [B,NH,G,CS,H]  */
    void*  grad_l_wrt_Z2_group;     /* bf16 [B,NH,G,CS,F]  */
    void*  grad_l_wrt_Z1_group;     /* bf16 [B,NH,G,CS,H]  */
    void*  x_hat_fused_group;       /* bf16 [B,NH,G,CS,F]  */
    void*  grad_x_hat_fused_group;  /* bf16 [B,NH,G,CS,F]  */
    void*  grad_output_fused_group; /* bf16 [B,NH,G,CS,F]  */
    float* std_fused_group;         /* f32  [B,NH,G,CS,1]  */
    /* upstream gradients */
    const float* grad_L_W1_last; const float* grad_L_b1_last;   /* [B,NH,F,H] / [B,NH,1,H] (zeros) */
    const float* grad_L_W2_last; const float* grad_L_b2_last;
    const void*  grad_L_XQW;        /* [B,NH,NC,CS,F] act */
    /* outputs */
    float* grad_L_ttt_norm_weight;  /* [B,NH,1,F] per batch element (caller sums, mlp_tk.py:277) */
    float* grad_L_ttt_norm_bias;
    float* grad_L_W1_init; float* grad_L_b1_init; float* grad_L_W2_init; float* grad_L_b2_init;
    void*  grad_L_last_eta;         /* [B,NH,NC,CS,1] act */
    void*  grad_L_XQ; void* grad_L_XK; void* grad_L_XV;          /* [B,NH,NC,CS,F] act */
} ttt_mlp_bwd_args;

/* ---- TTT-Linear: tensor contract of linear_triton.py:98-129 / 203-246, except that eta is
 * passed as its last row [B,NH,NC,CS,1] (the Triton kernels index the last row of the full tile
 * themselves, kernels/linear_forward.py:90-101; the binding slices it) ----------------------- */
typedef struct ttt_linear_fwd_args {
    const void*  XQ; const void* XK; const void* XV; const void* last_eta;
    const float* ttt_norm_weight; const float* ttt_norm_bias;   /* [NH,F] */
    const float* W1_init;        /* [B,NH,F,F] */
    const float* b1_init;        /* [B,NH,1,F] */
    float* W1_checkpoints;       /* [B,NH,K,F,F] */
    float* b1_checkpoints;       /* [B,NH,K,1,F] */
    void*  XQW;                  /* [B,NH,NC,CS,F] act */
} ttt_linear_fwd_args;

typedef struct ttt_linear_bwd_args {
    const void*  XQ; const void* XK; const void* XV; const void* last_eta;
    const float* ttt_norm_weight; const float* ttt_norm_bias;
    const float* W1_checkpoints; const float* b1_checkpoints;
    const float* grad_L_W1_last; const float* grad_L_b1_last;
    const void*  grad_L_XQW;
    float* W1_init_group;        /* [B,NH,G,F,F] scratch (linear_triton.py:172); opaque: the MFMA kernel parks packed */
    float* b1_init_group;        /* [B,NH,G,1,F] scratch    per-step states there, not the fp32 states of the Triton kernel */
    float* grad_L_ttt_norm_weight; float* grad_L_ttt_norm_bias;  /* [B,NH,1,F] */
    float* grad_L_W1_init; float* grad_L_b1_init;
    void*  grad_L_last_eta;      /* [B,NH,NC,CS,1] act */
    void*  grad_L_XQ; void* grad_L_XK; void* grad_L_XV;
} ttt_linear_bwd_args;

/* Bytes of extra device workspace the chosen implementation needs for this call (0 = none).
 * The binding allocates it (the reference allocates all scratch on the Python side too). */
size_t ttt_hip_mlp_forward_workspace(const ttt_dims* d);
size_t ttt_hip_mlp_backward_workspace(const ttt_dims* d);
size_t ttt_hip_linear_forward_workspace(const ttt_dims* d);
size_t ttt_hip_linear_backward_workspace(const ttt_dims* d);

int ttt_hip_mlp_forward(const ttt_dims* d, const ttt_mlp_fwd_args* a, void* workspace, size_t workspace_bytes, void* stream);
int ttt_hip_mlp_backward(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* workspace, size_t workspace_bytes, void* stream);
/* An extension BESIDE the 15-tensor forward (round 5), not instead of it: the TTT-MLP forward over steps [step0, step0 + nsteps)
 * of the sequence that `d` and `a` describe (d->NC = the whole sequence; the tensors of `a` are the whole sequence's), started
 * from the state in a->W1_init .. b2_init and leaving the state after its last step in W1_final .. b2_final ([B,NH,...] fp32 like
 * the initial state; may alias it; all four NULL: not stored).  Parts start and end at checkpoint-group boundaries (step0 % G
 * == 0); their outputs and checkpoints land where the one-call forward puts them, with the same bits (the state is handed on
 * in fp32, exactly as the kernel holds it).  Lets a caller run the projections of the next part of the sequence on the CUs the
 * sequential scan leaves idle (ttt_amd/models/ssm/pipeline.py).  MFMA scan at mini-batches of 64 only. */
int ttt_hip_mlp_forward_chunk(const ttt_dims* d, const ttt_mlp_fwd_args* a, int step0, int nsteps, float* W1_final, float* b1_final,
                              float* W2_final, float* b2_final, void* workspace, size_t workspace_bytes, void* stream);
int ttt_hip_linear_forward(const ttt_dims* d, const ttt_linear_fwd_args* a, void* workspace, size_t workspace_bytes, void* stream);
int ttt_hip_linear_backward(const ttt_dims* d, const ttt_linear_bwd_args* a, void* workspace, size_t workspace_bytes, void* stream);

/* ---- Fused pre- / post-processing of the TTT layer (bf16 activations, head_dim 64) ---------------------------
 * These replace chains of PyTorch elementwise kernels around the scan; tensors are named after the reference code.
 *   pre : ttt/models/ssm/ttt_layer.py:252-306 (process_input) - L2-normalise XQ, XK per head (:264-266), 3-D RoPE on
 *         video tokens (ssm/utils.py:82-108), LayerNorm reconstruction target for XV (:219-235), re-layout
 *         [B,L,NH*F] -> [B,NH,NC,CS,F] (:237-250) and the token permutation `src` (scene interleave :157-186 and/or
 *         the time reversal of the bidirectional pass, cogvideo/dit.py:247-263): scan position t reads token src[t]
 *         and is rotated by rope[pos[t]] (pos[t] < 0: text token, no rotation).  src/pos may be NULL (identity / none).
 *   post: ttt_layer.py:327-334 - [B,NH,NC,CS,F] -> [B,L,D], inverse permutation, post_norm LayerNorm(D, eps).
 *   gate: cogvideo/dit.py:219-222 - out = residual + tanh(alpha_text | alpha_video) * y (text = first n_text tokens).
 * Parameter gradients are returned as per-block partial sums [P, ...] (P from the *_partials queries); the caller
 * reduces over P. */
int ttt_hip_pre_forward(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                        const float* rope /* [n_pos, F/2, 2] (cos, sin) */, const int32_t* src, const int32_t* pos,
                        const float* ln_w, const float* ln_b /* [NH, F] */, void* XQ, void* XK, void* XV, void* stream);
/* The same for the scan positions [t0, t0 + tn) only - a part of the sequence (round 5: the projections / pre-processing of the next
 * part run beside the scan of the current one, ttt_hip_mlp_forward_chunk).  The tensors are the whole sequence's. */
int ttt_hip_pre_forward_range(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                              const float* rope, const int32_t* src, const int32_t* pos, const float* ln_w, const float* ln_b,
                              void* XQ, void* XK, void* XV, int t0, int tn, void* stream);
int ttt_hip_pre_backward_partials(int NH);
int ttt_hip_pre_backward(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                         const float* rope, const int32_t* src, const int32_t* pos, const float* ln_w,
                         const void* dXQ, const void* dXK, const void* dXV, void* dXQ_raw, void* dXK_raw, void* dXV_raw,
                         float* dlnw_part, float* dlnb_part /* [P, NH*F] */, void* stream);
/* The same with the raw gradients' token rows `ld_out` elements apart (see ttt_hip_attn_pre_backward_ld). */
int ttt_hip_pre_backward_ld(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                            const float* rope, const int32_t* src, const int32_t* pos, const float* ln_w,
                            const void* dXQ, const void* dXK, const void* dXV, void* dXQ_raw, void* dXK_raw, void* dXV_raw, int64_t ld_out,
                            float* dlnw_part, float* dlnb_part, void* stream);
int ttt_hip_post_partials(int B, int L);
int ttt_hip_post_forward(int B, int L, int NH, int F, float eps, const void* Y, const int32_t* src, const float* w, const float* b,
                         void* out, void* stream);
int ttt_hip_post_forward_range(int B, int L, int NH, int F, float eps, const void* Y, const int32_t* src, const float* w, const float* b,
                               void* out, int t0, int tn, void* stream);      /* scan positions [t0, t0 + tn) only */
int ttt_hip_post_backward(int B, int L, int NH, int F, float eps, const void* Y, const void* dOut, const int32_t* src, const float* w,
                          void* dY, float* dw_part, float* db_part /* [P, NH*F] */, void* stream);
int ttt_hip_gate_forward(int B, int L, int D, int n_text, const void* res, const void* y, const float* tanh_text,
                         const float* tanh_video, void* out, void* stream);
int ttt_hip_gate_backward_partials(int D);
int ttt_hip_gate_backward(int B, int L, int D, int n_text, const void* g, const void* y, const float* tanh_text,
                          const float* tanh_video, void* dy, float* dtanh_part /* [P, 2, D] */, void* stream);

/* ---- Segment self-attention (head_dim 64, bf16) -----------------------------------------------------------------
 * Replaces F.scaled_dot_product_attention(q, k, v) (non-causal, no mask, no dropout) of the reference's local attention,
 * ttt/models/cogvideo/dit.py:196-198, and its autograd backward.  Every tensor is described by a base pointer and
 * element strides (batch, head, token); the head dimension (64) is contiguous, so both the reference's [B,NH,S,D]
 * views of [B,S,NH*D] projections and contiguous tensors are consumed without copies.  O = softmax(Q K^T * scale) V;
 * LSE [B,NH,S] fp32 = log-sum-exp of the scaled scores (saved for the backward).  The backward needs a caller-allocated
 * fp32 workspace Delta [B,NH,S]. */
typedef struct ttt_attn_tensor {
    void*   ptr;
    int64_t stride_b, stride_h, stride_s;   /* in elements */
} ttt_attn_tensor;
typedef struct ttt_attn_fwd_args {
    ttt_attn_tensor Q, K, V, O;             /* bf16 */
    float*  LSE;                            /* [B,NH,S] fp32, out (may be NULL for inference) */
    int32_t B, NH, S, D;                    /* D must be 64 */
    float   scale;                          /* 1/sqrt(D) in the reference */
} ttt_attn_fwd_args;
typedef struct ttt_attn_bwd_args {
    ttt_attn_tensor Q, K, V, O, dO;         /* bf16, in  */
    ttt_attn_tensor dQ, dK, dV;             /* bf16, out */
    const float* LSE;                       /* [B,NH,S] from the forward */
    float*  Delta;                          /* [B,NH,S] fp32 workspace */
    int32_t B, NH, S, D;
    float   scale;
} ttt_attn_bwd_args;
int ttt_hip_attn_forward(const ttt_attn_fwd_args* a, void* stream);
int ttt_hip_attn_backward(const ttt_attn_bwd_args* a, void* stream);

/* Fused per-head LayerNorm(64, eps) + 3-D RoPE of the attention's q and k (reference cogvideo/dit.py:184-195,
 * cogvideo/utils.py:424-437): q_raw / k_raw / q / k / dq_raw / dk_raw are contiguous [B, S, NH*64] bf16; tokens
 * s >= n_text are rotated by row (s - n_text) of the [n_pos, 64] fp32 cos / sin tables; LayerNorm parameters [64] fp32.
 * The backward takes dq / dk as strided [B,NH,S,64] views and returns parameter-gradient partial sums
 * [P, 4, 64] (dw_q, db_q, dw_k, db_k), P = ttt_hip_attn_pre_partials(B, S, NH); the caller reduces over P. */
int ttt_hip_attn_pre_forward(int B, int S, int NH, int n_text, float eps, const void* q_raw, const void* k_raw,
                             const float* wq, const float* bq, const float* wk, const float* bk,
                             const float* cos_table, const float* sin_table, void* q, void* k, void* stream);
int ttt_hip_attn_pre_partials(int B, int S, int NH);
int ttt_hip_attn_pre_backward(int B, int S, int NH, int n_text, float eps, const void* q_raw, const void* k_raw,
                              const ttt_attn_tensor* dq, const ttt_attn_tensor* dk, const float* wq, const float* wk,
                              const float* cos_table, const float* sin_table, void* dq_raw, void* dk_raw, float* part,
                              void* stream);
/* The same with the raw gradients' token rows `ld_out` elements apart (>= NH*64, a multiple of 8): dq_raw / dk_raw (and the
 * attention backward's dV, a strided ttt_attn_tensor anyway) can then be the column blocks of ONE [B, S, 3*NH*64] buffer, which
 * makes the weight gradients of the q / k / v projections one GEMM over the concatenated output gradient (round 5). */
int ttt_hip_attn_pre_backward_ld(int B, int S, int NH, int n_text, float eps, const void* q_raw, const void* k_raw,
                                 const ttt_attn_tensor* dq, const ttt_attn_tensor* dk, const float* wq, const float* wk,
                                 const float* cos_table, const float* sin_table, void* dq_raw, void* dk_raw, int64_t ld_out, float* part,
                                 void* stream);

/* ---- TransformerLayer glue (bf16 activations, fp32 parameter vectors) --------------------------------------------------
 * adaln: out[B, Lt+Lv, D] = [ shift_t + LN(text) * scale1p_t | shift_v + LN(vid) * scale1p_v ]  - LayerNorm(D, eps) with
 *        (w, b), per-(batch, group) modulation vectors shift / scale1p = 1 + scale laid out [B, 2, D] (group 0 = text,
 *        1 = video): reference cogvideo/dit.py:353-357 and :366-371 (layernorm, modulate, torch.cat) in one pass.  The
 *        backward returns d vid, d text and parameter-gradient partials [B*2*P, 4, D] (dw, db, d scale1p, d shift per
 *        block; block index = (batch * 2 + group) * P + p), P = ttt_hip_adaln_backward_partials().
 * resgate: new_vid = vid + gate_v * y[:, Lt:], new_text = text + gate_t * y[:, :Lt]  (dit.py:358-359, :372-373), gate
 *        [B, 2, D]; the backward writes dy [B, Lt+Lv, D] and d gate partials [P, B, 2, D], P = ttt_hip_resgate_backward_partials(D)
 *        (the residual gradients are the incoming gradients themselves). */
int ttt_hip_adaln_forward(int B, int Lt, int Lv, int D, float eps, const void* vid, const void* text, const float* w, const float* b,
                          const float* shift, const float* scale1p, void* out, void* stream);
int ttt_hip_adaln_backward_partials(void);
int ttt_hip_adaln_backward(int B, int Lt, int Lv, int D, float eps, const void* vid, const void* text, const void* dout,
                           const float* w, const float* b, const float* scale1p, void* dvid, void* dtext, float* part, void* stream);
int ttt_hip_resgate_forward(int B, int Lt, int Lv, int D, const void* vid, const void* text, const void* y, const float* gate,
                            void* ovid, void* otext, void* stream);
int ttt_hip_resgate_backward_partials(int D);
int ttt_hip_resgate_backward(int B, int Lt, int Lv, int D, const void* dvid, const void* dtext, const void* y, const float* gate,
                             void* dy, float* dgate_part, void* stream);

/* Which implementation TTT_IMPL_AUTO resolves to for these dims (returns TTT_IMPL_GENERIC/MFMA). */
int ttt_hip_resolve_impl(const ttt_dims* d, int is_mlp, int is_backward);

/* DEBUG: when given a device buffer of 16 zero-initialised uint64, the MFMA kernels' workgroup 0 adds
 * per-phase shader-cycle totals into it (NULL switches the instrumentation off). */
void        ttt_hip_debug_timing(void* device_buffer);
/* DEBUG: force the number of checkpoint groups the MFMA backward re-materialises per chunk (0 = automatic,
 * sized to cover the 256 CUs); lets tests exercise the chunk-to-chunk gradient hand-over at small sizes. */
void        ttt_hip_debug_groups_per_chunk(int groups);
/* DEBUG knobs by name (five; every A/B option of rounds 2 - 4 was decided on hardware and removed with the losing code in round 5):
 * "groups_per_chunk" (checkpoint groups per backward chunk, 0 = automatic: tests exercise the chunk hand-over at small sizes),
 * "overlap_tail" (TTT-MLP backward: 1 (default) = the tail kernel of a chunk runs on an internal side stream beside the next chunk's
 * sweep and the caller's stream joins it before the call returns; 0 = everything on the caller's stream; identical results),
 * "fast_records" (sweep hand-over: 1 (default) = plain, L2-resident records once the four workgroups of a cluster have proven that they
 * share an XCD; 0 = write-through records always; identical results), "sweep_fast_count" (query: returns -2 - the number of cluster
 * workgroup launches that took the plain form), "sweep_fault" (fault injection for the tests of the hand-over failure path: workgroup
 * 3 of every backward cluster leaves before its first hand-over).  Returns 0, or -1 for an unknown name. */
int         ttt_hip_debug_option(const char* name, int value);
/* DEBUG: device buffer (>= 120000 floats) receiving the step-0 intermediates of workgroup 0 (NULL = off). */
/* TTT-MLP backward, cluster form (four workgroups per (b,h) exchanging partial tiles inside the launch; at most n_cu / 4
 * clusters per launch, so that the four are co-resident).  A bounded hand-over poll that gives up - a partner workgroup that is
 * never scheduled - is a HARD error: the kernel fills what it still writes (dV, d(eta), the state / LayerNorm gradients) with
 * NaN and records 1 + the (b,h) index in a host-mapped word; from then on ttt_hip_mlp_forward / ttt_hip_mlp_backward return -3 on
 * entry (no synchronisation) until ttt_hip_sweep_error_clear() acknowledges it.  ttt_hip_debug_sweep_error() synchronises the
 * device and returns the word (0 = no hand-over has given up). */
unsigned    ttt_hip_debug_sweep_error(void);
void        ttt_hip_sweep_error_clear(void);
/* DEBUG (stress tests of the cluster hand-over): enqueue on `stream` a kernel of `workgroups` workgroups that each hold
 * `lds_bytes` of LDS (<= 160 KiB: a CU that hosts one cannot host a sweep workgroup) and do nothing for `microseconds`
 * (<= 100 000) of wall-clock time - what a collective's kernels do to the CUs a cluster launch counts on.  Returns 0 / -1. */
int         ttt_hip_debug_occupy_cus(int workgroups, int lds_bytes, int microseconds, void* stream);
void        ttt_hip_debug_dump(float* device_buffer);
/* A HIP stream confined to the compute units of `cu_mask` (`mask_words` x 32 bits, hipExtStreamCreateWithCUMask): kernels enqueued on
 * it can never occupy a CU outside the mask - e.g. the weight-gradient GEMMs or the collectives of a training step beside the TTT-MLP
 * backward's cluster sweep, which needs its four workgroups per (b,h) co-resident.  Wrap it with torch.cuda.ExternalStream
 * (test_time_training.masked_stream).  ttt_hip_debug_placement_probe: `workgroups` single-wave workgroups holding `lds_bytes` of
 * LDS for `microseconds` on `stream`, each storing (XCC_ID << 16 | HW_ID[15:0]) of the CU it ran on into device_out[workgroup] - how a
 * mask bit maps to (XCD, shader engine, CU) on this part.  Return 0, or a negative code (ttt_hip_last_error()). */
int         ttt_hip_stream_create_masked(const unsigned* cu_mask, int mask_words, void** stream);
int         ttt_hip_stream_destroy(void* stream);
int         ttt_hip_debug_placement_probe(unsigned* device_out, int workgroups, int lds_bytes, int microseconds, void* stream);

int         ttt_hip_abi_version(void);
const char* ttt_hip_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* TTT_HIP_H */
