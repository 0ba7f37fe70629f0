// Host-side interface of the fused TTT pre- / post-processing kernels (ttt_prepost.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ttt {
namespace prepost {

struct PreArgs {
    const __bf16 *XQ_raw, *XK_raw, *XV_raw;   // [B, L, NH*64]
    const float* rope;                        // [n_pos, 32, 2] (cos, sin) or null
    const int32_t *src, *pos;                 // [L] or null
    const float *ln_w, *ln_b;                 // [NH, 64]
    __bf16 *XQ, *XK, *XV;                     // [B, NH, L, 64]
    int B, L, NH;
    int t0, tn;                               // scan positions [t0, t0 + tn) only (tn = 0: all L) - a part of the sequence
};
struct PreBwdArgs {
    const __bf16 *XQ_raw, *XK_raw, *XV_raw;
    const float* rope;
    const int32_t *src, *pos;
    const float* ln_w;
    const __bf16 *dXQ, *dXK, *dXV;            // [B, NH, L, 64]
    __bf16 *dXQ_raw, *dXK_raw, *dXV_raw;      // [B, L, NH*64], token rows `ld_out` elements apart (NH*64: contiguous; 3*NH*64: the
    float *dlnw_part, *dlnb_part;             // [P, NH*64]                      three column blocks of ONE [B, L, 3*NH*64] buffer)
    int B, L, NH;
    long ld_out;
};
struct PostArgs {
    const __bf16* Y;                          // [B, NH, L, 64]
    const int32_t* src;
    const float *w, *b;                       // [NH*64]
    __bf16* out;                              // [B, L, NH*64]
    int B, L, NH;
    float eps;
    int t0, tn;                               // scan positions [t0, t0 + tn) only (tn = 0: all L)
};
struct PostBwdArgs {
    const __bf16 *Y, *dOut;
    const int32_t* src;
    const float* w;
    __bf16* dY;
    float *dw_part, *db_part;                 // [P, NH*64]
    int B, L, NH;
    float eps;
    int wt;                                   // waves per token team (set by post_backward)
};
struct GateArgs {
    const __bf16 *res, *y;                    // [B, L, D]
    const float *tanh_text, *tanh_video;      // [D]
    __bf16* out;
    int B, L, D, n_text;
};
struct GateBwdArgs {
    const __bf16 *g, *y;
    const float *tanh_text, *tanh_video;
    __bf16* dy;
    float* dtanh_part;                        // [P, 2, D]
    int B, L, D, n_text;
};

struct AdaLNArgs {
    const __bf16 *vid, *text;                 // [B, Lv, D], [B, Lt, D]
    const float *w, *b;                       // LayerNorm [D]
    const float *shift, *scale1p;             // [B, 2, D]: group 0 = text, 1 = video; scale1p = 1 + scale
    __bf16* out;                              // [B, Lt + Lv, D] = [text | video]
    int B, Lt, Lv, D;
    float eps;
};
struct AdaLNBwdArgs {
    const __bf16 *vid, *text, *dout;
    const float *w, *b, *scale1p;
    __bf16 *dvid, *dtext;
    float* part;                              // [B * 2 * P, 4, D]: dw, db, d scale1p, d shift per block
    int B, Lt, Lv, D, P;
    float eps;
    int wt;                                   // waves per token team (set by adaln_backward)
};
struct ResGateArgs {
    const __bf16 *vid, *text, *y;             // residual streams and y = [text | video] [B, Lt + Lv, D]
    const float* gate;                        // [B, 2, D]: group 0 = text, 1 = video
    __bf16 *ovid, *otext;
    int B, Lt, Lv, D;
};
struct ResGateBwdArgs {
    const __bf16 *dvid, *dtext, *y;
    const float* gate;
    __bf16* dy;                               // [B, Lt + Lv, D]
    float* dgate_part;                        // [P, B, 2, D]
    int B, Lt, Lv, D;
};

void pre_forward(const PreArgs& a, hipStream_t s);
int  pre_backward_partials(int NH);
void pre_backward(const PreBwdArgs& a, hipStream_t s);
int  post_blocks(int B, int L);
void post_forward(const PostArgs& a, hipStream_t s);
void post_backward(const PostBwdArgs& a, hipStream_t s);
void gate_forward(const GateArgs& a, hipStream_t s);
int  gate_backward_partials(int D);
void gate_backward(const GateBwdArgs& a, hipStream_t s);
void adaln_forward(const AdaLNArgs& a, hipStream_t s);
int  adaln_backward_partials();
void adaln_backward(const AdaLNBwdArgs& a, hipStream_t s);
void resgate_forward(const ResGateArgs& a, hipStream_t s);
int  resgate_backward_partials(int D);
void resgate_backward(const ResGateBwdArgs& a, hipStream_t s);

}  // namespace prepost
}  // namespace ttt
