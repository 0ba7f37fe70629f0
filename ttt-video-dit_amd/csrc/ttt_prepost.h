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
};
struct PreBwdArgs {
    const __bf16 *XQ_raw, *XK_raw, *XV_raw;
    const float* rope;
    const int32_t *src, *pos;
    const float* ln_w;
    const __bf16 *dXQ, *dXK, *dXV;            // [B, NH, L, 64]
    __bf16 *dXQ_raw, *dXK_raw, *dXV_raw;      // [B, L, NH*64]
    float *dlnw_part, *dlnb_part;             // [P, NH*64]
    int B, L, NH;
};
struct PostArgs {
    const __bf16* Y;                          // [B, NH, L, 64]
    const int32_t* src;
    const float *w, *b;                       // [NH*64]
    __bf16* out;                              // [B, L, NH*64]
    int B, L, NH;
    float eps;
};
struct PostBwdArgs {
    const __bf16 *Y, *dOut;
    const int32_t* src;
    const float* w;
    __bf16* dY;
    float *dw_part, *db_part;                 // [P, NH*64]
    int B, L, NH;
    float eps;
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

void pre_forward(const PreArgs& a, hipStream_t s);
int  pre_backward_partials(int NH);
void pre_backward(const PreBwdArgs& a, hipStream_t s);
int  post_blocks(int B, int L);
void post_forward(const PostArgs& a, hipStream_t s);
void post_backward(const PostBwdArgs& a, hipStream_t s);
void gate_forward(const GateArgs& a, hipStream_t s);
int  gate_backward_partials(int D);
void gate_backward(const GateBwdArgs& a, hipStream_t s);

}  // namespace prepost
}  // namespace ttt
