// extern "C" boundary of libttt_hip.so (see include/ttt_hip.h): argument validation,
// implementation dispatch, error reporting.  No allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ttt_hip.h"
#include "ttt_generic.h"
#include "ttt_mfma.h"

static thread_local char g_err[512] = "";

static int fail(const char* fmt, const char* what = "") {
    snprintf(g_err, sizeof(g_err), fmt, what);
    return -1;
}

static int check_dims(const ttt_dims* d) {
    if (!d) return fail("ttt_hip: null dims");
    if (d->B <= 0 || d->NH <= 0 || d->NC <= 0 || d->CS <= 0 || d->F <= 0 || d->G <= 0)
        return fail("ttt_hip: non-positive dimension");
    if (d->act_dtype != TTT_DTYPE_BF16 && d->act_dtype != TTT_DTYPE_F32) return fail("ttt_hip: bad act_dtype");
    if (d->impl < TTT_IMPL_AUTO || d->impl > TTT_IMPL_MFMA) return fail("ttt_hip: bad impl");
    if (!(d->eps >= 0.f)) return fail("ttt_hip: bad eps");
    return 0;
}

static int resolve(const ttt_dims* d, bool mlp, bool bwd) {
    if (d->impl == TTT_IMPL_GENERIC) return ttt::generic::supports(d) ? TTT_IMPL_GENERIC : -1;
    if (d->impl == TTT_IMPL_MFMA) return ttt::mfma::supports(d, mlp, bwd) ? TTT_IMPL_MFMA : -1;
    if (ttt::mfma::supports(d, mlp, bwd)) return TTT_IMPL_MFMA;
    if (ttt::generic::supports(d)) return TTT_IMPL_GENERIC;
    return -1;
}

static int post_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "ttt_hip: %s launch failed: %s", what, hipGetErrorString(e));
        return -2;
    }
    return 0;
}

#define NEED(p) do { if (!(a->p)) return fail("ttt_hip: null pointer argument %s", #p); } while (0)

extern "C" {

int ttt_hip_abi_version(void) { return TTT_HIP_ABI_VERSION; }
void ttt_hip_debug_timing(void* device_buffer) { ttt::mfma::set_debug_timing(device_buffer); }
void ttt_hip_debug_groups_per_chunk(int groups) { ttt::mfma::set_debug_groups_per_chunk(groups); }
void ttt_hip_debug_variant(int v) { ttt::mfma::set_debug_variant(v); }
void ttt_hip_debug_dump(float* buf) { ttt::mfma::set_debug_dump(buf); }
const char* ttt_hip_last_error(void) { return g_err; }

int ttt_hip_resolve_impl(const ttt_dims* d, int is_mlp, int is_backward) {
    if (check_dims(d)) return -1;
    return resolve(d, is_mlp != 0, is_backward != 0);
}

static size_t ws_bytes(const ttt_dims* d, bool mlp, bool bwd) {
    if (check_dims(d)) return 0;
    int r = resolve(d, mlp, bwd);
    if (r == TTT_IMPL_GENERIC) return ttt::generic::workspace_bytes(d, mlp);
    if (r == TTT_IMPL_MFMA) return ttt::mfma::workspace_bytes(d, mlp, bwd);
    return 0;
}
size_t ttt_hip_mlp_forward_workspace(const ttt_dims* d) { return ws_bytes(d, true, false); }
size_t ttt_hip_mlp_backward_workspace(const ttt_dims* d) { return ws_bytes(d, true, true); }
size_t ttt_hip_linear_forward_workspace(const ttt_dims* d) { return ws_bytes(d, false, false); }
size_t ttt_hip_linear_backward_workspace(const ttt_dims* d) { return ws_bytes(d, false, true); }

int ttt_hip_mlp_forward(const ttt_dims* d, const ttt_mlp_fwd_args* a, void* ws, size_t wsb, void* stream) {
    if (check_dims(d)) return -1;
    if (!a) return fail("ttt_hip: null args");
    NEED(XQ); NEED(XK); NEED(XV); NEED(last_eta); NEED(ttt_norm_weight); NEED(ttt_norm_bias);
    NEED(W1_init); NEED(b1_init); NEED(W2_init); NEED(b2_init);
    NEED(W1_checkpoints); NEED(b1_checkpoints); NEED(W2_checkpoints); NEED(b2_checkpoints); NEED(XQW);
    int r = resolve(d, true, false);
    if (r < 0) return fail("ttt_hip: mlp_forward: unsupported geometry/dtype for the requested impl");
    if (wsb < ws_bytes(d, true, false) || (ws_bytes(d, true, false) && !ws)) return fail("ttt_hip: mlp_forward: workspace too small");
    if (r == TTT_IMPL_MFMA) ttt::mfma::mlp_forward(d, a, ws, (hipStream_t)stream);
    else ttt::generic::mlp_forward(d, a, ws, (hipStream_t)stream);
    return post_launch("mlp_forward");
}

int ttt_hip_mlp_backward(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, size_t wsb, void* stream) {
    if (check_dims(d)) return -1;
    if (!a) return fail("ttt_hip: null args");
    NEED(XQ); NEED(XK); NEED(XV); NEED(last_eta); NEED(ttt_norm_weight); NEED(ttt_norm_bias);
    NEED(W1_checkpoints); NEED(b1_checkpoints); NEED(W2_checkpoints); NEED(b2_checkpoints);
    NEED(W1_init_group); NEED(b1_init_group); NEED(W2_init_group); NEED(b2_init_group);
    NEED(grad_L_W1_last); NEED(grad_L_b1_last); NEED(grad_L_W2_last); NEED(grad_L_b2_last); NEED(grad_L_XQW);
    NEED(grad_L_ttt_norm_weight); NEED(grad_L_ttt_norm_bias); NEED(grad_L_W1_init); NEED(grad_L_b1_init);
    NEED(grad_L_W2_init); NEED(grad_L_b2_init); NEED(grad_L_last_eta); NEED(grad_L_XQ); NEED(grad_L_XK); NEED(grad_L_XV);
    int r = resolve(d, true, true);
    if (r < 0) return fail("ttt_hip: mlp_backward: unsupported geometry/dtype for the requested impl");
    if (wsb < ws_bytes(d, true, true) || (ws_bytes(d, true, true) && !ws)) return fail("ttt_hip: mlp_backward: workspace too small");
    if (r == TTT_IMPL_MFMA) ttt::mfma::mlp_backward(d, a, ws, (hipStream_t)stream);
    else ttt::generic::mlp_backward(d, a, ws, (hipStream_t)stream);
    return post_launch("mlp_backward");
}

int ttt_hip_linear_forward(const ttt_dims* d, const ttt_linear_fwd_args* a, void* ws, size_t wsb, void* stream) {
    if (check_dims(d)) return -1;
    if (!a) return fail("ttt_hip: null args");
    NEED(XQ); NEED(XK); NEED(XV); NEED(last_eta); NEED(ttt_norm_weight); NEED(ttt_norm_bias);
    NEED(W1_init); NEED(b1_init); NEED(W1_checkpoints); NEED(b1_checkpoints); NEED(XQW);
    int r = resolve(d, false, false);
    if (r < 0) return fail("ttt_hip: linear_forward: unsupported geometry/dtype for the requested impl");
    if (wsb < ws_bytes(d, false, false) || (ws_bytes(d, false, false) && !ws)) return fail("ttt_hip: linear_forward: workspace too small");
    if (r == TTT_IMPL_MFMA) ttt::mfma::linear_forward(d, a, ws, (hipStream_t)stream);
    else ttt::generic::linear_forward(d, a, ws, (hipStream_t)stream);
    return post_launch("linear_forward");
}

int ttt_hip_linear_backward(const ttt_dims* d, const ttt_linear_bwd_args* a, void* ws, size_t wsb, void* stream) {
    if (check_dims(d)) return -1;
    if (!a) return fail("ttt_hip: null args");
    NEED(XQ); NEED(XK); NEED(XV); NEED(last_eta); NEED(ttt_norm_weight); NEED(ttt_norm_bias);
    NEED(W1_checkpoints); NEED(b1_checkpoints); NEED(grad_L_W1_last); NEED(grad_L_b1_last); NEED(grad_L_XQW);
    NEED(W1_init_group); NEED(b1_init_group); NEED(grad_L_ttt_norm_weight); NEED(grad_L_ttt_norm_bias);
    NEED(grad_L_W1_init); NEED(grad_L_b1_init); NEED(grad_L_last_eta); NEED(grad_L_XQ); NEED(grad_L_XK); NEED(grad_L_XV);
    int r = resolve(d, false, true);
    if (r < 0) return fail("ttt_hip: linear_backward: unsupported geometry/dtype for the requested impl");
    if (wsb < ws_bytes(d, false, true) || (ws_bytes(d, false, true) && !ws)) return fail("ttt_hip: linear_backward: workspace too small");
    if (r == TTT_IMPL_MFMA) ttt::mfma::linear_backward(d, a, ws, (hipStream_t)stream);
    else ttt::generic::linear_backward(d, a, ws, (hipStream_t)stream);
    return post_launch("linear_backward");
}

}  // extern "C"
