// extern "C" boundary of libttt_hip.so (see include/ttt_hip.h): argument validation,
// implementation dispatch, error reporting.  No allocation, no synchronisation.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include "../../include/ttt_hip.h"
#include "ttt_generic.h"
#include "ttt_mfma.h"
#include "ttt_prepost.h"
#include "attn.h"
#include "ttt_mfma_int.h"

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
int ttt_hip_debug_option(const char* name, int value) {
    if (!name) return -1;
    if (!strcmp(name, "fast_records")) ttt::mfma::set_debug_fast_records(value);              // cluster sweep hand-over: 1 (default) / 0 = write-through records always
    else if (!strcmp(name, "sweep_fast_count")) return -2 - (int)ttt::mfma::read_sweep_fast_count();      // query: returns -2 - count
    else if (!strcmp(name, "overlap_tail")) ttt::mfma::set_debug_overlap_tail(value);            // backward schedule: 2 (default) tail of chunk c and recompute of chunk c-2 beside the sweep of chunk c-1 / 1 tail only / 0 one stream
    else if (!strcmp(name, "groups_per_chunk")) ttt::mfma::set_debug_groups_per_chunk(value);
    else if (!strcmp(name, "deriver_split")) ttt::mfma::set_debug_deriver_split(value);         // sweep: 1 (default) / 0 = barrier Bc behind the derivers' reverse step
    else if (!strcmp(name, "attn_prio")) ttt::attn::set_debug_attn_variant(value);               // attention backward: 1 (default) / 0 = without the s_setprio pair per kernel
    else if (!strcmp(name, "sweep_fault")) ttt::mfma::set_debug_sweep_fault(value);              // fault injection: workgroup 3 of every sweep cluster leaves early
    else if (!strcmp(name, "scan_pair")) ttt::mfma::set_debug_scan_pair(value);                  // forward scan at CS = 64: 1 (default) a pair of workgroups per (b,h) / 0 = one
    else if (!strcmp(name, "scan_fault")) ttt::mfma::set_debug_scan_fault(value);                // fault injection: role B of every scan pair leaves at once
    else return -1;
    return 0;
}
unsigned ttt_hip_debug_sweep_error(void) { return ttt::mfma::read_sweep_error(); }

// DEBUG: a kernel that holds CUs (through its LDS allocation) for a bounded wall-clock time and does nothing
__global__ __launch_bounds__(64) void occupy_cus_kernel(unsigned long long ticks) {
    extern __shared__ __attribute__((aligned(16))) char occ_lds[];
    occ_lds[threadIdx.x] = 0;
    const unsigned long long t0 = wall_clock64();                 // constant 100 MHz counter
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int ttt_hip_debug_occupy_cus(int workgroups, int lds_bytes, int microseconds, void* stream) {
    if (workgroups < 1 || workgroups > 1024 || lds_bytes < 64 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 100000) return -1;
    if (hipFuncSetAttribute((const void*)occupy_cus_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -1;
    hipLaunchKernelGGL(occupy_cus_kernel, dim3(workgroups), dim3(64), lds_bytes, (hipStream_t)stream, 100ull * (unsigned long long)microseconds);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
void ttt_hip_sweep_error_clear(void) { ttt::mfma::clear_sweep_error(); }

// Streams confined to a set of compute units (hipExtStreamCreateWithCUMask) and a probe that reports where workgroups of a stream
// actually run: word = HW_REG_XCC_ID[3:0] << 16 | HW_REG_HW_ID[15:0] (se_id [15:13], sh_id [12], cu_id [11:8]) per workgroup.
int ttt_hip_stream_create_masked(const unsigned* cu_mask, int mask_words, void** stream) {
    if (!cu_mask || mask_words < 1 || mask_words > 32 || !stream) return fail("ttt_hip: stream_create_masked: bad arguments");
    hipStream_t s = nullptr;
    if (hipExtStreamCreateWithCUMask(&s, (uint32_t)mask_words, cu_mask) != hipSuccess) {
        (void)hipGetLastError();
        return fail("ttt_hip: hipExtStreamCreateWithCUMask failed");
    }
    *stream = (void*)s;
    return 0;
}
int ttt_hip_stream_destroy(void* stream) { return hipStreamDestroy((hipStream_t)stream) == hipSuccess ? 0 : -1; }
__global__ __launch_bounds__(64) void placement_probe_kernel(unsigned* out, unsigned long long ticks) {
    extern __shared__ __attribute__((aligned(16))) char probe_lds[];
    probe_lds[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));       // HW_REG_HW_ID[15:0]
        out[blockIdx.x] = (xcc << 16) | (hw & 0xffffu);
    }
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int ttt_hip_debug_placement_probe(unsigned* device_out, int workgroups, int lds_bytes, int microseconds, void* stream) {
    if (!device_out || workgroups < 1 || workgroups > 4096 || lds_bytes < 64 || lds_bytes > 160 * 1024 || microseconds < 0 || microseconds > 100000) return -1;
    if (hipFuncSetAttribute((const void*)placement_probe_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes) != hipSuccess) return -1;
    hipLaunchKernelGGL(placement_probe_kernel, dim3(workgroups), dim3(64), lds_bytes, (hipStream_t)stream, device_out, 100ull * (unsigned long long)microseconds);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// A hand-over of the TTT-MLP backward that gave up has poisoned that call's gradients (NaN); it is also STICKY: every later
// TTT-MLP call of the process fails here, on entry, until the caller acknowledges it (no synchronisation: the word is host-mapped).
static int check_sweep_error(const char* what) {
    const unsigned e = ttt::mfma::peek_sweep_error();
    if (!e) return 0;
    snprintf(g_err, sizeof(g_err), "ttt_hip: %s refused: an earlier TTT-MLP hand-over timed out in this process (the backward's cluster or the "
             "forward scan's pair of (b,h) %u: a partner workgroup was never scheduled); that call's results were poisoned with NaN.  Acknowledge "
             "with ttt_hip_sweep_error_clear().",
             what, e - 1u);
    return -3;
}
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
    if (check_sweep_error("mlp_forward")) return -3;
    if (r == TTT_IMPL_MFMA) ttt::mfma::mlp_forward(d, a, ws, (hipStream_t)stream);
    else ttt::generic::mlp_forward(d, a, ws, (hipStream_t)stream);
    return post_launch("mlp_forward");
}

int ttt_hip_mlp_forward_chunk(const ttt_dims* d, const ttt_mlp_fwd_args* a, int step0, int nsteps, float* W1_final, float* b1_final,
                              float* W2_final, float* b2_final, void* ws, size_t wsb, void* stream) {
    if (check_dims(d)) return -1;
    if (!a) return fail("ttt_hip: null args");
    NEED(XQ); NEED(XK); NEED(XV); NEED(last_eta); NEED(ttt_norm_weight); NEED(ttt_norm_bias);
    NEED(W1_init); NEED(b1_init); NEED(W2_init); NEED(b2_init);
    NEED(W1_checkpoints); NEED(b1_checkpoints); NEED(W2_checkpoints); NEED(b2_checkpoints); NEED(XQW);
    if (resolve(d, true, false) != TTT_IMPL_MFMA || d->CS != 64)
        return fail("ttt_hip: mlp_forward_chunk: only the MFMA scan at mini-batches of 64 continues from a state");
    if (step0 < 0 || nsteps <= 0 || step0 + nsteps > d->NC || step0 % d->G != 0 || ((step0 + nsteps) % d->G != 0 && step0 + nsteps != d->NC))
        return fail("ttt_hip: mlp_forward_chunk: a part of the sequence starts and ends at checkpoint-group boundaries (or at the end)");
    if ((W1_final || b1_final || W2_final || b2_final) && !(W1_final && b1_final && W2_final && b2_final))
        return fail("ttt_hip: mlp_forward_chunk: give all four final-state buffers or none");
    if (check_sweep_error("mlp_forward_chunk")) return -3;
    // the workspace of ttt_hip_mlp_forward_workspace (the pair scan's ring); null / too small: the one-workgroup scan (ABI 4 callers)
    void* use_ws = (ws && wsb >= ws_bytes(d, true, false)) ? ws : nullptr;
    ttt::mfma::mlp_forward_chunk(d, a, step0, nsteps, W1_final, b1_final, W2_final, b2_final, use_ws, (hipStream_t)stream);
    return post_launch("mlp_forward_chunk");
}

int ttt_hip_mlp_backward(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, size_t wsb, void* stream) {
    if (check_dims(d)) return -1;
    if (!a) return fail("ttt_hip: null args");
    NEED(XQ); NEED(XK); NEED(XV); NEED(last_eta); NEED(ttt_norm_weight); NEED(ttt_norm_bias);
    NEED(W1_checkpoints); NEED(b1_checkpoints); NEED(W2_checkpoints); NEED(b2_checkpoints);
    NEED(grad_L_W1_last); NEED(grad_L_b1_last); NEED(grad_L_W2_last); NEED(grad_L_b2_last); NEED(grad_L_XQW);
    NEED(grad_L_ttt_norm_weight); NEED(grad_L_ttt_norm_bias); NEED(grad_L_W1_init); NEED(grad_L_b1_init);
    NEED(grad_L_W2_init); NEED(grad_L_b2_init); NEED(grad_L_last_eta); NEED(grad_L_XQ); NEED(grad_L_XK); NEED(grad_L_XV);
    int r = resolve(d, true, true);
    if (r < 0) return fail("ttt_hip: mlp_backward: unsupported geometry/dtype for the requested impl");
    if (wsb < ws_bytes(d, true, true) || (ws_bytes(d, true, true) && !ws)) return fail("ttt_hip: mlp_backward: workspace too small");
    if (check_sweep_error("mlp_backward")) return -3;
    // the caller-allocated re-materialisation scratch of the reference contract: only the generic kernels keep their per-step
    // state in the four *_init_group buffers; the MFMA backward works in `ws` and accepts NULL for all sixteen
    if (r != TTT_IMPL_MFMA) { NEED(W1_init_group); NEED(b1_init_group); NEED(W2_init_group); NEED(b2_init_group); }
    if (r == TTT_IMPL_MFMA) {
        const int rc = ttt::mfma::mlp_backward(d, a, ws, (hipStream_t)stream);
        if (rc == -10) return fail("ttt_hip: mlp_backward: the MFMA backward needs at least 4 visible compute units (four co-resident workgroups per (b,h))");
        if (rc == -12) return fail("ttt_hip: mlp_backward: a HIP event / stream call of the two-stream schedule failed; the gradients of this call are not valid");
        if (rc) return fail("ttt_hip: mlp_backward: could not allocate the host-mapped error word");
    } else ttt::generic::mlp_backward(d, a, ws, (hipStream_t)stream);
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

/* ---- fused pre / post-processing of the TTT layer (ttt_prepost.hip) --------------------------------- */
static int check_pp(int B, int L, int NH, int F) {
    if (B <= 0 || L <= 0 || NH <= 0) return fail("ttt_hip: non-positive dimension");
    if (F != 64) return fail("ttt_hip: fused pre/post kernels need head_dim 64");
    if (NH * 8 > 512) return fail("ttt_hip: too many heads for the post kernels (NH * 64 <= 4096 features)");
    if ((long long)B * L >= (1ll << 31) / 4) return fail("ttt_hip: B * L too large for the pre / post kernels' 32-bit token arithmetic");
    return 0;
}

int ttt_hip_pre_forward(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                        const float* rope, const int32_t* src, const int32_t* pos, const float* ln_w, const float* ln_b,
                        void* XQ, void* XK, void* XV, void* stream) {
    return ttt_hip_pre_forward_range(B, L, NH, F, XQ_raw, XK_raw, XV_raw, rope, src, pos, ln_w, ln_b, XQ, XK, XV, 0, L, stream);
}
int ttt_hip_pre_forward_range(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                              const float* rope, const int32_t* src, const int32_t* pos, const float* ln_w, const float* ln_b,
                              void* XQ, void* XK, void* XV, int t0, int tn, void* stream) {
    if (check_pp(B, L, NH, F)) return -1;
    if (!XQ_raw || !XK_raw || !XV_raw || !ln_w || !ln_b || !XQ || !XK || !XV) return fail("ttt_hip: pre_forward: null pointer");
    if (pos && !rope) return fail("ttt_hip: pre_forward: positions without a rotation table");
    if (t0 < 0 || tn <= 0 || t0 + tn > L) return fail("ttt_hip: pre_forward: the range of scan positions must lie inside the sequence");
    ttt::prepost::PreArgs a = {(const __bf16*)XQ_raw, (const __bf16*)XK_raw, (const __bf16*)XV_raw, rope, src, pos, ln_w, ln_b,
                               (__bf16*)XQ, (__bf16*)XK, (__bf16*)XV, B, L, NH, t0, tn};
    ttt::prepost::pre_forward(a, (hipStream_t)stream);
    return post_launch("pre_forward");
}
int ttt_hip_pre_backward_partials(int NH) { return ttt::prepost::pre_backward_partials(NH); }
int ttt_hip_pre_backward(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                         const float* rope, const int32_t* src, const int32_t* pos, const float* ln_w,
                         const void* dXQ, const void* dXK, const void* dXV, void* dXQ_raw, void* dXK_raw, void* dXV_raw,
                         float* dlnw_part, float* dlnb_part, void* stream) {
    return ttt_hip_pre_backward_ld(B, L, NH, F, XQ_raw, XK_raw, XV_raw, rope, src, pos, ln_w, dXQ, dXK, dXV, dXQ_raw, dXK_raw, dXV_raw,
                                   (int64_t)NH * F, dlnw_part, dlnb_part, stream);
}
int ttt_hip_pre_backward_ld(int B, int L, int NH, int F, const void* XQ_raw, const void* XK_raw, const void* XV_raw,
                            const float* rope, const int32_t* src, const int32_t* pos, const float* ln_w,
                            const void* dXQ, const void* dXK, const void* dXV, void* dXQ_raw, void* dXK_raw, void* dXV_raw, int64_t ld_out,
                            float* dlnw_part, float* dlnb_part, void* stream) {
    if (check_pp(B, L, NH, F)) return -1;
    if (ld_out < (int64_t)NH * F || (ld_out & 7) || ((uintptr_t)dXQ_raw & 15) || ((uintptr_t)dXK_raw & 15) || ((uintptr_t)dXV_raw & 15))
        return fail("ttt_hip: pre_backward: the raw gradients' row stride must be >= NH*F and a multiple of 8 elements, their bases 16-byte aligned");
    if (!XQ_raw || !XK_raw || !XV_raw || !ln_w || !dXQ || !dXK || !dXV || !dXQ_raw || !dXK_raw || !dXV_raw || !dlnw_part || !dlnb_part)
        return fail("ttt_hip: pre_backward: null pointer");
    ttt::prepost::PreBwdArgs a = {(const __bf16*)XQ_raw, (const __bf16*)XK_raw, (const __bf16*)XV_raw, rope, src, pos, ln_w,
                                  (const __bf16*)dXQ, (const __bf16*)dXK, (const __bf16*)dXV,
                                  (__bf16*)dXQ_raw, (__bf16*)dXK_raw, (__bf16*)dXV_raw, dlnw_part, dlnb_part, B, L, NH, (long)ld_out};
    ttt::prepost::pre_backward(a, (hipStream_t)stream);
    return post_launch("pre_backward");
}
int ttt_hip_post_partials(int B, int L) { return ttt::prepost::post_blocks(B, L); }
int ttt_hip_post_forward(int B, int L, int NH, int F, float eps, const void* Y, const int32_t* src, const float* w, const float* b,
                         void* out, void* stream) {
    return ttt_hip_post_forward_range(B, L, NH, F, eps, Y, src, w, b, out, 0, L, stream);
}
int ttt_hip_post_forward_range(int B, int L, int NH, int F, float eps, const void* Y, const int32_t* src, const float* w, const float* b,
                               void* out, int t0, int tn, void* stream) {
    if (check_pp(B, L, NH, F)) return -1;
    if (!Y || !w || !b || !out) return fail("ttt_hip: post_forward: null pointer");
    if (t0 < 0 || tn <= 0 || t0 + tn > L) return fail("ttt_hip: post_forward: the range of scan positions must lie inside the sequence");
    ttt::prepost::PostArgs a = {(const __bf16*)Y, src, w, b, (__bf16*)out, B, L, NH, eps, t0, tn};
    ttt::prepost::post_forward(a, (hipStream_t)stream);
    return post_launch("post_forward");
}
int ttt_hip_post_backward(int B, int L, int NH, int F, float eps, const void* Y, const void* dOut, const int32_t* src, const float* w,
                          void* dY, float* dw_part, float* db_part, void* stream) {
    if (check_pp(B, L, NH, F)) return -1;
    if (!Y || !dOut || !w || !dY || !dw_part || !db_part) return fail("ttt_hip: post_backward: null pointer");
    ttt::prepost::PostBwdArgs a = {(const __bf16*)Y, (const __bf16*)dOut, src, w, (__bf16*)dY, dw_part, db_part, B, L, NH, eps};
    ttt::prepost::post_backward(a, (hipStream_t)stream);
    return post_launch("post_backward");
}
int ttt_hip_gate_forward(int B, int L, int D, int n_text, const void* res, const void* y, const float* tanh_text,
                         const float* tanh_video, void* out, void* stream) {
    if (B <= 0 || L <= 0 || D <= 0 || D % 8) return fail("ttt_hip: gate_forward: bad dimensions");
    if (!res || !y || !tanh_text || !tanh_video || !out) return fail("ttt_hip: gate_forward: null pointer");
    ttt::prepost::GateArgs a = {(const __bf16*)res, (const __bf16*)y, tanh_text, tanh_video, (__bf16*)out, B, L, D, n_text};
    ttt::prepost::gate_forward(a, (hipStream_t)stream);
    return post_launch("gate_forward");
}
int ttt_hip_gate_backward_partials(int D) { return ttt::prepost::gate_backward_partials(D); }
int ttt_hip_gate_backward(int B, int L, int D, int n_text, const void* g, const void* y, const float* tanh_text,
                          const float* tanh_video, void* dy, float* dtanh_part, void* stream) {
    if (B <= 0 || L <= 0 || D <= 0 || D % 8) return fail("ttt_hip: gate_backward: bad dimensions");
    if (!g || !y || !tanh_text || !tanh_video || !dy || !dtanh_part) return fail("ttt_hip: gate_backward: null pointer");
    ttt::prepost::GateBwdArgs a = {(const __bf16*)g, (const __bf16*)y, tanh_text, tanh_video, (__bf16*)dy, dtanh_part, B, L, D, n_text};
    ttt::prepost::gate_backward(a, (hipStream_t)stream);
    return post_launch("gate_backward");
}


static int check_attn_tensor(const ttt_attn_tensor& t, const char* name) {
    if (!t.ptr) return fail("ttt_hip: attention: null tensor %s", name);
    if (((uintptr_t)t.ptr & 15) || (t.stride_b & 7) || (t.stride_h & 7) || (t.stride_s & 7))
        return fail("ttt_hip: attention: tensor %s must be 16-byte aligned with strides that are multiples of 8 elements", name);
    return 0;
}

int ttt_hip_attn_forward(const ttt_attn_fwd_args* a, void* stream) {
    if (!a) return fail("ttt_hip: null args");
    if (a->D != 64) return fail("ttt_hip: attention: head_dim must be 64");
    if (a->B <= 0 || a->NH <= 0 || a->S <= 0) return fail("ttt_hip: attention: non-positive dimension");
    if (check_attn_tensor(a->Q, "Q") || check_attn_tensor(a->K, "K") || check_attn_tensor(a->V, "V") || check_attn_tensor(a->O, "O")) return -1;
    ttt::attn::FwdParams p = {};
    p.Q = (const __bf16*)a->Q.ptr; p.K = (const __bf16*)a->K.ptr; p.V = (const __bf16*)a->V.ptr; p.O = (__bf16*)a->O.ptr;
    p.LSE = a->LSE;
    p.q_sb = a->Q.stride_b; p.q_sh = a->Q.stride_h; p.q_ss = a->Q.stride_s;
    p.k_sb = a->K.stride_b; p.k_sh = a->K.stride_h; p.k_ss = a->K.stride_s;
    p.v_sb = a->V.stride_b; p.v_sh = a->V.stride_h; p.v_ss = a->V.stride_s;
    p.o_sb = a->O.stride_b; p.o_sh = a->O.stride_h; p.o_ss = a->O.stride_s;
    p.B = a->B; p.NH = a->NH; p.S = a->S; p.scale = a->scale;
    ttt::attn::launch_forward(p, (hipStream_t)stream);
    return post_launch("attn_forward");
}

int ttt_hip_attn_backward(const ttt_attn_bwd_args* a, void* stream) {
    if (!a) return fail("ttt_hip: null args");
    if (a->D != 64) return fail("ttt_hip: attention: head_dim must be 64");
    if (a->B <= 0 || a->NH <= 0 || a->S <= 0) return fail("ttt_hip: attention: non-positive dimension");
    if (!a->LSE || !a->Delta) return fail("ttt_hip: attention backward: null LSE / Delta");
    if (check_attn_tensor(a->Q, "Q") || check_attn_tensor(a->K, "K") || check_attn_tensor(a->V, "V") || check_attn_tensor(a->O, "O") ||
        check_attn_tensor(a->dO, "dO") || check_attn_tensor(a->dQ, "dQ") || check_attn_tensor(a->dK, "dK") || check_attn_tensor(a->dV, "dV")) return -1;
    ttt::attn::BwdParams p = {};
    p.Q = (const __bf16*)a->Q.ptr; p.K = (const __bf16*)a->K.ptr; p.V = (const __bf16*)a->V.ptr; p.O = (const __bf16*)a->O.ptr;
    p.dO = (const __bf16*)a->dO.ptr; p.dQ = (__bf16*)a->dQ.ptr; p.dK = (__bf16*)a->dK.ptr; p.dV = (__bf16*)a->dV.ptr;
    p.LSE = a->LSE; p.Delta = a->Delta;
    p.q_sb = a->Q.stride_b; p.q_sh = a->Q.stride_h; p.q_ss = a->Q.stride_s;
    p.k_sb = a->K.stride_b; p.k_sh = a->K.stride_h; p.k_ss = a->K.stride_s;
    p.v_sb = a->V.stride_b; p.v_sh = a->V.stride_h; p.v_ss = a->V.stride_s;
    p.o_sb = a->O.stride_b; p.o_sh = a->O.stride_h; p.o_ss = a->O.stride_s;
    p.do_sb = a->dO.stride_b; p.do_sh = a->dO.stride_h; p.do_ss = a->dO.stride_s;
    p.dq_sb = a->dQ.stride_b; p.dq_sh = a->dQ.stride_h; p.dq_ss = a->dQ.stride_s;
    p.dk_sb = a->dK.stride_b; p.dk_sh = a->dK.stride_h; p.dk_ss = a->dK.stride_s;
    p.dv_sb = a->dV.stride_b; p.dv_sh = a->dV.stride_h; p.dv_ss = a->dV.stride_s;
    p.B = a->B; p.NH = a->NH; p.S = a->S; p.scale = a->scale;
    ttt::attn::launch_backward(p, (hipStream_t)stream);
    return post_launch("attn_backward");
}


int ttt_hip_attn_pre_partials(int B, int S, int NH) { return ttt::attn::pre_blocks((long)B * S * NH); }

int ttt_hip_attn_pre_forward(int B, int S, int NH, int n_text, float eps, const void* q_raw, const void* k_raw,
                             const float* wq, const float* bq, const float* wk, const float* bk,
                             const float* cos_table, const float* sin_table, void* q, void* k, void* stream) {
    if (B <= 0 || S <= 0 || NH <= 0 || n_text < 0) return fail("ttt_hip: attn_pre: bad dimension");
    if (!q_raw || !k_raw || !wq || !bq || !wk || !bk || !q || !k) return fail("ttt_hip: attn_pre_forward: null pointer");
    if (n_text < S && (!cos_table || !sin_table)) return fail("ttt_hip: attn_pre_forward: null rope table");
    ttt::attn::PreParams p = {(const __bf16*)q_raw, (const __bf16*)k_raw, wq, bq, wk, bk, cos_table, sin_table,
                              (__bf16*)q, (__bf16*)k, B, S, NH, n_text, eps};
    ttt::attn::launch_pre_forward(p, (hipStream_t)stream);
    return post_launch("attn_pre_forward");
}

int ttt_hip_attn_pre_backward(int B, int S, int NH, int n_text, float eps, const void* q_raw, const void* k_raw,
                              const ttt_attn_tensor* dq, const ttt_attn_tensor* dk, const float* wq, const float* wk,
                              const float* cos_table, const float* sin_table, void* dq_raw, void* dk_raw, float* part,
                              void* stream) {
    return ttt_hip_attn_pre_backward_ld(B, S, NH, n_text, eps, q_raw, k_raw, dq, dk, wq, wk, cos_table, sin_table, dq_raw, dk_raw,
                                        (int64_t)NH * 64, part, stream);
}
int ttt_hip_attn_pre_backward_ld(int B, int S, int NH, int n_text, float eps, const void* q_raw, const void* k_raw,
                                 const ttt_attn_tensor* dq, const ttt_attn_tensor* dk, const float* wq, const float* wk,
                                 const float* cos_table, const float* sin_table, void* dq_raw, void* dk_raw, int64_t ld_out, float* part,
                                 void* stream) {
    if (ld_out < (int64_t)NH * 64 || (ld_out & 7) || ((uintptr_t)dq_raw & 15) || ((uintptr_t)dk_raw & 15))
        return fail("ttt_hip: attn_pre_backward: the raw gradients' row stride must be >= NH*64 and a multiple of 8 elements, their bases 16-byte aligned");
    if (B <= 0 || S <= 0 || NH <= 0 || n_text < 0) return fail("ttt_hip: attn_pre: bad dimension");
    if (!q_raw || !k_raw || !dq || !dk || !wq || !wk || !dq_raw || !dk_raw || !part) return fail("ttt_hip: attn_pre_backward: null pointer");
    if (n_text < S && (!cos_table || !sin_table)) return fail("ttt_hip: attn_pre_backward: null rope table");
    if (check_attn_tensor(*dq, "dq") || check_attn_tensor(*dk, "dk")) return -1;
    ttt::attn::PreBwdParams p = {};
    p.q_raw = (const __bf16*)q_raw; p.k_raw = (const __bf16*)k_raw; p.dq = (const __bf16*)dq->ptr; p.dk = (const __bf16*)dk->ptr;
    p.dq_sb = dq->stride_b; p.dq_sh = dq->stride_h; p.dq_ss = dq->stride_s;
    p.dk_sb = dk->stride_b; p.dk_sh = dk->stride_h; p.dk_ss = dk->stride_s;
    p.wq = wq; p.wk = wk; p.cos = cos_table; p.sin = sin_table;
    p.dq_raw = (__bf16*)dq_raw; p.dk_raw = (__bf16*)dk_raw; p.part = part; p.ld_out = (long)ld_out;
    p.B = B; p.S = S; p.NH = NH; p.n_text = n_text; p.eps = eps;
    ttt::attn::launch_pre_backward(p, (hipStream_t)stream);
    return post_launch("attn_pre_backward");
}


static int check_glue_dims(int B, int Lt, int Lv, int D) {
    if (B <= 0 || Lt < 0 || Lv < 0 || Lt + Lv <= 0 || D <= 0 || (D & 7)) return fail("ttt_hip: glue kernels: bad dimension (D must be a multiple of 8)");
    if (D > 8 * 512) return fail("ttt_hip: glue kernels: D too large for one block per token (D / 8 <= 512 threads)");
    return 0;
}
int ttt_hip_adaln_backward_partials(void) { return ttt::prepost::adaln_backward_partials(); }
int ttt_hip_adaln_forward(int B, int Lt, int Lv, int D, float eps, const void* vid, const void* text, const float* w, const float* b,
                          const float* shift, const float* scale1p, void* out, void* stream) {
    if (check_glue_dims(B, Lt, Lv, D)) return -1;
    if ((Lv && !vid) || (Lt && !text) || !w || !b || !shift || !scale1p || !out) return fail("ttt_hip: adaln_forward: null pointer");
    ttt::prepost::AdaLNArgs a = {(const __bf16*)vid, (const __bf16*)text, w, b, shift, scale1p, (__bf16*)out, B, Lt, Lv, D, eps};
    ttt::prepost::adaln_forward(a, (hipStream_t)stream);
    return post_launch("adaln_forward");
}
int ttt_hip_adaln_backward(int B, int Lt, int Lv, int D, float eps, const void* vid, const void* text, const void* dout,
                           const float* w, const float* b, const float* scale1p, void* dvid, void* dtext, float* part, void* stream) {
    if (check_glue_dims(B, Lt, Lv, D)) return -1;
    if ((Lv && (!vid || !dvid)) || (Lt && (!text || !dtext)) || !dout || !w || !b || !scale1p || !part) return fail("ttt_hip: adaln_backward: null pointer");
    ttt::prepost::AdaLNBwdArgs a = {(const __bf16*)vid, (const __bf16*)text, (const __bf16*)dout, w, b, scale1p,
                                    (__bf16*)dvid, (__bf16*)dtext, part, B, Lt, Lv, D, 0, eps};
    ttt::prepost::adaln_backward(a, (hipStream_t)stream);
    return post_launch("adaln_backward");
}
int ttt_hip_resgate_backward_partials(int D) { return ttt::prepost::resgate_backward_partials(D); }
int ttt_hip_resgate_forward(int B, int Lt, int Lv, int D, const void* vid, const void* text, const void* y, const float* gate,
                            void* ovid, void* otext, void* stream) {
    if (check_glue_dims(B, Lt, Lv, D)) return -1;
    if ((Lv && (!vid || !ovid)) || (Lt && (!text || !otext)) || !y || !gate) return fail("ttt_hip: resgate_forward: null pointer");
    ttt::prepost::ResGateArgs a = {(const __bf16*)vid, (const __bf16*)text, (const __bf16*)y, gate, (__bf16*)ovid, (__bf16*)otext, B, Lt, Lv, D};
    ttt::prepost::resgate_forward(a, (hipStream_t)stream);
    return post_launch("resgate_forward");
}
int ttt_hip_resgate_backward(int B, int Lt, int Lv, int D, const void* dvid, const void* dtext, const void* y, const float* gate,
                             void* dy, float* dgate_part, void* stream) {
    if (check_glue_dims(B, Lt, Lv, D)) return -1;
    if ((Lv && !dvid) || (Lt && !dtext) || !y || !gate || !dy || !dgate_part) return fail("ttt_hip: resgate_backward: null pointer");
    ttt::prepost::ResGateBwdArgs a = {(const __bf16*)dvid, (const __bf16*)dtext, (const __bf16*)y, gate, (__bf16*)dy, dgate_part, B, Lt, Lv, D};
    ttt::prepost::resgate_backward(a, (hipStream_t)stream);
    return post_launch("resgate_backward");
}

}  // extern "C"
