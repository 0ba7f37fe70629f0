// Host-side entry points of the MFMA (bf16 matrix-core) TTT kernels for gfx950: geometry check and launch of
//   TTT-MLP forward   CS = 64: ttt_mfma2.hip (8-wave register-resident scan)   CS = 16: ttt_mfma16.hip
//   TTT-MLP backward  CS = 64: revision 4 - ttt_mfma_rc4.hip (group recompute), ttt_mfma_bwd4.hip (cluster sweep with deriver
//                     waves, tail), orchestrated by ttt_mfma_bwd2.hip
//   TTT-Linear forward / backward at CS = 16: ttt_mfma16.hip
// (Round 1's 4-wave scan / recompute kernel lived here; its last user, revision 3 of the backward, lost its round-3 A/B against
// revision 4 - 17.8 vs 13.4 ms per backward at NC = 804, profiles/r3h_* - and was removed with it.)
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"

namespace ttt {
namespace mfma {
using namespace ttt::mf;

static unsigned long long* g_dbg = nullptr;
void set_debug_timing(void* buf) { g_dbg = (unsigned long long*)buf; }
unsigned long long* get_debug_timing() { return g_dbg; }

bool supports(const ttt_dims* d, bool mlp, bool backward) {
    if (!(d->F == 64 && d->act_dtype == TTT_DTYPE_BF16)) return false;
    if (d->CS == 16) return !backward || !mlp;  // mini-batches of 16 (ttt_mfma16.hip): MLP forward, Linear forward + backward
    if (!mlp) return false;
    return d->CS == 64 && (!backward || bwd_available());
}

void mlp_forward(const ttt_dims* d, const ttt_mlp_fwd_args* a, void* ws, hipStream_t s) {
    ScanParams p = {};
    p.XQ = (const __bf16*)a->XQ; p.XK = (const __bf16*)a->XK; p.XV = (const __bf16*)a->XV; p.eta = (const __bf16*)a->last_eta;
    p.ln_w = a->ttt_norm_weight; p.ln_b = a->ttt_norm_bias;
    p.W1 = a->W1_init; p.b1 = a->b1_init; p.W2 = a->W2_init; p.b2 = a->b2_init;
    p.W1c = a->W1_checkpoints; p.b1c = a->b1_checkpoints; p.W2c = a->W2_checkpoints; p.b2c = a->b2_checkpoints;
    p.out = (__bf16*)a->XQW;
    p.NH = d->NH; p.NC = d->NC; p.G = d->G; p.K = (d->NC + d->G - 1) / d->G; p.eps = d->eps;
    if (d->CS == 16) launch_scan_forward_cs16(p, d->B * d->NH, g_dbg, s);
    else launch_scan_forward_v2(p, d->B * d->NH, ws, g_dbg, s);
}
// steps [step0, step0 + nsteps) of the CS = 64 scan: the same kernel over a part of the sequence, started from the state in
// a->*_init (what the previous part left in *_final), its checkpoints written at their places in the whole sequence's arrays
void mlp_forward_chunk(const ttt_dims* d, const ttt_mlp_fwd_args* a, int step0, int nsteps, float* W1f, float* b1f, float* W2f, float* b2f,
                       void* ws, hipStream_t s) {
    ScanParams p = {};
    const size_t t0 = (size_t)step0 * 64 * 64, e0 = (size_t)step0 * 64;
    p.XQ = (const __bf16*)a->XQ + t0; p.XK = (const __bf16*)a->XK + t0; p.XV = (const __bf16*)a->XV + t0; p.eta = (const __bf16*)a->last_eta + e0;
    p.ln_w = a->ttt_norm_weight; p.ln_b = a->ttt_norm_bias;
    p.W1 = a->W1_init; p.b1 = a->b1_init; p.W2 = a->W2_init; p.b2 = a->b2_init;
    p.W1c = a->W1_checkpoints; p.b1c = a->b1_checkpoints; p.W2c = a->W2_checkpoints; p.b2c = a->b2_checkpoints;
    p.out = (__bf16*)a->XQW + t0;
    p.NH = d->NH; p.NC = nsteps; p.G = d->G; p.K = (d->NC + d->G - 1) / d->G; p.eps = d->eps;
    p.NCs = d->NC; p.ck0 = step0 / d->G;
    p.W1f = W1f; p.b1f = b1f; p.W2f = W2f; p.b2f = b2f;
    launch_scan_forward_v2(p, d->B * d->NH, ws, g_dbg, s);
}
void linear_forward(const ttt_dims* d, const ttt_linear_fwd_args* a, void*, hipStream_t s) {
    wv::Lin16Params p = {};
    p.XQ = (const __bf16*)a->XQ; p.XK = (const __bf16*)a->XK; p.XV = (const __bf16*)a->XV; p.eta = (const __bf16*)a->last_eta;
    p.ln_w = a->ttt_norm_weight; p.ln_b = a->ttt_norm_bias;
    p.W1 = a->W1_init; p.b1 = a->b1_init;
    p.W1c = a->W1_checkpoints; p.b1c = a->b1_checkpoints;
    p.out = (__bf16*)a->XQW;
    p.NH = d->NH; p.NC = d->NC; p.G = d->G; p.K = (d->NC + d->G - 1) / d->G; p.eps = d->eps;
    launch_linear_forward_cs16(p, d->B * d->NH, s);
}
void linear_backward(const ttt_dims* d, const ttt_linear_bwd_args* a, void*, hipStream_t s) {
    wv::Lin16Params p = {};
    p.XQ = (const __bf16*)a->XQ; p.XK = (const __bf16*)a->XK; p.XV = (const __bf16*)a->XV; p.eta = (const __bf16*)a->last_eta;
    p.ln_w = a->ttt_norm_weight; p.ln_b = a->ttt_norm_bias;
    p.W1c = const_cast<float*>(a->W1_checkpoints); p.b1c = const_cast<float*>(a->b1_checkpoints);      // read only here
    p.dOut = (const __bf16*)a->grad_L_XQW;
    p.dW1_last = a->grad_L_W1_last; p.db1_last = a->grad_L_b1_last;
    p.scratch_w = (char*)a->W1_init_group; p.scratch_b = a->b1_init_group;      // G x 16 KiB and G x 64 floats per (b,h)
    p.dln_w = a->grad_L_ttt_norm_weight; p.dln_b = a->grad_L_ttt_norm_bias;
    p.dW1 = a->grad_L_W1_init; p.db1 = a->grad_L_b1_init;
    p.deta = (__bf16*)a->grad_L_last_eta; p.dXQ = (__bf16*)a->grad_L_XQ; p.dXK = (__bf16*)a->grad_L_XK; p.dXV = (__bf16*)a->grad_L_XV;
    p.NH = d->NH; p.NC = d->NC; p.G = d->G; p.K = (d->NC + d->G - 1) / d->G; p.eps = d->eps;
    launch_linear_backward_cs16(p, d->B * d->NH, s);
}

}  // namespace mfma
}  // namespace ttt
