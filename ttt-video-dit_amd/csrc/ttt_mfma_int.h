// Internal interface between the MFMA translation units (scan/recompute kernel and reverse sweep).
#pragma once
#include "ttt_wave_types.h"
#include <hip/hip_runtime.h>
#include <stddef.h>
#include "../../include/ttt_hip.h"

namespace ttt {
namespace mfma {

struct ScanParams {
    const __bf16 *XQ, *XK, *XV, *eta;
    const float *ln_w, *ln_b;
    const float *W1, *b1, *W2, *b2;        // initial state (forward only)
    float *W1c, *b1c, *W2c, *b2c;          // checkpoints: written by the forward, read by the recompute
    __bf16* out;                           // XQW (forward only)
    int NH, NC, G, K;
    float eps;
    // a launch over a PART of the sequence (round 5, ttt_hip_mlp_forward_chunk): NC steps starting at a checkpoint-group boundary.
    // The tile pointers are pre-offset to the first step; consecutive heads are NCs tiles apart (0: NC), checkpoint index ck0 + i / G,
    // the state after the last step goes to W1f .. b2f ([B,NH,...] like the initial state, may alias it; null: not stored).
    int NCs, ck0;
    float *W1f, *b1f, *W2f, *b2f;
    unsigned long long* dbg;               // optional per-phase cycle totals of workgroup 0
    float* dump;                           // DEBUG: intermediates of workgroup 0, step 0 (revision-2 forward)
};

bool bwd_available();
int groups_per_chunk(const ttt_dims* d);
// revision-2 forward scan (ttt_mfma2.hip): 8 waves per (b,h), VGPR-form MFMA, LDS transposed reads
// `ws`: the caller's forward workspace (scan_pair_workspace_bytes) or null; with it, the scan runs as a PAIR of workgroups per (b,h)
// (round 6: role A carries the state, role B runs the output path from the records A publishes - ttt_mfma2.hip)
void launch_scan_forward_v2(const ScanParams& p, int n_bh, void* ws, unsigned long long* dbg, hipStream_t s);
size_t scan_pair_workspace_bytes(int n_bh);
void set_debug_scan_pair(int v);      // 1 (default) pair form / 0 one workgroup per (b,h)
void set_debug_scan_fault(int v);     // DEBUG fault injection: role B of every pair leaves at once
int device_cu_count();                // compute units of the current device (cached)
int get_debug_fast_records();
// mini-batches of 16 tokens, forward only (ttt_mfma16.hip): the evaluation / sampling geometry
void launch_scan_forward_cs16(const ScanParams& p, int n_bh, unsigned long long* dbg, hipStream_t s);
void launch_linear_forward_cs16(const wv::Lin16Params& p, int n_bh, hipStream_t s);   // TTT-Linear, one wave per (b,h)
void launch_linear_backward_cs16(const wv::Lin16Params& p, int n_bh, hipStream_t s);
void set_debug_dump(float* buf);
unsigned long long* get_debug_timing();
void set_debug_overlap_tail(int v);   // backward schedule: 0 one stream, 1 tail of chunk c beside the sweep of chunk c-1, 2 (default) the next recompute too
void set_debug_fast_records(int v);   // cluster sweep: 1 (default) plain records on a proven common XCD, 0 write-through always
unsigned read_sweep_error();           // 0, or 1 + (b,h) of a cluster workgroup whose partner never arrived (synchronises)
unsigned peek_sweep_error();           // the same word without synchronising (entry check of the TTT-MLP calls)
void clear_sweep_error();              // acknowledge (synchronises)
unsigned* sweep_error_word();          // device pointer of the host-mapped word (allocated on first use; nullptr on failure)
void set_debug_deriver_split(int v);  // sweep: 1 (default, round 6) barrier Bc inside the derivers' reverse step, 0 = behind it
void set_debug_sweep_fault(int v);     // DEBUG fault injection: workgroup 3 of every sweep cluster leaves before its first hand-over
unsigned read_sweep_fast_count();   // DEBUG statistic (synchronises)

}  // namespace mfma
}  // namespace ttt
