// TTT-MLP backward, revision 4 (round 3): the SLIM step record and the interfaces between its three kernels.
//
// Round 2 stored, per scan step and (b,h), 570 KiB of re-materialised intermediates as MFMA register images (ttt_mfma_dev.h:
// 16 fragment arrays incl. per-step W1 / W2 / W2^T images and four second-orientation copies) and moved 25x the algorithmic
// bytes through HBM.  Revision 4 keeps per step only what a hidden-unit slice cannot rebuild locally:
//   written by the group recompute (phase A, ttt_mfma_rc4.hip):
//     A_Z1, A_Z1B   the pre-activations Z1 = K W1 + b1 and Z1b = Q W1' + b1' (bf16, 32 KiB each)
//     gZ2 tile      (bf16, 8 KiB)            owner rows  x_hat, y - target, x_hat of the output LayerNorm (fp32) + 2 row stats
//   written by the reverse sweep (phase B, ttt_mfma_bwd4.hip) for the dK / dQ tail (phase C):
//     A_DZ1, A_DZ1B (compute waves), A_GZ1 (the sweep's DERIVER waves)
// X2 = gelu(Z1), gelu'(Z1), gelu''(Z1), X2b, gelu'(Z1b), gX2 = gZ2 W2^T, gZ1, M = gX2 gelu''(Z1) and every second orientation
// are re-derived inside the sweep by two deriver waves per workgroup, and the per-step W2 is obtained by REVERSING the state update
// in fp32,  W2_i = W2_{i+1} + (eta X2_i)^T gZ2_i , re-anchored at every forward checkpoint (the state after the last step of the
// sequence is written by phase A).  Round 6: the per-step W1 ( W1_i = W1_{i+1} + (eta K_i)^T gZ1_i ) and dW1' are needed by the tail
// only, and the tail - one workgroup per checkpoint group, sequential over its steps - rebuilds both from one anchor per group
// itself (rounds 3 - 5 stored a packed image of each per step: A_DW1, A_W1, 64 KiB).  216.5 KiB per step, of which phase A writes
// 104.5 KiB (the inner LayerNorm's owner rows as bf16); the sweep reads those once.
//
// Fragment arrays are indexed like the round-2 images (fr_idx(a, b, s), 8 fragments of 1 KiB = 64 lanes x 16 B per hidden slice
// q = wave pair of the 8-wave decomposition; T = tile (rows = t, lane = n)):
//   A_Z1   [ti][nj][s]  Z1, T          A_Z1B  [ti][nj][s]  Z1b, T
//   A_DZ1  [ti][nj][s]  dZ1, T         A_DZ1B [ti][nj][s]  dZ1b, T       A_GZ1 [ti][nj][s]  gZ1, T
#pragma once
#include "ttt_mfma_dev.h"
#include "ttt_mfma_bwd_dev.h"

namespace ttt {
namespace mfma {
namespace s4 {
using namespace ttt::mf;

enum { A_Z1 = 0, A_Z1B, A_DZ1, A_DZ1B, A_GZ1, A_COUNT };
constexpr size_t SLICE_BYTES = (size_t)A_COUNT * 8 * FRAG_BYTES;             // 40 KiB per hidden slice
constexpr size_t SLOT4_FR = 4 * SLICE_BYTES;                                 // 160 KiB
constexpr size_t SLOT4_OWN = SLOT_OWN;                                       // three fp32 [64][64] owner arrays + 64 x (rstd, rstd_out)
constexpr size_t SLOT4_G = SLOT_G;                                           // gZ2 tile, bf16 row-major [t][f]
constexpr size_t SLOT4_BYTES = SLOT4_FR + SLOT4_OWN + SLOT4_G;               // 216.5 KiB
constexpr int fro4(int arr, int idx) { return (arr * 8 + idx) * (int)FRAG_BYTES; }      // byte offset of a fragment inside a slice region

// state after the last step of the sequence (phase A's last workgroup of a (b,h) writes it; the anchor of the topmost chunk):
// natural layouts W1 [64][256], W2 [256][64], fp32
constexpr size_t FINAL_FLOATS = 2 * 64 * 256;

struct RecomputeParams {
    const __bf16 *XQ, *XK, *XV, *eta;
    const float *ln_w, *ln_b;
    const float *W1c, *b1c, *W2c, *b2c;    // forward checkpoints (state entering steps 0, G, 2G, ...)
    char* slots; size_t slot_stride_bh;    // slot s of (b,h) <-> step chunk_lo + s
    float* wfinal;                         // [B NH][FINAL_FLOATS]
    int NH, NC, G, K;
    int chunk_group0, chunk_groups, chunk_lo;
    int nt;                                // 1: non-temporal stores of the step records
    int own16;                             // 1: the owner rows x_hat and y - target of the INNER LayerNorm as bf16 (first half of their fp32 arrays' space)
    int item0;                             // (set by the launcher) first work item of this launch; item = item0 + blockIdx.x
    float eps;
};
// max_workgroups > 0: the B NH chunk_groups work items are covered by consecutive launches of at most that many workgroups
// (beside a cluster sweep: never more workgroups in flight than CUs are free)
void launch_recompute4(const RecomputeParams& p, int n_bh, int max_workgroups, hipStream_t s);

// the cluster sweep of revision 4: revision 3's parameters + what the deriver waves need to anchor the reversed state update
struct SweepParams4 : b2::SweepParams2 {
    const float *W1c, *W2c;                // forward checkpoints [B NH][K][64][256], [B NH][K][256][64]
    const float* wfinal;                   // [B NH][FINAL_FLOATS]: the state after the last step of the sequence (phase A)
    char* park;                            // [B NH][4 workgroups][2 deriver waves][PARK4_BYTES]: R4 fragments between derivation and staging
    float* danchor;                        // [B NH][K][64][256] fp32: dW1 entering the top step of every checkpoint group (the tail's anchor)
    int G, K;
    int prefetch;                          // 1 (always, since round 5): owners / derivers touch the records of step i - 2 (L2 prefetch)
    int own16;                             // as RecomputeParams::own16 (both kernels of a backward call agree)
    int split;                             // 1 (round 6): barrier Bc INSIDE the derivers' reverse step, behind its W2 update; 0: behind the step
};
constexpr size_t PARK4_BYTES = 12 * FRAG_BYTES;
void launch_sweep_cluster4(const SweepParams4& bp, int nbh, hipStream_t s);

// the group-sequential tail (mlp_bwd_tail5_kernel): one workgroup per (b, h, checkpoint group of the chunk)
struct Tail5Args {
    const __bf16 *XQ, *XK, *dOut, *eta, *dXV;
    char* slots; size_t slot_stride_bh;
    const float *W1c, *wfinal, *danchor;
    __bf16 *dXQ, *dXK;
    int NC, G, K, chunk_lo, group0, ngroups;
};
void launch_tail5(const Tail5Args& a, int nbh, hipStream_t s);

}  // namespace s4
}  // namespace mfma
}  // namespace ttt
