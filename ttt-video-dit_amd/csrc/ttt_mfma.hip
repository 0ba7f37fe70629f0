// MFMA kernels - placeholder dispatch (filled in as kernels land).
#include "ttt_mfma.h"
namespace ttt {
namespace mfma {
bool supports(const ttt_dims*, bool, bool) { return false; }
size_t workspace_bytes(const ttt_dims*, bool, bool) { return 0; }
void mlp_forward(const ttt_dims*, const ttt_mlp_fwd_args*, void*, hipStream_t) {}
void mlp_backward(const ttt_dims*, const ttt_mlp_bwd_args*, void*, hipStream_t) {}
void linear_forward(const ttt_dims*, const ttt_linear_fwd_args*, void*, hipStream_t) {}
void linear_backward(const ttt_dims*, const ttt_linear_bwd_args*, void*, hipStream_t) {}
}  // namespace mfma
}  // namespace ttt
