// Host-side entry points of the generic (fp32 VALU) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ttt_hip.h"

namespace ttt {
namespace generic {
bool   supports(const ttt_dims* d);
size_t workspace_bytes(const ttt_dims* d, bool mlp);
void mlp_forward(const ttt_dims* d, const ttt_mlp_fwd_args* a, void* ws, hipStream_t s);
void mlp_backward(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, hipStream_t s);
void linear_forward(const ttt_dims* d, const ttt_linear_fwd_args* a, void* ws, hipStream_t s);
void linear_backward(const ttt_dims* d, const ttt_linear_bwd_args* a, void* ws, hipStream_t s);
}  // namespace generic
}  // namespace ttt
