// Host-side entry points of the MFMA (bf16 matrix-core) kernels: CS=64, F=64, bf16 activations.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/ttt_hip.h"

namespace ttt {
namespace mfma {
bool   supports(const ttt_dims* d, bool mlp, bool backward);
size_t workspace_bytes(const ttt_dims* d, bool mlp, bool backward);
void set_debug_timing(void* device_buffer_16_u64);
void set_debug_groups_per_chunk(int groups);   // 0 = automatic
void set_debug_dump(float* device_buffer);     // revision-2 forward: intermediates of workgroup 0, step 0 (>= 120000 floats)
void mlp_forward(const ttt_dims* d, const ttt_mlp_fwd_args* a, void* ws, hipStream_t s);
void mlp_forward_chunk(const ttt_dims* d, const ttt_mlp_fwd_args* a, int step0, int nsteps, float* W1f, float* b1f, float* W2f, float* b2f,
                       void* ws, hipStream_t s);          // CS = 64 only; ws: the forward workspace or null (one workgroup per (b,h))
int  mlp_backward(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, hipStream_t s);   // 0, or < 0: no kernel was launched
void linear_forward(const ttt_dims* d, const ttt_linear_fwd_args* a, void* ws, hipStream_t s);
void linear_backward(const ttt_dims* d, const ttt_linear_bwd_args* a, void* ws, hipStream_t s);
}  // namespace mfma
}  // namespace ttt
