// Segment attention backward (dQ and dK / dV): the backend-templated workgroup bodies of attn_body.h instantiated with the gfx950
// instructions; the bodies themselves are executed on the CPU by the wave emulator against the fp64 oracle
// (tests/test_emul_attention_cpu.py), the kernels on the device against the same oracle (tests/test_attention_gpu.py).
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "../../include/ttt_hip.h"
#include "attn.h"
#include "attn_body.h"
#include "once_per_device.h"

namespace ttt {
namespace attn {

template <int VAR>
struct AttnDeviceWave {
    static constexpr bool kPrio = VAR != 0;                           // s_setprio around one MFMA cluster per kernel (attn_body.h)
    __device__ __forceinline__ void setprio(int v) const { if (v) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
    typedef __bf16* tile_t;                                           // element pointer into the workgroup's LDS
    typedef __attribute__((address_space(3))) wv::bf16x4 lds_b4;
    tile_t base;
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    __device__ __forceinline__ int wave() const { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    __device__ __forceinline__ int thread() const { return threadIdx.x; }
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
    __device__ __forceinline__ float exp2(float x) const { return __builtin_amdgcn_exp2f(x); }
    __device__ __forceinline__ float log(float x) const { return __logf(x); }
    __device__ __forceinline__ tile_t lds_base() const { return base; }
    template <class T> __device__ __forceinline__ T ld(const __bf16* p) const { return *reinterpret_cast<const T*>(p); }
    template <class T> __device__ __forceinline__ void st(__bf16* p, T v) const { *reinterpret_cast<T*>(p) = v; }
    __device__ __forceinline__ wv::bf16x4 tr(const __bf16* p) const { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)p); }
    __device__ __forceinline__ wv::f32x16 mma3216(wv::bf16x8 a, wv::bf16x8 b, wv::f32x16 c) const {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    __device__ __forceinline__ float xor_read(float v, int mask) const { return __shfl_xor(v, mask, 64); }
    __device__ __forceinline__ bool any(bool b) const { return __builtin_amdgcn_ballot_w64(b) != 0; }
};

// dK / dV through the body of attn_body.h with the accumulators started from the per-row -LSE / scale and -Delta (32 registers and 32
// VALU per tile fewer than the plain form) and the freed registers used for a third wave per SIMD (12 waves), NSUB query tiles of 64 per
// LDS stage: one workgroup barrier per NSUB tiles - the same arithmetic in the same order as the one-tile form (emulator:
// tests/test_emul_attention_cpu.py).  Round 4, one MI355X, 48 heads x S = 18 048 (profiles/r4a_attn_backward_stages_ab.log, r4b_*): tiles
// per stage dQ : dK/dV 1:1 14.29 ms per backward, 2:2 13.56, 1:2 13.81, 2:3 13.52, 2:4 14.23; the XOR-swizzled unpadded tiles of round 3
// LOST (15.08 ms).  Shipped: two tiles per stage; the other device instantiations and their options were removed in round 5.
template <int NW, bool ACC_INIT, int MINW, int NSUB, int VAR>
__global__ __launch_bounds__(64 * NW, MINW) void attn_dkdv2s_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bh, kvb;
    attnb::head_of_block(blockIdx.x, (p.S + 32 * NW - 1) / (32 * NW), p.B * p.NH, bh, kvb);
    AttnDeviceWave<VAR> bk{reinterpret_cast<__bf16*>(smem)};
    attnb::dkdv_staged<NW, ACC_INIT, NSUB, false>(bk, p, bh, kvb);
}
// dQ with 64 query rows per wave (attn_body.h dq_wide: every K / V fragment read from LDS feeds two MFMAs; 2 waves of <= 256
// registers per SIMD, one 8-wave workgroup of 512 rows per CU), NSUB key tiles per stage.  Round 4, one box
// (profiles/r4c_attn_dq_wide_ab.log): 13.26 - 13.44 ms per backward against 13.79 for 32 rows per wave; bit-identical (a query row's
// arithmetic and its order over the keys are unchanged); one tile per stage measured best here.
template <int NSUB, int VAR>
__global__ __launch_bounds__(512, 2) void attn_dq_wide_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bh, qb;
    attnb::head_of_block(blockIdx.x, (p.S + 2 * attnb::QB - 1) / (2 * attnb::QB), p.B * p.NH, bh, qb);
    AttnDeviceWave<VAR> bk{reinterpret_cast<__bf16*>(smem)};
    attnb::dq_wide<NSUB, 2>(bk, p, bh, qb);
}

static int g_attn_variant = 1;          // DEBUG A/B (ttt_hip_debug_option "attn_prio"): 1 (default) the priority form of attn_body.h, 0 without
void set_debug_attn_variant(int v) { g_attn_variant = v; }

template <int VAR>
static void launch_dq_var(const BwdParams& p, hipStream_t s) {
    constexpr int NSUB = 1;
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_dq_wide_kernel<NSUB, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, NSUB * attnb::LDS_DQ);
    });
    const int nb = (p.S + 2 * attnb::QB - 1) / (2 * attnb::QB);
    hipLaunchKernelGGL((attn_dq_wide_kernel<NSUB, VAR>), dim3(p.B * p.NH * nb), dim3(512), NSUB * attnb::LDS_DQ, s, p);
}
void launch_dq_v2(const BwdParams& p, hipStream_t s) {
    if (g_attn_variant) launch_dq_var<1>(p, s);
    else launch_dq_var<0>(p, s);
}

template <int VAR>
static void launch_dkdv_var(const BwdParams& p, hipStream_t s) {
    constexpr int NW = 12, NSUB = 2;
    static_assert(NSUB * attnb::LDS_DKV <= 160 * 1024, "LDS budget");
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_dkdv2s_kernel<NW, true, 3, NSUB, VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, NSUB * attnb::LDS_DKV);
    });
    const int nb = (p.S + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL((attn_dkdv2s_kernel<NW, true, 3, NSUB, VAR>), dim3(p.B * p.NH * nb), dim3(64 * NW), NSUB * attnb::LDS_DKV, s, p);
}
void launch_dkdv_v2(const BwdParams& p, hipStream_t s) {
    if (g_attn_variant) launch_dkdv_var<1>(p, s);
    else launch_dkdv_var<0>(p, s);
}

}  // namespace attn
}  // namespace ttt
