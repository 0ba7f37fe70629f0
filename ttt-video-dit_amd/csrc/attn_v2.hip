// Segment attention, revision 2 (forward and dQ): the backend-templated workgroup bodies of attn_body.h instantiated with the
// gfx950 instructions.  Opt-in (debug option "attn_variant" = 2) until it has been timed against revision 1 on an MI355X; the
// bodies themselves are executed on the CPU by the wave emulator against the fp64 oracle (tests/test_emul_cpu.py), and
// tests/test_attention_gpu.py compares the two revisions bit for bit on the device.
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include "../../include/ttt_hip.h"
#include "attn.h"
#include "attn_body.h"
#include "once_per_device.h"

namespace ttt {
namespace attn {

struct AttnDeviceWave {
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

template <int W>
__global__ __launch_bounds__(512, W) void attn_dq2_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bh, qb;
    attnb::head_of_block(blockIdx.x, (p.S + attnb::QB - 1) / attnb::QB, p.B * p.NH, bh, qb);
    AttnDeviceWave bk{reinterpret_cast<__bf16*>(smem)};
    attnb::dq(bk, p, bh, qb);
}

// dK / dV through the body of attn_body.h: <8 waves, plain> restates revision 1 (bit-identical), <8, ACC_INIT> starts the
// accumulators from the per-row -LSE / scale and -Delta (32 registers and 32 VALU per tile fewer), <12, ACC_INIT> uses
// the freed registers for a third wave per SIMD
template <int NW, bool ACC_INIT, int MINW>
__global__ __launch_bounds__(64 * NW, MINW) void attn_dkdv2_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bh, kvb;
    attnb::head_of_block(blockIdx.x, (p.S + 32 * NW - 1) / (32 * NW), p.B * p.NH, bh, kvb);
    AttnDeviceWave bk{reinterpret_cast<__bf16*>(smem)};
    attnb::dkdv<NW, ACC_INIT>(bk, p, bh, kvb);
}

// SEVERAL tiles of 64 per LDS stage (attn_body.h dq_staged / dkdv_staged): one workgroup barrier per NSUB tiles instead of one per
// tile, the same arithmetic in the same order - bit-identical to the one-tile kernels (emulator: tests/test_emul_attention_cpu.py,
// device: tests/test_attention_gpu.py::test_attention_tiles_per_stage_equal_the_one_tile_kernels).  Round 4, one MI355X,
// 48 heads x S = 18 048 (profiles/r4a_attn_backward_stages_ab.log): two tiles per stage 13.51 ms per backward against 14.17 ms
// (one tile); the XOR-swizzled unpadded tiles of round 3 - conflict-free under the bank model - LOST on the device (15.08 ms
// alone, 15.02 ms with two tiles: the extra address arithmetic costs more than the replays it removes) and their device
// instantiations are gone (the body keeps the template parameter for the emulator's bank-model tests).
template <int W, int NSUB>
__global__ __launch_bounds__(512, W) void attn_dq2s_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bh, qb;
    attnb::head_of_block(blockIdx.x, (p.S + attnb::QB - 1) / attnb::QB, p.B * p.NH, bh, qb);
    AttnDeviceWave bk{reinterpret_cast<__bf16*>(smem)};
    attnb::dq_staged<NSUB, false>(bk, p, bh, qb);
}
template <int NW, bool ACC_INIT, int MINW, int NSUB>
__global__ __launch_bounds__(64 * NW, MINW) void attn_dkdv2s_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bh, kvb;
    attnb::head_of_block(blockIdx.x, (p.S + 32 * NW - 1) / (32 * NW), p.B * p.NH, bh, kvb);
    AttnDeviceWave bk{reinterpret_cast<__bf16*>(smem)};
    attnb::dkdv_staged<NW, ACC_INIT, NSUB, false>(bk, p, bh, kvb);
}
// dQ with 64 query rows per wave (attn_body.h dq_wide: every K / V fragment read from LDS feeds two MFMAs; 2 waves of <= 256
// registers per SIMD, one 8-wave workgroup of 512 rows per CU), NSUB key tiles per stage.  Debug option "attn_dq_wide".
template <int NSUB>
__global__ __launch_bounds__(512, 2) void attn_dq_wide_kernel(BwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int bh, qb;
    attnb::head_of_block(blockIdx.x, (p.S + 2 * attnb::QB - 1) / (2 * attnb::QB), p.B * p.NH, bh, qb);
    AttnDeviceWave bk{reinterpret_cast<__bf16*>(smem)};
    attnb::dq_wide<NSUB, 2>(bk, p, bh, qb);
}
static int g_dq_wide = 1;                // round 4, one box (profiles/r4c_attn_dq_wide_ab.log): 13.26 - 13.44 ms per backward against 13.79
void set_debug_attn_dq_wide(int v) { g_dq_wide = v; }
template <int NSUB>
static void launch_dq_wide(const BwdParams& p, hipStream_t s) {
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_dq_wide_kernel<NSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, NSUB * attnb::LDS_DQ);
    });
    const int nb = (p.S + 2 * attnb::QB - 1) / (2 * attnb::QB);
    hipLaunchKernelGGL((attn_dq_wide_kernel<NSUB>), dim3(p.B * p.NH * nb), dim3(512), NSUB * attnb::LDS_DQ, s, p);
}

// tiles of 64 per LDS stage: dQ 1 / 2 (two workgroups of 73 KiB share a CU), dK / dV 1 .. 4 (one workgroup of 768 threads per CU:
// up to 148 KiB).  Debug option "attn_stage" sets both, "attn_stage_dq" / "attn_stage_dkdv" one of them (A/B).
static int g_stage_dq = 1, g_stage_dkdv = 2;      // (dQ: the wide kernel holds one workgroup per CU; one tile per stage measured best there)
void set_debug_attn_stage(int which, int v) {
    if (which != 2) g_stage_dq = (v >= 1 && v <= 2) ? v : 1;
    if (which != 1) g_stage_dkdv = (v >= 1 && v <= 4) ? v : 2;
}

template <int NSUB>
static void launch_dq_staged(const BwdParams& p, hipStream_t s) {
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_dq2s_kernel<4, NSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, NSUB * attnb::LDS_DQ);
    });
    const int nb = (p.S + attnb::QB - 1) / attnb::QB;
    hipLaunchKernelGGL((attn_dq2s_kernel<4, NSUB>), dim3(p.B * p.NH * nb), dim3(512), NSUB * attnb::LDS_DQ, s, p);
}
void launch_dq_v2(const BwdParams& p, hipStream_t s) {
    if (g_dq_wide) return g_stage_dq == 2 ? launch_dq_wide<2>(p, s) : launch_dq_wide<1>(p, s);
    if (g_stage_dq == 2) return launch_dq_staged<2>(p, s);
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_dq2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, attnb::LDS_DQ);
    });
    const int nb = (p.S + attnb::QB - 1) / attnb::QB;
    hipLaunchKernelGGL(attn_dq2_kernel<4>, dim3(p.B * p.NH * nb), dim3(512), attnb::LDS_DQ, s, p);
}

template <int NSUB>
static void launch_dkdv_staged(const BwdParams& p, hipStream_t s) {
    constexpr int NW = 12;
    static_assert(NSUB * attnb::LDS_DKV <= 160 * 1024, "LDS budget");
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_dkdv2s_kernel<NW, true, 3, NSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, NSUB * attnb::LDS_DKV);
    });
    const int nb = (p.S + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL((attn_dkdv2s_kernel<NW, true, 3, NSUB>), dim3(p.B * p.NH * nb), dim3(64 * NW), NSUB * attnb::LDS_DKV, s, p);
}
void launch_dkdv_v2(const BwdParams& p, hipStream_t s) {       // accumulator-initialised row scalars, 12 waves (3 per SIMD)
    constexpr int NW = 12;
    if (g_stage_dkdv == 2) return launch_dkdv_staged<2>(p, s);
    if (g_stage_dkdv == 3) return launch_dkdv_staged<3>(p, s);
    if (g_stage_dkdv == 4) return launch_dkdv_staged<4>(p, s);
    static ttt::OncePerDevice attr;
    attr.run([&] {
        (void)hipFuncSetAttribute((const void*)attn_dkdv2_kernel<NW, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, attnb::LDS_DKV);
    });
    const int nb = (p.S + 32 * NW - 1) / (32 * NW);
    hipLaunchKernelGGL((attn_dkdv2_kernel<NW, true, 3>), dim3(p.B * p.NH * nb), dim3(64 * NW), attnb::LDS_DKV, s, p);
}

}  // namespace attn
}  // namespace ttt
