// MFMA TTT-MLP forward scan for mini-batches of 16 tokens (gfx950): the evaluation / sampling geometry of the reference
// (configs/eval/ttt-mlp/*.toml: mini_batch_size = 16, no scan checkpoints; ttt_layer.py:429-473 -> mlp_tk.py forward).
//
// Same primal-form step as the CS = 64 kernel (ttt_mfma2.hip) but a different machine mapping: with
// 16 tokens the products that carry a token dimension are 16 wide, so everything runs on the 16x16 MFMA shapes
//   mma32 = v_mfma_f32_16x16x32_bf16  (contractions over the 64 features / the hidden units),
//   mma16 = v_mfma_f32_16x16x16_bf16  (contractions over the 16 tokens: the state updates),
// and the state lives in 16x16 fp32 accumulator tiles (lane (g, i) = (l >> 4, l & 15) holds D[4g + r][i], r = 0..3).
// Layout algebra: a 16x16 tile X (rows = R, lane = C) is, in place,
//   * an mma16 operand contracting over R: lane's k-slot e carries row 4g + e            (identity order),
//   * half of an mma32 operand contracting over R: two tiles stacked along R give k-slot (g, e) = row 4g + e of the first
//     (e < 4) or of the second (e >= 4) tile ("rho" order); the partner operand presents the same order from a row-major
//     LDS tile with two 8-byte reads at columns c0 + 4g and c0 + 16 + 4g.
// Contraction over the LANE index goes through LDS: the wave writes its tile as an image [lane index][row index] (one
// 8-byte store per lane) and reads it back with ds_read_b64_tr_b16 (private region, no barrier).
//
// Work split: 8 waves; wave w owns hidden units Hw = [32w, 32w + 32): W1[:, Hw] (tiles rows = f, lane = n), W2[Hw, :]
// (rows = n, lane = f) and a second accumulator copy W2^T[:, Hw] (rows = f, lane = n) for the contraction over f in
// gX2 = gZ2 W2^T.  Layer 1 is local to the wave; the two contractions over the hidden units leave 8 fp32 partials per
// element in LDS, which "owner" threads (16 lanes x 4 features per token) reduce.  Per step i, B* = workgroup barriers:
//   A1  Z1 = K W1 + b1 -> X2 = gelu, D1 = gelu'   (rows = t, lane = n) ; X2 image
//   A2  partial Z2^T[f, t] = W2[Hw, :]^T X2[:, Hw]^T -> redA
//   B1
//   P3  waves 0-3: sum partials + b2, fused LayerNorm / L2 backward, Gs = -eta gZ2 -> LDS [t][f] bf16
//   P6  waves 4-7, for step i-1 (off the critical path): sum the redB partials + b2, LayerNorm, + Q -> XQW
//       all: park the inputs of step i+1 (loaded one step earlier)
//   B2
//   C   b2 += colsum Gs (ones MFMA) ; W2 += X2^T Gs ; W2^T += Gs^T X2 ; gX2s = Gs W2^T (entering W2) ; gZ1s = gX2s * D1 ;
//       W1 += K^T gZ1s ; b1 += colsum gZ1s ; Z1b = Q W1' + b1' ; X2b = gelu ; X2b image
//   E   partial Z2b^T = W2'[Hw, :]^T X2b^T -> redB
// Two barriers per step are enough because every shared buffer has one writer phase and one reader phase on opposite sides
// of a barrier: redA (A2 | P3), Gs (P3 | C), redB (E | P6 of the next step), b2 in LDS (C | P3, P6), K / V double-buffered
// and Q triple-buffered (parked between B1 and B2 of the step before they are used; Q of step i-1 is still being read by
// P6 at that time, hence the third buffer).
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"
#define TTT_WV_FN __device__ __forceinline__
#include "ttt_lin16_body.h"
#include "ttt_mlp16_body.h"
#include "once_per_device.h"

namespace ttt {
namespace mfma {
using namespace ttt::mf;

namespace v16 {

constexpr int NT16 = 512;
constexpr int CS16 = 16;
constexpr int TILE16 = CS16 * TS;                     // elements of a padded [16][64] bf16 tile
constexpr int IS = 24;                                // row stride of a wave's [32 n][16 t] image (4 rows on disjoint banks)
constexpr int L_K = 0;                                // K, V: 2 buffers each, Q: 3
constexpr int L_V = L_K + 2 * TILE16 * 2;
constexpr int L_Q = L_V + 2 * TILE16 * 2;
constexpr int L_G = L_Q + 3 * TILE16 * 2;
constexpr int L_IMG = L_G + TILE16 * 2;
constexpr int IMG_BYTES = 32 * IS * 2;
constexpr int L_REDA = L_IMG + 8 * IMG_BYTES;
constexpr int RED16_BYTES = 8 * CS16 * PS * 4;        // [8 waves][16 t][PS] fp32
constexpr int L_REDB = L_REDA + RED16_BYTES;
constexpr int L_SMALL = L_REDB + RED16_BYTES;         // eta[2][16], b2[64], gamma[64], beta[64]
constexpr int LDS_V16 = L_SMALL + (32 + 64 + 64 + 64) * 4;
static_assert(LDS_V16 <= 160 * 1024, "LDS budget");
static_assert(L_IMG % 16 == 0 && L_REDA % 16 == 0 && L_SMALL % 16 == 0, "alignment");

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mma32(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma16(bf16x4 a, bf16x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }
__device__ __forceinline__ bf16x4 pack4(f32x4 v) {
    bf16x4 r = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    return r;
}
// two tiles stacked along their row index -> one K = 32 operand in rho order
__device__ __forceinline__ bf16x8 stack(f32x4 a, f32x4 b) {
    bf16x8 r = {(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
    return r;
}
// rho-order operand from this lane's row of a row-major tile: columns c0 + 4g .. +3 and c0 + 16 + 4g .. +3
__device__ __forceinline__ bf16x8 rho_read(const __bf16* rowp, int c0, int g) {
    const bf16x4 lo = *reinterpret_cast<const bf16x4*>(rowp + c0 + 4 * g);
    const bf16x4 hi = *reinterpret_cast<const bf16x4*>(rowp + c0 + 16 + 4 * g);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// transposed read: lane (g, i) gets img[row0 + 4g + e][col0 + i], e = 0..3  (operand with outer = column, k = row)
__device__ __forceinline__ bf16x4 tr4(const __bf16* img, int stride, int row0, int col0, int l) {
    const int g = l >> 4, i = l & 15;
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + (row0 + 4 * g + (i >> 2)) * stride + col0 + 4 * (i & 3)));
}

template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum16(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror
    return v;
}

// (The hand-placed 8-wave kernel of round 1, mlp_scan16_kernel, lost its round-2 A/B against the backend-templated body of
// ttt_mlp16_body.h - 3.006 vs 2.924 ms per scan at NH = 48, NC = 1128, batch 2 - and was removed; the body is also what the CPU
// suite runs on the wave emulator.)

// ---------------------------------------------------------------------------------------------------------------------------
// TTT-Linear at mini-batches of 16 tokens (the reference trains and evaluates TTT-Linear at mini_batch_size 16:
// configs/train/ttt-linear/*.toml; replaces the Triton launches linear_triton.py:98-129, :203-246).  The kernel bodies live
// in ttt_lin16_body.h, written against a wave backend so that the CPU test-suite can execute the same code on a lane-level
// emulator (tests/emul); here is the device backend: the gfx950 instructions behind those primitives.  One wave per (b, h),
// 4 waves (4 heads) per workgroup only to fill the CU's four SIMDs; no workgroup barriers.
struct DeviceWave {
    char* base;                                                       // this wave's private LDS region
    __device__ __forceinline__ int lane() const { return threadIdx.x & 63; }
    __device__ __forceinline__ int wave() const { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    __device__ __forceinline__ int thread() const { return threadIdx.x; }
    __device__ __forceinline__ void barrier() const { __syncthreads(); }
    __device__ __forceinline__ float exp2(float x) const { return __builtin_amdgcn_exp2f(x); }
    __device__ __forceinline__ float rcp(float x) const { return __builtin_amdgcn_rcpf(x); }
    __device__ __forceinline__ int opaque(int v) const { asm volatile("" : "+v"(v)); return v; }
    __device__ __forceinline__ void lds_fence() const { asm volatile("" ::: "memory"); }    // LDS is in order within a wave
    template <class T> __device__ __forceinline__ T& lds(int byte_off) const { return *reinterpret_cast<T*>(base + byte_off); }
    __device__ __forceinline__ char* lds_ptr(int byte_off) const { return base + byte_off; }
    template <class T> __device__ __forceinline__ T lds_load(int byte_off) const { return *reinterpret_cast<const T*>(base + byte_off); }
    template <class T> __device__ __forceinline__ void lds_store(int byte_off, T v) const { *reinterpret_cast<T*>(base + byte_off) = v; }
    __device__ __forceinline__ float rsq(float x) const { return __builtin_amdgcn_rsqf(x); }
    __device__ __forceinline__ f32x4 mma32(bf16x8 a, bf16x8 b, f32x4 c) const { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
    __device__ __forceinline__ f32x4 mma16(bf16x4 a, bf16x4 b, f32x4 c) const {
        return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a), __builtin_bit_cast(s16x4, b), c, 0, 0, 0);
    }
    __device__ __forceinline__ bf16x4 tr_read(int byte_addr) const {
        typedef __attribute__((address_space(3))) bf16x4 lds_b4;
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(base + byte_addr));
    }
    __device__ __forceinline__ float sum16(float v) const { return v16::sum16(v); }
    __device__ __forceinline__ float xor_add(float v, int mask) const { return v + __shfl_xor(v, mask, 64); }
};

constexpr int LIN_WAVES = 4;
constexpr int LDS_LIN = LIN_WAVES * lin16::WAVE_LDS;

__global__ __launch_bounds__(64 * LIN_WAVES) void linear_scan16_kernel(wv::Lin16Params p, int n_bh) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int bh = blockIdx.x * LIN_WAVES + w;
    if (bh >= n_bh) return;                                          // whole wave; no barriers in this kernel
    DeviceWave bk{smem + w * lin16::WAVE_LDS};
    lin16::forward(bk, p, bh);
}

// TTT-MLP forward scan at mini-batches of 16: the backend-templated workgroup body of ttt_mlp16_body.h
__global__ __launch_bounds__(NT16) void mlp_scan16_body_kernel(wv::Mlp16Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DeviceWave bk{smem};
    mlp16::forward(bk, p, blockIdx.x);
}

// backward: one wave per workgroup (48 .. 96 scans on 256 CUs: a CU of its own per scan; up to 512 registers per lane)
__global__ __launch_bounds__(64) void linear_bwd16_kernel(wv::Lin16Params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    DeviceWave bk{smem};
    lin16::backward(bk, p, blockIdx.x);
}

}  // namespace v16

static void lin_attr_once() {
    static ttt::OncePerDevice done;
    done.run([&] {
        (void)hipFuncSetAttribute((const void*)v16::linear_scan16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, v16::LDS_LIN);
        (void)hipFuncSetAttribute((const void*)v16::linear_bwd16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lin16::WAVE_LDS_BWD);
    });
}
void launch_linear_forward_cs16(const wv::Lin16Params& p, int n_bh, hipStream_t s) {
    lin_attr_once();
    const int blocks = (n_bh + v16::LIN_WAVES - 1) / v16::LIN_WAVES;
    hipLaunchKernelGGL(v16::linear_scan16_kernel, dim3(blocks), dim3(64 * v16::LIN_WAVES), v16::LDS_LIN, s, p, n_bh);
}
void launch_linear_backward_cs16(const wv::Lin16Params& p, int n_bh, hipStream_t s) {
    lin_attr_once();
    hipLaunchKernelGGL(v16::linear_bwd16_kernel, dim3(n_bh), dim3(64), lin16::WAVE_LDS_BWD, s, p);
}

void launch_scan_forward_cs16(const ScanParams& p, int n_bh, unsigned long long*, hipStream_t s) {
    static ttt::OncePerDevice done;
    done.run([&] {
        (void)hipFuncSetAttribute((const void*)v16::mlp_scan16_body_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, mlp16::GROUP_LDS);
    });
    wv::Mlp16Params q = {};
    q.XQ = p.XQ; q.XK = p.XK; q.XV = p.XV; q.eta = p.eta; q.ln_w = p.ln_w; q.ln_b = p.ln_b;
    q.W1 = p.W1; q.b1 = p.b1; q.W2 = p.W2; q.b2 = p.b2; q.W1c = p.W1c; q.b1c = p.b1c; q.W2c = p.W2c; q.b2c = p.b2c;
    q.out = p.out; q.NH = p.NH; q.NC = p.NC; q.G = p.G; q.K = p.K; q.eps = p.eps;
    hipLaunchKernelGGL(v16::mlp_scan16_body_kernel, dim3(n_bh), dim3(v16::NT16), mlp16::GROUP_LDS, s, q);
}

}  // namespace mfma
}  // namespace ttt
