// MFMA TTT-MLP backward for gfx950, revision 4 (round 3): the reverse sweep in CLUSTER form - the sweep of one (b,h) runs on
// FOUR workgroups (four CUs) with role-specialised waves - over the SLIM step record of ttt_bwd4_dev.h, and the dK / dQ tail.
//
// Why a cluster (round 2, profiles/r1f, r2a): one workgroup per (b,h) is bound by what ONE CU can pull from memory (a CU sustains
// ~10 bytes / cycle of misses) and by one CU's issue slots; 48 scans occupy 48 of 256 CUs.  Everything the sweep carries or
// loads is sliced by hidden unit: dW1[:, H], dW2[H, :], db1[H] and the step record's fragment arrays.  So workgroup cq of a
// cluster takes hidden slice cq (64 units: a quarter of the record, of the state and of the MFMAs per CU).  The ONLY quantity
// that crosses the slices per step is the partial d(gZ2)^T [64 x 64] fp32 (+ the per-token d(eta) partials): every workgroup
// publishes its partial and reads all four (an all-gather), and all four run the cheap owner stage redundantly with the same
// summation order, so dZ2 is bit-identical on the four CUs and ONE hand-over per step suffices.
//
// Hand-over = the placement-independent recipe of the CDNA4 guide (cdna_hip_programming.md Guideline 16, form R1): payload
// stored write-through (16-byte sc1 buffer stores), every storing wave drains (s_waitcnt vmcnt(0)), workgroup barrier, ONE
// lane stores the step number into the workgroup's flag word (relaxed, agent scope); the consumer polls the three partner
// flags (relaxed, agent scope, one lane each, bounded), then reads the payload with sc1 loads (never served by the reading
// CU's L1).  Nothing depends on where the four workgroups run; when the first hand-over PROVES that they share an XCD (they
// exchange HW_REG_XCC_ID; blocks bh + q nbh do when nbh % 8 == 0) later records are stored plain and stay in that XCD's L2.
// Records are double-buffered by step parity - a workgroup can be at most one hand-over ahead of its slowest partner -, the
// flags are zeroed by a memset node ahead of every launch.  A poll that gives up (a partner that is not running) does not hang
// the GPU and does not return plausible numbers either: it stores 1 + (b,h) into the process's host-mapped error word
// (p.err, system scope), POISONS the workgroup - every later poll returns at once, everything it writes from then on (dV,
// d(eta), the carried / final state gradients, dgamma / dbeta) is NaN - and the next extension call fails on entry
// (capi.hip reads the word without synchronising).  The four workgroups must be co-resident: the host launches at most
// n_cu / 4 clusters at a time.
//
// Workgroup = 8 waves (wave w on SIMD w % 4):
//   waves 0, 1  COMPUTE: wave pp owns the 32 hidden units [64 cq + 32 pp, +32): the carried dW1 / dW2 (both orientations) /
//               db1 tiles and every MFMA of the gradient chain;
//   waves 2, 3, 6, 7  OWNERS (256 threads = 64 tokens x 4 lanes x 16 features): staging of the next step's K / gZ2 / Q / eta
//               tiles into LDS (double-buffered), the output-LayerNorm backward of the next step, the hand-over (flag, poll,
//               record reads), the fused-LN / L2 backward-of-backward -> dZ2, dV, d(eta), dgamma / dbeta, L2 prefetch touches;
//   waves 4, 5  DERIVERS (round 3; beside the compute waves on SIMDs 0 / 1, which idle during the hand-over): wave pp loads the
//               slice's Z1 / Z1b fragments (16 KiB per step and CU where round 2's owners DMA-staged 80 KiB of register
//               images), re-derives X2, gelu', gelu'', X2b, gelu'(Z1b), gX2, gZ1, M and the second orientations, REVERSES the
//               state update to obtain the per-step W1 / W2 (fp32, re-anchored at every forward checkpoint) and writes exactly
//               the operand fragments the compute waves read into the staging regions R1 .. R4; its per-step arithmetic is
//               ttt_bwd4_aux_body.h (also executed on the CPU wave emulator).  It additionally stores gZ1 (N) and the packed
//               W1 per step for the tail kernel.
// Per step i (j = i - 1), 4 workgroup barriers; every staging region keeps ONE buffer, each write sits between the region's
// last reader and its next reader:
//   compute:  S1 (u^T, d(eta) partial, first half of d(gZ2)^T)  |Ba|  S2 (second half -> published record), drain  |Bb|
//             snapshot of the state operands, OUTPUT PATH OF STEP j (it needs no partner data: it fills the hand-over
//             latency)  |Bc|  S4a (dZ1, state updates of step i), publish state for step j  |Bd|
//   owners:   requests of step j's tiles / step i's rows  |Ba|  tiles -> LDS, output-LN backward of step j  |Bb|  flag, poll,
//             records, owner math -> dZ2_i  |Bc|  prefetch touches of step i - 2  |Bd|
//   derivers: R4 <- D1 | M | X2 of step i (parked in L2 since they were derived), R3.W2T <- W2_i^T, loads of Z1b_j  |Ba|
//             R3 <- gelu'(Z1b_j) | X2b_j, loads of Z1_j  |Bb|  (K_j, gZ2_j, eta_j tiles are visible) reverse_step(j): R1 <- gZ1 |
//             D1 | X2 (N) of step j, R2 <- W2_j, park D1 | M | X2 of step j  |Bc|  |Bd|
// What bounds it (round 3, profiles/r3d - r3g): per CU and step ~330 KiB pass through the memory pipeline in ~36 k cycles - the
// step is a chain of dependent memory phases, not of MFMAs; spilled registers are part of that traffic (a build with 399
// spilled dwords ran 24.4 ms per backward, the 138-dword build 14.3 ms: tests/test_kernel_resources_cpu.py pins the budget).
// Math: SURVEY.md Appendix A backward; oracle/ttt_oracle.py:_mlp_step_bwd is the executable spec.
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"
#include "ttt_mfma_bwd_dev.h"
#include "ttt_bwd4_dev.h"
#define TTT_WV_FN __device__ __forceinline__
#include "ttt_bwd4_aux_body.h"
#include <mutex>

namespace ttt {
namespace mfma {
using namespace ttt::mf;

namespace b4 {
using namespace ttt::mfma::b2;
using namespace ttt::mfma::s4;

constexpr int NTC = 512;                                  // 2 compute waves + 4 owner waves + 2 deriver waves
// Wave roles.  Waves are placed on the CU's four SIMDs round-robin (wave w on SIMD w % 4): the compute waves 0, 1 and the
// deriver waves DW0, DW0 + 1 = 4, 5 share SIMDs 0 / 1 - the compute waves idle there during the hand-over, which is when the
// derivers do most of their (VALU-heavy) work -, the owner waves 2, 3, 6, 7 have SIMDs 2 / 3 to themselves: their hand-over
// chain (poll, record reads, LayerNorm backward-of-backward) is the critical path of a step.
// (round 4: DW0 is a template parameter of the kernel - 4 as above, or 2: the derivers on SIMDs 2 / 3 beside two of the owner
// waves, the other two owner waves beside the compute waves - the stage stamps of profiles/r4b show the Bb .. Bc phase bounded by
// the derivers' reverse_step (12.5 k cycles) with the owners finished after 8.2 k and the compute waves after 5.1 k)
constexpr int TILE_B = TILE_ELEMS * 2;                    // 9216 bytes: one padded [64][64] bf16 tile
constexpr int L_K = 0;                                    // K   [2][t][f]  (by step parity)
constexpr int L_G = L_K + 2 * TILE_B;                     // gZ2 [2][t][f]
constexpr int L_Q = L_G + 2 * TILE_B;                     // Q_j   [t][f]
constexpr int L_A = L_Q + TILE_B;                         // dZ2b_j [t][f]
constexpr int L_B = L_A + TILE_B;                         // dZ2_i  [t][f]
constexpr int L_XU = L_B + TILE_B;                        // u^T exchange between the two compute waves: 2 x 4 fragments x 1 KiB
constexpr int L_XD = L_XU + 2 * 4 * 1024;                 // dW2 block exchange: 2 x 2 fragments
constexpr int L_SM = L_XD + 2 * 2 * 1024;                 // floats: eta[2][64], db1[64], db2[64], gamma[64], sync word, db2o[2][64]
constexpr int SM_FLOATS = 2 * 64 + 64 + 64 + 64 + 4 + 2 * 64;
static_assert(2 * 2 * 1024 >= 16 * 64 * 4, "the dW2 exchange region doubles as the 16 x 64 column-sum partials of dZ2b");
// Fragment staging regions, written by the deriver waves (lane-linear fragment images: fragment f of an array at f * 1 KiB +
// lane * 16, the layout of round 2's slot arrays) and read by the compute waves with ds_read_b128:
//   R1  S1 operands of the step        GZ1T | D1N | XT           written for step j after Bb(i)
//   R2  S2 operand                     W2                        written for step j after Bb(i)
//   R3  output-path operands           D1B(j) | X2B(j) | W2T(i)  D1B / X2B after Ba(i), W2T after Bd(i+1) (W2T also feeds S4a)
//   R4  S4a operands                   D1 | GX2 | X2             written for step i after Bd(i+1)
constexpr int FRK = 8 * 1024;                             // one fragment array of a wave pair
constexpr int L_R1 = (L_SM + SM_FLOATS * 4 + 1023) / 1024 * 1024;
constexpr int L_R2 = L_R1 + 3 * FRK;
constexpr int L_R3 = L_R2 + FRK;
constexpr int L_R4 = L_R3 + 3 * FRK;
constexpr int LDS_CL = L_R4 + 3 * FRK;
static_assert(LDS_CL <= 160 * 1024, "LDS budget");
static_assert(2 * TILE_B >= 256 * 16 * 4, "the final dgamma / dbeta reduction re-uses the K / gZ2 tiles");

__device__ unsigned g_fast_count4 = 0;       // DEBUG statistic: cluster workgroups that proved same-XCD placement and switched to plain records

template <int CTRL>
__device__ __forceinline__ float dppq(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum4(float v) {          // over the 4 adjacent lanes that share an owner token
    v += dppq<0xB1>(v);
    v += dppq<0x4E>(v);
    return v;
}
__device__ __forceinline__ void ld16f(__amdgpu_buffer_rsrc_t r, int voff, int soff, float (&o)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = bld4f(r, voff + 16 * q, soff);
        o[4 * q] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
    }
}

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void glb_void;
// fragment `idx` of a staged array (region byte offset `off`) for lane l
__device__ __forceinline__ bf16x8 lfr(const char* smem, int off, int idx, int l) {
    return *reinterpret_cast<const bf16x8*>(smem + off + idx * 1024 + l * 16);
}
// owner / deriver barrier: LDS operations only, global loads stay in flight.  (Round 4 checked the ISA: on ROCm 7.2 a plain
// __syncthreads() compiles to the same pair - no vmcnt(0) - and swapping the two forms changes neither bits nor time, profiles/r4n_*.)
__device__ __forceinline__ void owner_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define TTT_STAMP4(k)                                                        \
    if (DBG && p.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {    \
        const unsigned long long _t = __builtin_readcyclecounter();          \
        p.dbg[16 + (k)] += _t - t_last;                                      \
        t_last = _t;                                                         \
    }

// fragment stores that only the tail kernel (a later launch) reads: non-temporal, so that they do not displace the hand-over
// records, the parked R4 fragments and the prefetched lines of the next steps from this XCD's L2
__device__ __forceinline__ void bst8_nt(__amdgpu_buffer_rsrc_t r, int voff, int soff, bf16x8 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, voff, soff, 2);      // aux bit 1 = nt
}
// the wave backend of ttt_bwd4_aux_body.h on the device
template <bool DBG>
struct DeriverBackend {
    char* base;
    int l;                                  // lane id, re-made opaque every step: everything derived from it (fragment / tile addresses)
                                            // is then recomputed inside the step instead of being hoisted out of the loop and spilled
    __device__ __forceinline__ int lane() const { return l; }
    __device__ __forceinline__ void refresh() { int v = threadIdx.x & 63; asm volatile("" : "+v"(v)); l = v; }
    __device__ __forceinline__ float exp2(float x) const { return __builtin_amdgcn_exp2f(x); }
    __device__ __forceinline__ float rcp(float x) const { return __builtin_amdgcn_rcpf(x); }
    template <class T> __device__ __forceinline__ T lds_load(int byte_off) const { return *reinterpret_cast<const T*>(base + byte_off); }
    template <class T> __device__ __forceinline__ void lds_store(int byte_off, T v) const { *reinterpret_cast<T*>(base + byte_off) = v; }
    __device__ __forceinline__ f32x16 mma3216(bf16x8 a, bf16x8 b, f32x16 c) const { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    __device__ __forceinline__ bf16x4 tr_read(int byte_addr) const {
        return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(base + byte_addr));
    }
    __device__ __forceinline__ bf16x8 opaque8(bf16x8 v) const { asm volatile("" : "+v"(v)); return v; }
    // global store of data that only a LATER kernel reads (the tail): non-temporal
    __device__ __forceinline__ void store_stream(char* p, bf16x8 v) const { __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(p)); }
    // DEBUG: cycle stamps INSIDE reverse_step (entries 36 + k of the timing buffer; deriver wave 0 of workgroup 0 only)
    unsigned long long* dbg = nullptr;
    unsigned long long t_in = 0;
    __device__ __forceinline__ void stamp(int k) {
        if constexpr (DBG) {
            if (dbg != nullptr) {
                const unsigned long long t = __builtin_readcyclecounter();
                dbg[36 + k] += t - t_in;
                t_in = t;
            }
        }
    }
};

#define TTT_PIN_RECORDS(dep)                                                                                                              \
    asm volatile("; records consumed from here"                                                                                           \
                 : "+v"(pa[0][0]), "+v"(pa[0][1]), "+v"(pa[0][2]), "+v"(pa[0][3]), "+v"(pa[1][0]), "+v"(pa[1][1]), "+v"(pa[1][2]), "+v"(pa[1][3]), \
                   "+v"(pa[2][0]), "+v"(pa[2][1]), "+v"(pa[2][2]), "+v"(pa[2][3]), "+v"(pa[3][0]), "+v"(pa[3][1]), "+v"(pa[3][2]), "+v"(pa[3][3]) \
                 : "v"(dep))

// OVL (round 4; the shipped instantiation has it on, the template parameter remains): the owners' partner-independent arithmetic runs
// while the partner records are in flight; off = the round-3 order (wait for the records, then all the arithmetic).
#define TTT_PIN_RECORDS16(dep)                                                                                                            \
    asm volatile("; records consumed from here"                                                                                           \
                 : "+v"(pb[0][0]), "+v"(pb[0][1]), "+v"(pb[1][0]), "+v"(pb[1][1]), "+v"(pb[2][0]), "+v"(pb[2][1]), "+v"(pb[3][0]), "+v"(pb[3][1]) \
                 : "v"(dep))
// R16 (round 4; on in the shipped instantiation): the partial d(gZ2) tiles of the hand-over records travel as bf16 - half the
// bytes a workgroup publishes (and drains in front of barrier Bb) and half the loads of the owners' chain; every workgroup
// still sums the same four rounded partials in the same order, so dZ2 stays bit-identical on the four CUs.  Precision budget:
// tools/diag/lr_gate_full_emul_cpu.py point "P_rec" - no gradient of the DiT fixtures moves beyond its run-to-run spread of
// the other roundings (worst 3.0e-2 -> 3.6e-2 / 2.6e-2 -> 2.5e-2).  [t][PS16] bf16 inside the fp32 tile's area of the record.
constexpr int PS16 = 72;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
// OWN16 (round 4; on in the shipped instantiation): the inner LayerNorm's owner rows of the step record (x_hat, y - target) are bf16
// Round 6: the dK / dQ tail is the group-sequential `mlp_bwd_tail5_kernel`, which rebuilds the per-step W1 and dW1' from one anchor per
// checkpoint group - the derivers carry and store no W1, the compute waves store dW1 (fp32) at the top step of every group instead of a
// packed image every step, gZ1 goes out in the T orientation.
template <bool DBG, bool OVL, bool R16, int DW0, bool OWN16>
__global__ __launch_bounds__(NTC) void mlp_bwd_cluster4_kernel(SweepParams4 p) {
    constexpr int OW0 = DW0 == 2 ? 4 : 2;                        // first owner wave (it polls the partner flags)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Kt2 = reinterpret_cast<__bf16*>(smem + L_K);
    __bf16* Gt2 = reinterpret_cast<__bf16*>(smem + L_G);
    __bf16* Qt = reinterpret_cast<__bf16*>(smem + L_Q);
    __bf16* At = reinterpret_cast<__bf16*>(smem + L_A);
    __bf16* Bt = reinterpret_cast<__bf16*>(smem + L_B);
    char* exu = smem + L_XU;
    char* exd = smem + L_XD;
    float* etaL2 = reinterpret_cast<float*>(smem + L_SM);
    float* db1L = etaL2 + 128;
    float* db2L = db1L + 64;
    float* gamL = db2L + 64;
    unsigned* syncw = reinterpret_cast<unsigned*>(gamL + 64);      // [0] hand-over number the owners have seen complete, [1] fast-path verdict, [2] poisoned
    // Round 4: the OUTPUT path's share of db2 (column sums of dZ2b over the 64 tokens of a step) is summed by the owners from
    // their fp32 dZ2b values, not by a ones-MFMA over the bf16 tile the compute waves consume: db2 enters d(eta) of every token
    // of every earlier step, so the rounding of that tile showed up 1 : 1 in the learning-rate-gate gradients (0.32 on the
    // 3-scene DiT fixture against 0.02 - 0.03 with this sum in fp32; tools/diag/lr_gate_full_emul_cpu.py has the ablation).
    // A step's 16 partials per feature (4 owner waves x 4 DPP rows of 4 tokens) go through the dW2 exchange region, which is
    // idle between Ba and the publish_state behind Bc; owner wave 2 adds them while it waits for its partners.
    float* db2oL = reinterpret_cast<float*>(syncw + 4);            // [2][64] by iteration parity: the owners' share of db2 entering the step

    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cq = __builtin_amdgcn_readfirstlane((int)(blockIdx.x / p.nbh));     // workgroup of the cluster = wave pair w
    const int bh = p.bh0 + (int)(blockIdx.x % p.nbh), head = bh % p.NH;
    const int NC = p.NC;
    char* slots = p.slots + (size_t)bh * p.slot_stride_bh;
    float* carry = p.carry + (size_t)bh * CARRY_FLOATS2;
    const __amdgpu_buffer_rsrc_t rS = make_srd(slots, p.slot_stride_bh);
    const size_t act_bytes = (size_t)NC * 4096 * 2;
    const __amdgpu_buffer_rsrc_t rK = make_srd(p.XK + (size_t)bh * NC * 4096, act_bytes);
    const __amdgpu_buffer_rsrc_t rQ = make_srd(p.XQ + (size_t)bh * NC * 4096, act_bytes);
    const __amdgpu_buffer_rsrc_t rO = make_srd(p.dOut + (size_t)bh * NC * 4096, act_bytes);
    const __amdgpu_buffer_rsrc_t rV = make_srd(p.dXV + (size_t)bh * NC * 4096, act_bytes);
    const __amdgpu_buffer_rsrc_t rE = make_srd(p.eta + (size_t)bh * NC * 64, (size_t)NC * 64 * 2);
    const __amdgpu_buffer_rsrc_t rX = make_srd(p.xch + (size_t)bh * XCH_BH_BYTES, XCH_BH_BYTES);
    const int WREG = cq * (int)SLICE_BYTES;
    auto slot_off = [&](int step) { return (step - p.chunk_lo) * (int)SLOT4_BYTES; };
    const int i0 = p.chunk_hi - 1;
    unsigned* const my_flag = p.flags + ((size_t)bh * 4 + cq) * FLAG_STRIDE;
    if (p.fault && cq == 3) return;             // DEBUG fault injection (tests): this workgroup's partners must time out loudly

    // (Round 6 tried issue priorities by role - s_setprio 3 for the owners' hand-over chain, 2 for the compute waves, 0 for the
    // VALU-heavy derivers: 10.37 against 10.26 ms per backward at NC = 804, worse at every position of Bc; profiles/r6e_*.  Gone.)
    if (wv < 2) {
        // =========================================================================================================== COMPUTE
        const int pp = wv;
        const int nO = 64 * cq + 32 * pp;                      // first hidden unit of this wave
        const int fO = 32 * pp, fX = 32 * (1 - pp);
        f32x16 dW1t[2];      // [a]  dW1[f in 32a.., n in Hp]                                   (rows = f, lane = n)
        f32x16 dW2t[2];      // [0] dW2[n in Hp, f in Fp], [1] dW2[n in Hp, f in Fx]           (rows = n, lane = f)
        f32x16 dW2Tt[2];     // same blocks transposed                                          (rows = f, lane = n)
        float db1v, db2v;    // db1[nO + c] ; db2[fO + c] (every workgroup carries db2: its owners need it)
        {
            const int l = tid & 63, h = l >> 5, c = l & 31;
            const float* g1 = p.first ? p.uW1 + (size_t)bh * 64 * 256 : carry + C_DW1;
            const float* g2 = p.first ? p.uW2 + (size_t)bh * 256 * 64 : carry + C_DW2;
            const float* gb1 = p.first ? p.ub1 + (size_t)bh * 256 : carry + C_DB1;
            const float* gb2 = p.first ? p.ub2 + (size_t)bh * 64 : carry + C_DB2;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = row_of(r, h);
                dW1t[0][r] = g1[(size_t)ro * 256 + nO + c];
                dW1t[1][r] = g1[(size_t)(32 + ro) * 256 + nO + c];
                dW2t[0][r] = g2[(size_t)(nO + ro) * 64 + fO + c];
                dW2t[1][r] = g2[(size_t)(nO + ro) * 64 + fX + c];
                dW2Tt[0][r] = g2[(size_t)(nO + c) * 64 + fO + ro];
                dW2Tt[1][r] = g2[(size_t)(nO + c) * 64 + fX + ro];
            }
            db1v = gb1[nO + c];
            db2v = gb2[fO + c];
        }
        bf16x8 ONES;
#pragma unroll
        for (int e = 0; e < 8; ++e) ONES[e] = (__bf16)1.0f;

        // output path of step j: W2' = state entering step j + 1 (slot j + 1), X2b / gelu'(Z1b) from slot j, dZ2b_j in At, Q_j in Qt
        auto add_output_path = [&](int j) {
            const int l = tid & 63;
            const int sj = slot_off(j) + WREG, l16 = l * 16;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 dz = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    dz = mma(row_pi(At, ti, fO, s, l), lfr(smem, L_R3 + 2 * FRK, fr_idx(pp, pp, s), l), dz);        // W2T of slot j + 1
                    dz = mma(row_pi(At, ti, fX, s, l), lfr(smem, L_R3 + 2 * FRK, fr_idx(1 - pp, pp, s), l), dz);
                }
                const f32x16 d1b = unpack2(lfr(smem, L_R3, fr_idx(ti, pp, 0), l), lfr(smem, L_R3, fr_idx(ti, pp, 1), l));
#pragma unroll
                for (int r = 0; r < 16; ++r) dz[r] *= d1b[r];
                db1v += tile_colsum(dz);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 zf = pack(dz, s);                       // dZ1b (k = t rows, j = n lane)
                    bst8_nt(rS, l16, sj + fro4(A_DZ1B, fr_idx(ti, pp, s)), zf);
                    dW1t[0] = mma(tr_pi(Qt, 32 * ti, s, 0, l), zf, dW1t[0]);
                    dW1t[1] = mma(tr_pi(Qt, 32 * ti, s, 32, l), zf, dW1t[1]);
                    const bf16x8 xb = lfr(smem, L_R3 + FRK, fr_idx(ti, pp, s), l);          // X2b (m = n lane, k = t rows)
                    const bf16x8 aO = tr_pi(At, 32 * ti, s, fO, l), aX = tr_pi(At, 32 * ti, s, fX, l);
                    dW2t[0] = mma(xb, aO, dW2t[0]);
                    dW2t[1] = mma(xb, aX, dW2t[1]);
                    dW2Tt[0] = mma(aO, xb, dW2Tt[0]);
                    dW2Tt[1] = mma(aX, xb, dW2Tt[1]);
                    // (db2 += column sums of dZ2b: the owners' share, in fp32 - db2oL)
                }
            }
        };
        // what the next S1 / the owners / the tail kernel need: complete dW1 (slot j), db1, db2, the dW2 block the partner contracts over
        auto publish_state = [&](int j) {
            const int l = tid & 63, h = l >> 5, c = l & 31;
            const int sj = slot_off(j) + WREG;
            if ((j + 1) % p.G == 0 || j == NC - 1) {           // top step of a checkpoint group: the tail's anchor, natural [f][256] fp32
                float* da = p.danchor + ((size_t)bh * p.K + j / p.G) * (64 * 256);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = row_of(r, h);
                    __builtin_nontemporal_store(dW1t[0][r], da + (size_t)ro * 256 + nO + c);
                    __builtin_nontemporal_store(dW1t[1][r], da + (size_t)(32 + ro) * 256 + nO + c);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) *reinterpret_cast<bf16x8*>(exd + ((size_t)(pp * 2 + s) * 64 + l) * 16) = pack(dW2t[1], s);
            if (h == 0) { db1L[32 * pp + c] = db1v; db2L[fO + c] = db2v; }
        };

        __syncthreads();                       // P0 (owners: gamma row, sync word)
        __syncthreads();                       // P1: tiles of step i0, dZ2b_i0 parked by the owners
        add_output_path(i0);
        publish_state(i0);
        __syncthreads();                       // P2

        unsigned long long t_last = __builtin_readcyclecounter();
        for (int i = i0; i >= p.chunk_lo; --i) {
            int l_op = tid & 63;
            asm volatile("" : "+v"(l_op));           // opaque lane id: keeps address arithmetic inside the loop (no hoist + spill)
            const int l = l_op, h = l >> 5, c = l & 31;
            const bool more = i > p.chunk_lo;
            const int cur = (i0 - i) & 1;
            const __bf16* Kt = Kt2 + cur * TILE_ELEMS;
            const __bf16* Gt = Gt2 + cur * TILE_ELEMS;
            const float* etaL = etaL2 + cur * 64;
            const int sI = slot_off(i), sw = sI + WREG, l16 = l * 16;
            const unsigned epoch = (unsigned)(i0 - i) + 1;                 // hand-over number of this step, 1-based
            const int xmine = ((int)(epoch & 1) * 4 + cq) * XCH_REC_BYTES;   // this workgroup's record of this step
            // Records are published write-through (sc1: placement-independent).  When the first hand-over has PROVEN that the four
            // workgroups sit on one XCD (they exchanged their HW_REG_XCC_ID), later records are stored plain: the lines stay in
            // that XCD's L2, where the partners' sc1 loads (L1-bypassing) find them - the same protocol with less latency.
            const bool fast = epoch > 1 && __hip_atomic_load(syncw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u;

            // ================= S1 : (rows = n, lane = t) products, u^T, d(eta) partial, first half of d(gZ2)^T ==============
            f32x16 P[2];                       // [ti]  d(gZ2)^T partial (rows = f in Fp, lane = t)
            bf16x8 uN[2][2];                   // [ti][s]  u^T (k = n rows, j = t lane)
            float se2[2];
            {
                const f32x16 db1R = rows_from_lds(db1L + 32 * pp, 0, h);
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    f32x16 e1 = db1R;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        e1 = mma(pack(dW1t[0], s), row_pi(Kt, ti, 0, s, l), e1);
                        e1 = mma(pack(dW1t[1], s), row_pi(Kt, ti, 32, s, l), e1);
                    }
                    const float ec = -etaL[32 * ti + c];
                    float se = 0.f;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 g1 = lfr(smem, L_R1, fr_idx(pp, ti, s), l);              // gZ1^T
                        const bf16x8 d1 = lfr(smem, L_R1 + FRK, fr_idx(pp, ti, s), l);        // gelu'(Z1)^T
                        bf16x8 uf;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float ev = e1[8 * s + e];
                            se += (float)g1[e] * ev;
                            uf[e] = (__bf16)(ec * ev * (float)d1[e]);
                        }
                        uN[ti][s] = uf;
                        *reinterpret_cast<bf16x8*>(exu + ((size_t)(pp * 4 + ti * 2 + s) * 64 + l) * 16) = uf;
                    }
                    se2[ti] = se;
                }
            }
            {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    f32x16 a2 = zero16();
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        a2 = mma(pack(dW2Tt[0], s), row_pi(Gt, ti, fO, s, l), a2);
                        a2 = mma(pack(dW2Tt[1], s), row_pi(Gt, ti, fX, s, l), a2);
                    }
                    float se = se2[ti];
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 x2 = lfr(smem, L_R1 + 2 * FRK, fr_idx(pp, ti, s), l);    // X2^T
#pragma unroll
                        for (int e = 0; e < 8; ++e) se += (float)x2[e] * a2[8 * s + e];
                    }
                    se = xor_add(se, 32);
                    if (h == 0) {
                        if (fast) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, -se), rX, XCH_PART_BYTES + (pp * 64 + 32 * ti + c) * 4, xmine, 0);
                        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, -se), rX, XCH_PART_BYTES + (pp * 64 + 32 * ti + c) * 4, xmine, 16);
                    }
                }
            }
            {
                const bf16x8 D2o0 = pack(dW2t[0], 0), D2o1 = pack(dW2t[0], 1);
                const bf16x8 D2x0 = *reinterpret_cast<const bf16x8*>(exd + ((size_t)((pp ^ 1) * 2 + 0) * 64 + l) * 16);
                const bf16x8 D2x1 = *reinterpret_cast<const bf16x8*>(exd + ((size_t)((pp ^ 1) * 2 + 1) * 64 + l) * 16);
#pragma unroll
                for (int ti = 0; ti < 2; ++ti) {
                    f32x16 pa = zero16();
                    pa = mma(D2o0, lfr(smem, L_R1 + 2 * FRK, fr_idx(pp, ti, 0), l), pa);
                    pa = mma(D2o1, lfr(smem, L_R1 + 2 * FRK, fr_idx(pp, ti, 1), l), pa);
                    pa = mma(D2x0, lfr(smem, L_R1 + 2 * FRK, fr_idx(1 - pp, ti, 0), l), pa);
                    pa = mma(D2x1, lfr(smem, L_R1 + 2 * FRK, fr_idx(1 - pp, ti, 1), l), pa);
                    const float ec = -etaL[32 * ti + c];
#pragma unroll
                    for (int r = 0; r < 16; ++r) pa[r] *= ec;
                    P[ti] = pa;
                }
            }
            TTT_STAMP4(0)
            __syncthreads();                   // Ba: u^T fragments visible
            TTT_STAMP4(1)

            // ================= S2 : second half of d(gZ2)^T partial -> LDS (own owners) + published record (partners) ======
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 ux = *reinterpret_cast<const bf16x8*>(exu + ((size_t)((pp ^ 1) * 4 + ti * 2 + s) * 64 + l) * 16);
                    P[ti] = mma(lfr(smem, L_R2, fr_idx(pp, pp, s), l), uN[ti][s], P[ti]);
                    P[ti] = mma(lfr(smem, L_R2, fr_idx(1 - pp, pp, s), l), ux, P[ti]);
                }
                if constexpr (R16) {
                    const int vo = ((32 * ti + c) * PS16 + 32 * pp + 4 * h) * 2;   // [t][f] bf16 image (row stride PS16)
#pragma unroll
                    for (int s2 = 0; s2 < 2; ++s2) {
                        const u32x4 v = __builtin_bit_cast(u32x4, pack(P[ti], s2));      // rows 8 s2 .. + 8: f = 16 s2 + 4 h + j | + 8
                        const u32x2 lo = {v[0], v[1]}, hi = {v[2], v[3]};
                        if (fast) {
                            __builtin_amdgcn_raw_buffer_store_b64(lo, rX, vo + 32 * s2, xmine, 0);
                            __builtin_amdgcn_raw_buffer_store_b64(hi, rX, vo + 32 * s2 + 16, xmine, 0);
                        } else {
                            __builtin_amdgcn_raw_buffer_store_b64(lo, rX, vo + 32 * s2, xmine, 16);
                            __builtin_amdgcn_raw_buffer_store_b64(hi, rX, vo + 32 * s2 + 16, xmine, 16);
                        }
                    }
                } else {
                    const int vo = ((32 * ti + c) * PS + 32 * pp + 4 * h) * 4;   // [t][f] image (row stride PS), write-through
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const f32x4 v = {P[ti][4 * q4], P[ti][4 * q4 + 1], P[ti][4 * q4 + 2], P[ti][4 * q4 + 3]};
                        if (fast) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rX, vo + 32 * q4, xmine, 0);
                        else bst4f_sc1(rX, vo + 32 * q4, xmine, v);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0) ; drain: the record is in memory before the flag is stored" ::: "memory");
            TTT_STAMP4(2)
            __syncthreads();                   // Bb: this workgroup's record is complete and drained
            TTT_STAMP4(3)

            // ================= S4a operands of the state ENTERING step i, then the output path of step j =====================
            // (the output path adds to the carried state; S4a below must contract over the state before those additions)
            const float db1_old = db1v;
            bf16x8 D1p[2][2], D2p[2][2];       // [a][s] pack(dW1t[a], s) ; pack(dW2Tt[a], s)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) { D1p[a][s] = pack(dW1t[a], s); D2p[a][s] = pack(dW2Tt[a], s); }
            if (more) add_output_path(i - 1);
            TTT_STAMP4(4)
            __syncthreads();                   // Bc: dZ2_i (Bt) written by the owners
            TTT_STAMP4(5)

            // ================= S4a : first-layer gradients and this step's state updates ====================================
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const f32x16 etaR = rows_from_lds(etaL, 32 * ti, h);
                f32x16 e1 = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    e1 = mma(row_pi(Kt, ti, 0, s, l), D1p[0][s], e1);
                    e1 = mma(row_pi(Kt, ti, 32, s, l), D1p[1][s], e1);
                }
                bf16x8 d1f[2], uf[2];                                       // gelu'(Z1) fragments ; u (m = n lane, k = t rows)
                f32x16 dz;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    d1f[s] = lfr(smem, L_R4, fr_idx(ti, pp, s), l);
                    const bf16x8 mm = lfr(smem, L_R4 + FRK, fr_idx(ti, pp, s), l);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float dg = -etaR[8 * s + e] * (e1[8 * s + e] + db1_old);      // d(gZ1)
                        uf[s][e] = (__bf16)(dg * (float)d1f[s][e]);
                        dz[8 * s + e] = dg * (float)mm[e];
                    }
                }
                {
                    f32x16 dx = zero16();
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        dx = mma(row_pi(Gt, ti, fO, s, l), D2p[0][s], dx);
                        dx = mma(row_pi(Gt, ti, fX, s, l), D2p[1][s], dx);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dx[r] *= -etaR[r];          // -eta A2
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        dx = mma(row_pi(Bt, ti, fO, s, l), lfr(smem, L_R3 + 2 * FRK, fr_idx(pp, pp, s), l), dx);
                        dx = mma(row_pi(Bt, ti, fX, s, l), lfr(smem, L_R3 + 2 * FRK, fr_idx(1 - pp, pp, s), l), dx);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dz[r] += dx[r] * (float)d1f[r >> 3][r & 7];     // dZ1
                }
                db1v += tile_colsum(dz);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 zf = pack(dz, s);                        // dZ1 (k = t rows, j = n lane)
                    bst8_nt(rS, l16, sw + fro4(A_DZ1, fr_idx(ti, pp, s)), zf);
                    dW1t[0] = mma(tr_pi(Kt, 32 * ti, s, 0, l), zf, dW1t[0]);
                    dW1t[1] = mma(tr_pi(Kt, 32 * ti, s, 32, l), zf, dW1t[1]);
                    const bf16x8 gO = tr_pi(Gt, 32 * ti, s, fO, l), gX = tr_pi(Gt, 32 * ti, s, fX, l);
                    dW2t[0] = mma(uf[s], gO, dW2t[0]);
                    dW2t[1] = mma(uf[s], gX, dW2t[1]);
                    dW2Tt[0] = mma(gO, uf[s], dW2Tt[0]);
                    dW2Tt[1] = mma(gX, uf[s], dW2Tt[1]);
                    const bf16x8 xf = lfr(smem, L_R4 + 2 * FRK, fr_idx(ti, pp, s), l);       // X2 (m = n lane, k = t rows)
                    const bf16x8 zO = tr_pi(Bt, 32 * ti, s, fO, l), zX = tr_pi(Bt, 32 * ti, s, fX, l);
                    dW2t[0] = mma(xf, zO, dW2t[0]);
                    dW2t[1] = mma(xf, zX, dW2t[1]);
                    dW2Tt[0] = mma(zO, xf, dW2Tt[0]);
                    dW2Tt[1] = mma(zX, xf, dW2Tt[1]);
                    const f32x16 acc = mma(ONES, zO, zero16());
                    db2v += acc[0];
                }
            }
            if (more) publish_state(i - 1);
            TTT_STAMP4(6)
            __syncthreads();                   // Bd: every read of K_i, gZ2_i, dZ2_i, dZ2b_j, Q_j is done; state published
            TTT_STAMP4(7)
        }

        // ---- hand the state gradient to the next chunk, or emit the final results (this workgroup's slices) -------------------
        {
            const int l = tid & 63, h = l >> 5, c = l & 31;
            float* o1 = p.last ? p.dW1 + (size_t)bh * 64 * 256 : carry + C_DW1;
            float* o2 = p.last ? p.dW2 + (size_t)bh * 256 * 64 : carry + C_DW2;
            float* ob1 = p.last ? p.db1 + (size_t)bh * 256 : carry + C_DB1;
            float* ob2 = p.last ? p.db2 + (size_t)bh * 64 : carry + C_DB2;
            // (the last Bd ordered the owners' poison word before this read)
            const float poison = syncw[2] != 0u ? __builtin_nanf("") : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = row_of(r, h);
                o1[(size_t)ro * 256 + nO + c] = dW1t[0][r] + poison;
                o1[(size_t)(32 + ro) * 256 + nO + c] = dW1t[1][r] + poison;
                o2[(size_t)(nO + ro) * 64 + fO + c] = dW2t[0][r] + poison;
                o2[(size_t)(nO + ro) * 64 + fX + c] = dW2t[1][r] + poison;
            }
            if (h == 0) ob1[nO + c] = db1v + poison;
            // (the owners' share after the last step: its last writer is ordered by the last Bc / Bd)
            if (cq == 0 && h == 0) ob2[fO + c] = db2v + db2oL[((i0 - p.chunk_lo) & 1) * 64 + fO + c] + poison;
        }
        if (p.last) __syncthreads();           // (the owners' final reduction)
    } else if (wv != DW0 && wv != DW0 + 1) {
        // =========================================================================================================== OWNERS
        int ow = ((DW0 == 2 ? wv - 4 : (wv < DW0 ? wv - 2 : wv - 4)) << 6) | (tid & 63);      // 0 .. 255 over the four owner waves
        int ot = ow >> 2, of0 = 16 * (ow & 3);                  // token, first of this thread's 16 features
        float dgam[16], dbet[16];
        if (p.first) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { dgam[k] = 0.f; dbet[k] = 0.f; }
        } else {
            load16_f32(carry + C_DG + (size_t)ow * 16, dgam);
            load16_f32(carry + C_DBT + (size_t)ow * 16, dbet);
        }
        if (ow < 64) gamL[ow] = p.ln_w[(size_t)head * 64 + ow];
        float gam[16];
        load16_f32(p.ln_w + (size_t)head * 64 + of0, gam);
        const unsigned my_xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]: the XCD this workgroup runs on
        if (ow == 0) {
            syncw[0] = 0u; syncw[1] = 0u; syncw[2] = 0u;
            __hip_atomic_store(my_flag + 1, my_xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // drained before P1, i.e. before flag 1
        }

        // Everything the owners need of a step ahead of its turn - the K / gZ2 / Q / eta tiles (256 threads x 32 bytes per tile)
        // and the inputs of the output-LayerNorm backward - is REQUESTED at the top of an iteration and CONSUMED after barrier
        // Ba: the requests fly while the compute waves run S1, and nothing queues in front of the hand-over loads later on.
        struct StepLoads {
            uint4 k0, k1, g0, g1, q0, q1;
            bf16x8 da, db;
            float xl[16];
            float rstdl;
            unsigned short ev;
        };
        auto request_step = [&](int step, StepLoads& L) {
            const int vo = ow * 32;
            L.k0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rK, vo, step * 8192, 0));
            L.k1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rK, vo + 16, step * 8192, 0));
            const int sg = slot_off(step) + (int)(SLOT4_FR + SLOT4_OWN);
            L.g0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rS, vo, sg, 0));
            L.g1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rS, vo + 16, sg, 0));
            L.q0 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rQ, vo, step * 8192, 0));
            L.q1 = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rQ, vo + 16, step * 8192, 0));
            L.ev = 0;
            if (ow < 64) L.ev = __builtin_amdgcn_raw_buffer_load_b16(rE, ow * 2, step * 128, 0);
            const int so = slot_off(step) + (int)SLOT4_FR;
            L.da = bld8(rO, vo, step * 8192);
            L.db = bld8(rO, vo + 16, step * 8192);
            ld16f(rS, ow * 64, so + 2 * (int)SLOT_OWN_ARR, L.xl);
            L.rstdl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, ot * 8 + 4, so + 3 * (int)SLOT_OWN_ARR, 0));
        };
        // tiles -> LDS (K, gZ2, eta into buffer `dst`, Q into Qt); backward of the output LayerNorm -> dZ2b tile (At), dgamma / dbeta
        auto consume_step = [&](int dst, const StepLoads& L, float* part) {
            __bf16* kd = Kt2 + dst * TILE_ELEMS + ot * TS + of0;
            __bf16* gd = Gt2 + dst * TILE_ELEMS + ot * TS + of0;
            __bf16* qd = Qt + ot * TS + of0;
            *reinterpret_cast<uint4*>(kd) = L.k0; *reinterpret_cast<uint4*>(kd + 8) = L.k1;
            *reinterpret_cast<uint4*>(gd) = L.g0; *reinterpret_cast<uint4*>(gd + 8) = L.g1;
            *reinterpret_cast<uint4*>(qd) = L.q0; *reinterpret_cast<uint4*>(qd + 8) = L.q1;
            if (ow < 64) etaL2[dst * 64 + ow] = __builtin_bit_cast(float, (unsigned)L.ev << 16);
            float d[16], g[16];
#pragma unroll
            for (int k = 0; k < 8; ++k) { d[k] = (float)L.da[k]; d[8 + k] = (float)L.db[k]; }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                dgam[k] += d[k] * L.xl[k];
                dbet[k] += d[k];
                g[k] = d[k] * gam[k];
                s1 += g[k]; s2 += g[k] * L.xl[k];
            }
            s1 = sum4(s1); s2 = sum4(s2);
#pragma unroll
            for (int k = 0; k < 16; ++k) g[k] = (64.0f * g[k] - s1 - L.xl[k] * s2) * L.rstdl * (1.0f / 64.0f);
            store16_bf16(At + ot * TS + of0, g);
            // column sums of the fp32 dZ2b: two DPP row shifts leave the sum of a row's four tokens in its lanes 12 .. 15
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                g[k] += dppq<0x114>(g[k]);                 // row_shr:4 (zero beyond the row)
                g[k] += dppq<0x118>(g[k]);                 // row_shr:8
            }
            if ((tid & 12) == 12) {                        // partial (owner wave, row) x features of0 .. of0 + 16
                float* pw = part + (ow >> 4) * 64 + of0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = {g[4 * q], g[4 * q + 1], g[4 * q + 2], g[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(pw + 4 * q) = v;
                }
            }
        };
        // owner wave 2, lane = feature: the owners' share of db2 entering the NEXT step = this one + the 16 partials (fixed order)
        auto add_parts = [&](const float* part, const float* from, float* to) {
            float s4 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) s4 += part[q * 64 + (tid & 63)];
            to[tid & 63] = (from ? from[tid & 63] : 0.f) + s4;
        };
        __syncthreads();                       // P0: gamma row, sync word visible to all owner waves
        {
            StepLoads L0;
            request_step(i0, L0);
            consume_step(0, L0, reinterpret_cast<float*>(exu));       // (exd is written by publish_state(i0) between P1 and P2)
        }
        owner_barrier();                       // P1
        if (wv == OW0) add_parts(reinterpret_cast<const float*>(exu), nullptr, db2oL);
        owner_barrier();                       // P2

        unsigned long long t_prev = 0;
        for (int i = i0; i >= p.chunk_lo; --i) {
            // opaque owner index per step (as the compute / deriver waves do with their lane id): what derives from it - record
            // and tile offsets, LDS rows - is re-made inside the step instead of being carried through the loop in spilled registers
            asm volatile("" : "+v"(ow));
            ot = ow >> 2; of0 = 16 * (ow & 3);
            const bool more = i > p.chunk_lo;
            const int cur = (i0 - i) & 1;
            const int sI = slot_off(i);
            const unsigned epoch = (unsigned)(i0 - i) + 1;
            const int xrec = (int)(epoch & 1) * 4 * XCH_REC_BYTES;
            // ---- requests of this iteration: the owner inputs of step i, the tiles / output-LayerNorm inputs of step j - all in
            //      flight across Ba ------------------------------------------------------------------------------------------------
            float xh[16], go[16];
            u32x4 xg16[4];                     // own16: the two rows as bf16, 2 x 16 bytes each, converted behind Bb
            const int so = sI + (int)SLOT4_FR;
            if constexpr (OWN16) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    xg16[u] = __builtin_amdgcn_raw_buffer_load_b128(rS, ow * 32 + 16 * u, so, 0);
                    xg16[2 + u] = __builtin_amdgcn_raw_buffer_load_b128(rS, ow * 32 + 16 * u, so + (int)SLOT_OWN_ARR, 0);
                }
            } else {
                ld16f(rS, ow * 64, so, xh);
                ld16f(rS, ow * 64, so + (int)SLOT_OWN_ARR, go);
            }
            const float r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, ot * 8, so + 3 * (int)SLOT_OWN_ARR, 0));
            StepLoads Lj;
            if (more) request_step(i - 1, Lj);
            // DEBUG stamps 32 .. 35 (owner wave 0 of workgroup 0): Bd .. Ba arrival, wait at Ba, consume_step, wait at Bb
            unsigned long long t_o = 0;
#define TTT_OSTAMP2(k)                                                           \
            if (DBG && p.dbg != nullptr && blockIdx.x == 0 && ow == 0) {         \
                const unsigned long long _t = __builtin_readcyclecounter();      \
                if (t_prev) p.dbg[32 + (k)] += _t - t_prev;                      \
                t_prev = _t;                                                     \
            }
            TTT_OSTAMP2(0)
            owner_barrier();                   // Ba (nothing of the owners is due yet: they arrive at once)
            TTT_OSTAMP2(1)
            if (more) consume_step(cur ^ 1, Lj, reinterpret_cast<float*>(exd));
            TTT_OSTAMP2(2)
            owner_barrier();                   // Bb: this workgroup's record is complete and drained; At / Q_j / R3 visible
            TTT_OSTAMP2(3)
            if (DBG && p.dbg != nullptr && blockIdx.x == 0 && ow == 0) t_o = __builtin_readcyclecounter();
#define TTT_OSTAMP(k)                                                            \
            if (DBG && p.dbg != nullptr && blockIdx.x == 0 && ow == 0) {         \
                const unsigned long long _t = __builtin_readcyclecounter();      \
                p.dbg[24 + (k)] += _t - t_o;                                     \
                t_o = _t;                                                        \
            }
            // ---- O-C: hand-over ------------------------------------------------------------------------------------------------
            if (ow == 0) __hip_atomic_store(my_flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (OWN16) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        xh[8 * u + 2 * w] = __builtin_bit_cast(float, xg16[u][w] << 16);
                        xh[8 * u + 2 * w + 1] = __builtin_bit_cast(float, xg16[u][w] & 0xffff0000u);
                        go[8 * u + 2 * w] = __builtin_bit_cast(float, xg16[2 + u][w] << 16);
                        go[8 * u + 2 * w + 1] = __builtin_bit_cast(float, xg16[2 + u][w] & 0xffff0000u);
                    }
            }
            if (wv == OW0) {
                const int l = tid & 63;
                if (more) add_parts(reinterpret_cast<const float*>(exd), db2oL + cur * 64, db2oL + (cur ^ 1) * 64);
                if (l < 3) {
                    const unsigned* f = p.flags + ((size_t)bh * 4 + ((cq + 1 + l) & 3)) * FLAG_STRIDE;
                    // A partner that is not running: give up LOUDLY instead of hanging the GPU - after a WALL-CLOCK time (the
                    // constant 100-MHz counter, looked at every 256 polls), not after a number of polls: how long a poll takes
                    // depends on what else the chip is doing (a profiler, RCCL kernels on the CUs a partner is waiting for).
                    // 2 s is 5 orders above a step; fault injection: 2 ms.  A poisoned workgroup (an earlier poll of this
                    // launch gave up) does not wait at all.
                    const unsigned long long t_poll = wall_clock64();
                    const unsigned long long t_lim = p.fault ? 200000ull : 200000000ull;
                    unsigned spins = 0u;
                    bool give_up = syncw[2] != 0u;
                    while (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 255u) == 0u && wall_clock64() - t_poll > t_lim) give_up = true;
                        if (give_up) {
                            __hip_atomic_store(p.err, 1u + (unsigned)bh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            __hip_atomic_store(syncw + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            break;
                        }
                    }
                }
                // (the wave re-converges here: all three partner records of this step are published)
                if (epoch == 1) {              // where do the partners run?  (their XCC words were stored before their first flag)
                    bool same = true;
                    if (l < 3) {
                        const unsigned* f = p.flags + ((size_t)bh * 4 + ((cq + 1 + l) & 3)) * FLAG_STRIDE;
                        same = __hip_atomic_load(f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == my_xcc + 1u;
                    }
                    const bool all_same = __builtin_amdgcn_ballot_w64(!same) == 0ull;
                    const bool use_fast = all_same && p.fast_records != 0;
                    if (l == 0) {
                        __hip_atomic_store(syncw + 1, use_fast ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        if (use_fast) atomicAdd(&g_fast_count4, 1u);
                    }
                }
                if (l == 0) __hip_atomic_store(syncw, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            } else {
                // (no bound of its own: wave 2 of this workgroup always arrives - its poll above is bounded by the wall clock -, and
                // a bound counted in polls here could expire BEFORE that one and let these waves go on with stale records)
                while (__hip_atomic_load(syncw, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < epoch) __builtin_amdgcn_s_sleep(1);
            }
            TTT_OSTAMP(0)                      // flag + poll
            float G_[16];
            float dep[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // all four records - this workgroup's own included - come back through memory with sc1 loads: no branch on cq,
            // and the same summation order q = 0..3 on all four workgroups -> bit-identical dZ2 everywhere
            f32x4 pa[R16 ? 1 : 4][4];
            u32x4 pb[4][2];                    // R16: 16 bf16 per record
            {
                if constexpr (R16) {
                    const int vo = (ot * PS16 + of0) * 2;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                        for (int u = 0; u < 2; ++u) pb[qq][u] = __builtin_amdgcn_raw_buffer_load_b128(rX, vo + 16 * u, xrec + qq * XCH_REC_BYTES, 16);
                } else {
                    const int vo = (ot * PS + of0) * 4;
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                        for (int u = 0; u < 4; ++u) pa[qq][u] = bld4f_sc1(rX, vo + 16 * u, xrec + qq * XCH_REC_BYTES);
                }
                if (cq == 0 && (ow & 3) == 0) {            // workgroup 0 finishes d(eta): request the eight per-wave partials now
#pragma unroll
                    for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                            dep[2 * qq + u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, XCH_PART_BYTES + (u * 64 + ot) * 4,
                                                                                                        xrec + qq * XCH_REC_BYTES, 16));
                }
            }
            {
                // ---- what needs no partner - gZ2 of this step in fp32 and its row statistics, the db2 term of d(eta) - runs while
                // the records are in flight (round 4: the ISA used to wait for all sixteen loads first and start this arithmetic
                // afterwards; the asm statement below pins the order)
                const float eta_t = etaL2[cur * 64 + ot];
                if constexpr (!OVL && !R16) TTT_PIN_RECORDS(eta_t);
                if constexpr (!OVL && R16) TTT_PIN_RECORDS16(eta_t);
                float gxh[16], gz[16];
                float s1g = 0.f, s2g = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    gxh[k] = go[k] * gam[k];
                    s1g += gxh[k]; s2g += gxh[k] * xh[k];
                }
                s1g = sum4(s1g); s2g = sum4(s2g);
                float se = 0.f, s1 = 0.f, s2 = 0.f;
                float edb[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    gz[k] = (64.0f * gxh[k] - s1g - xh[k] * s2g) * r * (1.0f / 64.0f);      // gZ2 (fp32)
                    const float db2 = db2L[of0 + k] + db2oL[cur * 64 + of0 + k];
                    se += gz[k] * db2;
                    edb[k] = eta_t * db2;
                }
                se = sum4(se);
                // the records are "produced" here as far as the compiler can tell: their sums cannot be scheduled (and waited
                // for) above the arithmetic that `se` depends on
                if constexpr (R16) {
                    if constexpr (OVL) TTT_PIN_RECORDS16(se);
                    // bf16 pair (lo, hi) of dword w = features 2w, 2w + 1: the same four partials in the same order on all four CUs
                    auto lo = [](unsigned w) { return __builtin_bit_cast(float, w << 16); };
                    auto hi = [](unsigned w) { return __builtin_bit_cast(float, w & 0xffff0000u); };
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            G_[8 * u + 2 * w] = ((lo(pb[0][u][w]) + lo(pb[1][u][w])) + lo(pb[2][u][w])) + lo(pb[3][u][w]);
                            G_[8 * u + 2 * w + 1] = ((hi(pb[0][u][w]) + hi(pb[1][u][w])) + hi(pb[2][u][w])) + hi(pb[3][u][w]);
                        }
                } else {
                    if constexpr (OVL) TTT_PIN_RECORDS(se);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const f32x4 v = ((pa[0][u] + pa[1][u]) + pa[2][u]) + pa[3][u];
                        G_[4 * u] = v[0]; G_[4 * u + 1] = v[1]; G_[4 * u + 2] = v[2]; G_[4 * u + 3] = v[3];
                    }
                }
                TTT_OSTAMP(1)                  // the four records have arrived
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    G_[k] -= edb[k];                                                         // d(gZ2) complete
                    const float m = -G_[k] * r;
                    s1 += m; s2 += m * xh[k];
                }
                s1 = sum4(s1); s2 = sum4(s2);
                float a1 = 0.f, a2 = 0.f, dxh[16];
                bf16x8 dv0, dv1;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const float g = gam[k];
                    const float m = -G_[k] * r;
                    const float dgxh = r * G_[k] + (s1 + xh[k] * s2) * (1.0f / 64.0f);
                    const float dy = g * dgxh;
                    dgam[k] += go[k] * dgxh + dy * xh[k];
                    dbet[k] += dy;
                    if (k < 8) dv0[k] = (__bf16)(-dy); else dv1[k - 8] = (__bf16)(-dy);     // dt = -dy ; dV = dt
                    dxh[k] = dy * g + (gxh[k] * s2 + s2g * m) * (1.0f / 64.0f);
                    const float dstd = -dxh[k] * xh[k] * r - G_[k] * gz[k] * r;
                    a1 += dxh[k]; a2 += dstd;
                }
                a1 = sum4(a1); a2 = sum4(a2);
#pragma unroll
                for (int k = 0; k < 16; ++k) G_[k] = dxh[k] * r - a1 * r * (1.0f / 64.0f) + a2 * xh[k] * (1.0f / 64.0f);   // dZ2
                store16_bf16(Bt + ot * TS + of0, G_);
                if (cq == 0) {
                    // (ordered after the polling wave's poison store by the acquire of syncw[0] above)
                    if (__hip_atomic_load(syncw + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) { dv0[k] = (__bf16)__builtin_nanf(""); dv1[k] = (__bf16)__builtin_nanf(""); }
                        se = __builtin_nanf("");
                    }
                    bst8(rV, ow * 32, i * 8192, dv0);
                    bst8(rV, ow * 32 + 16, i * 8192, dv1);
                    if ((ow & 3) == 0) {      // workgroup 0 finishes d(eta): the eight per-wave partials of the four records
                        float de = -se;
#pragma unroll
                        for (int u = 0; u < 8; ++u) de += dep[u];
                        p.deta[((size_t)bh * NC + i) * 64 + ot] = (__bf16)de;
                    }
                }
            }
            TTT_OSTAMP(2)                      // owner math, dZ2 / dV / d(eta) stores
            owner_barrier();                   // Bc: dZ2_i visible to the compute waves
            TTT_OSTAMP(3)                      // wait for the compute waves at Bc
            // ---- L2 prefetch, two steps ahead: one dword per 128-byte line of what the owners of this CU will request for step
            // i - 2 (owner rows, gZ2 tile: 452 lines; K, Q, dOut tiles: 192 lines).  Issued HERE, behind Bc: the owners idle
            // during S4a, and the issue of an instruction whose 64 lanes miss 64 different lines blocks the wave for a while
            // (in front of Bc it cost the step 7 k cycles, profiles/r3e_*); a wave's vector loads return in order, and the next
            // loads of these waves are the requests at the top of the next iteration.  (Round 2 tried touches with the 570-KiB
            // records and lost - three steps of six heads did not fit an XCD's 4-MB L2; with the slim record they are 2.7 MB.)
            unsigned touch = 0u;
            if (p.prefetch && i - 2 >= p.chunk_lo) {
                const int s2 = slot_off(i - 2) + (int)SLOT4_FR;
                if (!(OWN16 && (ow & 64))) touch = __builtin_amdgcn_raw_buffer_load_b32(rS, ow * 128, s2, 0);      // (own16: the second halves of the first two arrays are unused)
                if (ow < 452 - 256) touch += __builtin_amdgcn_raw_buffer_load_b32(rS, (256 + ow) * 128, s2, 0);
                if (ow < 192) {
                    const __amdgpu_buffer_rsrc_t rT = ow < 64 ? rK : ow < 128 ? rQ : rO;
                    touch += __builtin_amdgcn_raw_buffer_load_b32(rT, (ow & 63) * 128, (i - 2) * 8192, 0);
                }
            }
            owner_barrier();                   // Bd
            asm volatile("" :: "v"(touch));    // (keeps the prefetch loads alive; they landed long ago)
            if (DBG && p.dbg != nullptr && blockIdx.x == 0 && ow == 0) t_prev = __builtin_readcyclecounter();
        }

        // ---- dgamma / dbeta: to the next chunk, or reduced over the 64 tokens ----------------------------------------------------
        if (syncw[2] != 0u) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { dgam[k] = __builtin_nanf(""); dbet[k] = __builtin_nanf(""); }
        }
        if (!p.last) {
            if (cq == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 a = {dgam[4 * q], dgam[4 * q + 1], dgam[4 * q + 2], dgam[4 * q + 3]};
                    f32x4 b = {dbet[4 * q], dbet[4 * q + 1], dbet[4 * q + 2], dbet[4 * q + 3]};
                    *reinterpret_cast<f32x4*>(carry + C_DG + (size_t)ow * 16 + 4 * q) = a;
                    *reinterpret_cast<f32x4*>(carry + C_DBT + (size_t)ow * 16 + 4 * q) = b;
                }
            }
        } else {
            float* sg = reinterpret_cast<float*>(smem + L_K);        // [256][16]
            float* sb = reinterpret_cast<float*>(smem + L_G);
#pragma unroll
            for (int k = 0; k < 16; ++k) { sg[ow * 16 + k] = dgam[k]; sb[ow * 16 + k] = dbet[k]; }
            __syncthreads();
            if (ow < 64 && cq == 0) {
                const int o = ow >> 4, k = ow & 15;
                float a = 0.f, b = 0.f;
                for (int t = 0; t < 64; ++t) { a += sg[(t * 4 + o) * 16 + k]; b += sb[(t * 4 + o) * 16 + k]; }
                p.dlnw[(size_t)bh * 64 + ow] = a;
                p.dlnb[(size_t)bh * 64 + ow] = b;
            }
        }
    } else {
        // =========================================================================================================== DERIVERS
        const int pp = wv - DW0;                                // the 32 hidden units of compute wave pp
        const int l = tid & 63, h = l >> 5, c = l & 31;
        const int nO = 64 * cq + 32 * pp;
        DeriverBackend<DBG> bk{smem, (int)(threadIdx.x & 63)};
        if (DBG && p.dbg != nullptr && blockIdx.x == 0 && tid == 64 * DW0) bk.dbg = reinterpret_cast<unsigned long long*>(p.dbg);
        bwd4::AuxState st;
        // the state entering step `step` (a multiple of the checkpoint group size, or the end of the sequence): the forward's
        // checkpoint, or the state phase A wrote after the last step
        auto load_anchor = [&](int step) {
            const float* W2g;
            if (step >= NC) W2g = p.wfinal + (size_t)bh * FINAL_FLOATS + 64 * 256;
            else W2g = p.W2c + ((size_t)bh * p.K + step / p.G) * 256 * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = row_of(r, h);
#pragma unroll
                for (int a = 0; a < 2; ++a) st.W2t[a][r] = W2g[(size_t)(nO + ro) * 64 + 32 * a + c];
            }
        };
        auto load_frags = [&](int step, int arr, bwd4::Frags4& F) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int s = 0; s < 2; ++s) F.f[ti][s] = bld8(rS, l * 16, slot_off(step) + WREG + fro4(arr, fr_idx(ti, pp, s)));
        };
        load_anchor(p.chunk_hi);
        bwd4::Frags4 Z1, Z1B;
        char* const park = p.park + ((size_t)(bh * 4 + cq) * 2 + pp) * bwd4::PARK_BYTES;     // this wave's R4 parking area (L2-resident)
        load_frags(i0, A_Z1, Z1);
        load_frags(i0, A_Z1B, Z1B);
        owner_barrier();                       // P0
        bwd4::stage_w2t(bk, st, pp, L_R3 + 2 * FRK);                  // W2' of the chunk's last step (its output path runs before P2)
        bwd4::derive_z1b(bk, Z1B, pp, L_R3, L_R3 + FRK);
        owner_barrier();                       // P1: the tiles of step i0 (K, gZ2, eta) are visible
        bwd4::reverse_step(bk, st, pp, L_K, L_G, L_SM, Z1, L_R1, L_R2, slots + (size_t)slot_off(i0) + WREG, fro4(A_GZ1, 0), park);
        owner_barrier();                       // P2

        // DEBUG cycle stamps of deriver wave 0 of workgroup 0 (entries 28 .. 31: staging after Bd, derive_z1b, reverse_step, barriers)
        unsigned long long t_d = 0;
#define TTT_DSTAMP(k)                                                                \
        if (DBG && p.dbg != nullptr && blockIdx.x == 0 && tid == 64 * DW0) {              \
            const unsigned long long _t = __builtin_readcyclecounter();              \
            p.dbg[28 + (k)] += _t - t_d;                                             \
            t_d = _t;                                                                \
        }
        if (DBG && p.dbg != nullptr && blockIdx.x == 0 && tid == 64 * DW0) t_d = __builtin_readcyclecounter();
        for (int i = i0; i >= p.chunk_lo; --i) {
            const bool more = i > p.chunk_lo;
            const int nxt = ((i0 - i) & 1) ^ 1;                 // tile buffer of step j = i - 1
            bk.refresh();
            bwd4::stage_r4(bk, pp, L_R4, park);                 // of step i: S4a of the step before is behind Bd / P2
            bwd4::stage_w2t(bk, st, pp, L_R3 + 2 * FRK);        // W2_i^T: S4a of step i, output path of step j
            if (more) load_frags(i - 1, A_Z1B, Z1B);
            TTT_DSTAMP(0)
            owner_barrier();                   // Ba
            TTT_DSTAMP(3)
            if (more) {
                bwd4::derive_z1b(bk, Z1B, pp, L_R3, L_R3 + FRK);
                load_frags(i - 1, A_Z1, Z1);                    // in flight across Bb
            }
            TTT_DSTAMP(1)
            owner_barrier();                   // Bb: K_j, gZ2_j, eta_j staged by the owners are visible
            TTT_DSTAMP(3)
            // Round 6 (p.split): barrier Bc falls INSIDE the reverse step, behind its W2 update - the token tiles run beside the
            // compute waves' S4a (Bc .. Bd), where the derivers used to idle, instead of holding Bc back (see reverse_step's `mid`)
            const bool split = p.split != 0 && more;
            if (more) {
                if (i % p.G == 0) load_anchor(i);               // group boundary: the exact state entering step i
                if constexpr (DBG) bk.t_in = __builtin_readcyclecounter();
                auto mid = [&] {
                    if (split) {
                        TTT_DSTAMP(2)
                        owner_barrier();       // Bc
                        TTT_DSTAMP(3)
                        if constexpr (DBG) bk.t_in = __builtin_readcyclecounter();
                    }
                };
                bwd4::reverse_step(bk, st, pp, L_K + nxt * TILE_B, L_G + nxt * TILE_B, L_SM + nxt * 64 * 4, Z1, L_R1, L_R2,
                                   slots + (size_t)slot_off(i - 1) + WREG, fro4(A_GZ1, 0), park, mid);
            }
            // L2 prefetch of this wave's share of the slice's Z1 / Z1b fragments of step i - 2 (two consecutive 8-KiB arrays)
            unsigned touch = 0u;
            if (p.prefetch && i - 2 >= p.chunk_lo)
                touch = __builtin_amdgcn_raw_buffer_load_b32(rS, (pp * 64 + l) * 128, slot_off(i - 2) + WREG + fro4(A_Z1, 0), 0);
            TTT_DSTAMP(2)
            if (!split) owner_barrier();       // Bc
            owner_barrier();                   // Bd
            asm volatile("" :: "v"(touch));
            TTT_DSTAMP(3)
        }
        if (p.last) owner_barrier();           // (the owners' final reduction)
    }
}


__device__ __forceinline__ bf16x8 ld_frag4(const char* slice, int arr, int idx, int lane) {
    return *reinterpret_cast<const bf16x8*>(slice + fro4(arr, idx) + lane * 16);
}

// =========================================================================================================================
// Tail kernel, group-sequential form (round 6): one workgroup (8 waves, wave w <-> the 32 hidden units [32 w, +32)) per (b, h, checkpoint
// group) walks the group's steps from the top down and REBUILDS what the per-step tail was handed:
//   W1_i   = W1_{i+1} + (eta_i K_i)^T gZ1_i           from the forward's checkpoint of the next group (or the state after the last step),
//   dW1'_i = dW1'_{i+1} + K_{i+1}^T dZ1_{i+1} + Q_i^T dZ1b_i   from the fp32 anchor the sweep stores at the group's top step,
// (rounds 3 - 5: a fully parallel kernel, one workgroup per step, that was HANDED packed images of both: 64 KiB more per step written by the
// sweep, 96 KiB more read here, and the derivers' whole W1 reversal existed for nothing else; removed after the A/B of profiles/r6q_*)
// both as fp32 accumulator tiles in the orientation the contractions over the hidden units want (rows = n, lane = f: in-place A
// operands), with the bf16 operand roundings the sweep's waves apply.  Per step it reads dZ1, dZ1b (compute waves) and gZ1 (derivers; T
// orientation) - 96 KiB where the per-step tail read 192 KiB of fragment arrays - plus the K, Q, dOut, dV tiles:
//   dQ = dOut + dZ1b W1_{i+1}^T          dK = -eta (gZ1 dW1'^T) + dZ1 W1_i^T - dV
struct TailParams5 {
    const __bf16 *XQ, *XK, *dOut, *eta, *dXV;
    char* slots; size_t slot_stride_bh;
    const float *W1c, *wfinal, *danchor;
    __bf16 *dXQ, *dXK;
    int NC, G, K, chunk_lo, group0, ngroups;
};
constexpr int T5_L_K = 0, T5_L_KS = TILE_ELEMS * 2, T5_L_Q = 2 * TILE_ELEMS * 2, T5_L_RED = 3 * TILE_ELEMS * 2;
// EIGHT waves, wave w <-> the 32 hidden units [32 w, +32) (64 registers of state per wave, <= 256 registers: two waves per SIMD - the
// kernel is a chain of dependent loads and MFMAs over 16 sequential steps, and a second wave per SIMD is what hides them; the first cut,
// four waves of 64 units at 380 registers, ran 0.32 ms per chunk beside the sweep and cost the backward 2 % inside the step).  The eight
// partial tiles meet in LDS one 32-feature HALF at a time: [8 waves][64 t][PSH] fp32 = 72 KiB.
constexpr int NT5 = 512, PSH = 36;
constexpr int LDS_TAIL5 = T5_L_RED + 8 * 64 * PSH * 4;
static_assert(LDS_TAIL5 <= 160 * 1024, "LDS budget");
__device__ __forceinline__ void write_partial_half(float* redw, const f32x16 (&P)[2], int h, int c) {      // P[ti]: rows = f in the half, lane = t
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 v = {P[ti][4 * q], P[ti][4 * q + 1], P[ti][4 * q + 2], P[ti][4 * q + 3]};
            *reinterpret_cast<f32x4*>(redw + (32 * ti + c) * PSH + 8 * q + 4 * h) = v;
        }
}
__device__ __forceinline__ f32x4 gather_partial_half(const float* red, int t, int f0) {       // token t, features f0 .. f0 + 3 of the half: sum of the 8 waves
    f32x4 z = *reinterpret_cast<const f32x4*>(red + (size_t)t * PSH + f0);
#pragma unroll
    for (int w = 1; w < 8; ++w) z += *reinterpret_cast<const f32x4*>(red + ((size_t)w * 64 + t) * PSH + f0);
    return z;
}
__device__ __forceinline__ f32x4 ld4_bf16(const __bf16* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4_bf16(__bf16* p, f32x4 z) {
    bf16x4 v = {(__bf16)z[0], (__bf16)z[1], (__bf16)z[2], (__bf16)z[3]};
    *reinterpret_cast<bf16x4*>(p) = v;
}

__global__ __launch_bounds__(NT5) void mlp_bwd_tail5_kernel(TailParams5 p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Kt = reinterpret_cast<__bf16*>(smem + T5_L_K);
    __bf16* Ks = reinterpret_cast<__bf16*>(smem + T5_L_KS);          // eta-scaled K (the W1 update's operand)
    __bf16* Qt = reinterpret_cast<__bf16*>(smem + T5_L_Q);
    float* red = reinterpret_cast<float*>(smem + T5_L_RED);
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, h = l >> 5, c = l & 31;
    const int cq = w >> 1, nj = w & 1;                              // slice region of the step record, half of it
    const int bh = blockIdx.x / p.ngroups, g = p.group0 + (int)(blockIdx.x % p.ngroups);
    const int lo = g * p.G, hi = (lo + p.G < p.NC) ? lo + p.G : p.NC;
    const bf16x8 I0 = ident_pi(0, h, c), I1 = ident_pi(1, h, c);
    const int ot = tid >> 3, of0 = 4 * (tid & 7);                   // owner mapping of the reduction: token, 4 features of the current half
    const int srow = tid >> 3, scol = 8 * (tid & 7);                // staging: 512 threads x 8 elements per tile

    // state tiles [fj]: rows = n in 32 w .., lane = f in 32 fj ..
    f32x16 W1T[2], DT[2];
    {
        const float* W1g = hi >= p.NC ? p.wfinal + (size_t)bh * FINAL_FLOATS : p.W1c + ((size_t)bh * p.K + (g + 1)) * (64 * 256);
        const float* Dg = p.danchor + ((size_t)bh * p.K + g) * (64 * 256);
#pragma unroll
        for (int fj = 0; fj < 2; ++fj)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t o = (size_t)(32 * fj + c) * 256 + 32 * w + row_of(r, h);
                W1T[fj][r] = W1g[o];
                DT[fj][r] = Dg[o];
            }
    }
    for (int i = hi - 1; i >= lo; --i) {
        const size_t tile = (size_t)bh * p.NC + i;
        const char* slot_w = p.slots + (size_t)bh * p.slot_stride_bh + (size_t)(i - p.chunk_lo) * SLOT4_BYTES + (size_t)cq * SLICE_BYTES;
        // ---- stage K, eta K, Q of the step ------------------------------------------------------------------------------------
        {
            float kv[8], qv[8];
            load8_bf16(p.XK + tile * 4096 + (size_t)srow * 64 + scol, kv);
            load8_bf16(p.XQ + tile * 4096 + (size_t)srow * 64 + scol, qv);
            const float e = (float)p.eta[tile * 64 + srow];
            store8_bf16(Kt + srow * TS + scol, kv);
            store8_bf16(Qt + srow * TS + scol, qv);
#pragma unroll
            for (int k = 0; k < 8; ++k) kv[k] *= e;
            store8_bf16(Ks + srow * TS + scol, kv);
        }
        __syncthreads();
        // ---- dW1' of this step: + Q_i^T dZ1b_i (the anchor of the group's top step has it already) ------------------------------------
        if (i < hi - 1) {
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 zf = ld_frag4(slot_w, A_DZ1B, fr_idx(ti, nj, s), l);
                    DT[0] = mma(zf, tr_pi(Qt, 32 * ti, s, 0, l), DT[0]);
                    DT[1] = mma(zf, tr_pi(Qt, 32 * ti, s, 32, l), DT[1]);
                }
        }
        f32x16 PA[2][2];                             // [fj][ti]  partial (rows = f, lane = t) over this wave's 32 hidden units
        // ---- dQ partial = W1_{i+1} dZ1b^T ---------------------------------------------------------------------------------------
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const f32x16 zt = transpose_tile(ld_frag4(slot_w, A_DZ1B, fr_idx(ti, nj, 0), l), ld_frag4(slot_w, A_DZ1B, fr_idx(ti, nj, 1), l), I0, I1);
            PA[0][ti] = mma(pack(W1T[0], 0), pack(zt, 0), zero16());
            PA[1][ti] = mma(pack(W1T[1], 0), pack(zt, 0), zero16());
            PA[0][ti] = mma(pack(W1T[0], 1), pack(zt, 1), PA[0][ti]);
            PA[1][ti] = mma(pack(W1T[1], 1), pack(zt, 1), PA[1][ti]);
        }
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
            if (fj) __syncthreads();                 // the owners of the first half have read `red`
            write_partial_half(red + (size_t)w * 64 * PSH, PA[fj], h, c);
            __syncthreads();
            const size_t off = tile * 4096 + (size_t)ot * 64 + 32 * fj + of0;
            st4_bf16(p.dXQ + off, gather_partial_half(red, ot, of0) + ld4_bf16(p.dOut + off));
        }
        // ---- dK partial: -eta (gZ1 dW1'^T), then W1_i, then + dZ1 W1_i^T ---------------------------------------------------------
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const bf16x8 g0 = ld_frag4(slot_w, A_GZ1, fr_idx(ti, nj, 0), l), g1 = ld_frag4(slot_w, A_GZ1, fr_idx(ti, nj, 1), l);   // gZ1, T
            const f32x16 gt = transpose_tile(g0, g1, I0, I1);                                                                      // gZ1, N
            PA[0][ti] = mma(pack(DT[0], 0), pack(gt, 0), zero16());
            PA[1][ti] = mma(pack(DT[1], 0), pack(gt, 0), zero16());
            PA[0][ti] = mma(pack(DT[0], 1), pack(gt, 1), PA[0][ti]);
            PA[1][ti] = mma(pack(DT[1], 1), pack(gt, 1), PA[1][ti]);
            const float el = -(float)p.eta[tile * 64 + 32 * ti + c];
#pragma unroll
            for (int r = 0; r < 16; ++r) { PA[0][ti][r] *= el; PA[1][ti][r] *= el; }
            // W1_i = W1_{i+1} + (eta K)^T gZ1 : A = gZ1 (T fragment in place: m = n lane, k = t), B = eta K by transposed reads
            W1T[0] = mma(g0, tr_pi(Ks, 32 * ti, 0, 0, l), W1T[0]);
            W1T[1] = mma(g0, tr_pi(Ks, 32 * ti, 0, 32, l), W1T[1]);
            W1T[0] = mma(g1, tr_pi(Ks, 32 * ti, 1, 0, l), W1T[0]);
            W1T[1] = mma(g1, tr_pi(Ks, 32 * ti, 1, 32, l), W1T[1]);
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            const bf16x8 z0 = ld_frag4(slot_w, A_DZ1, fr_idx(ti, nj, 0), l), z1 = ld_frag4(slot_w, A_DZ1, fr_idx(ti, nj, 1), l);     // dZ1, T
            const f32x16 zt = transpose_tile(z0, z1, I0, I1);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 zb = pack(zt, s);
                PA[0][ti] = mma(pack(W1T[0], s), zb, PA[0][ti]);
                PA[1][ti] = mma(pack(W1T[1], s), zb, PA[1][ti]);
            }
            // dW1' of the step below: + K_i^T dZ1_i
            DT[0] = mma(z0, tr_pi(Kt, 32 * ti, 0, 0, l), DT[0]);
            DT[1] = mma(z0, tr_pi(Kt, 32 * ti, 0, 32, l), DT[1]);
            DT[0] = mma(z1, tr_pi(Kt, 32 * ti, 1, 0, l), DT[0]);
            DT[1] = mma(z1, tr_pi(Kt, 32 * ti, 1, 32, l), DT[1]);
        }
#pragma unroll
        for (int fj = 0; fj < 2; ++fj) {
            __syncthreads();                         // the owners of the pass before have read `red`
            write_partial_half(red + (size_t)w * 64 * PSH, PA[fj], h, c);
            __syncthreads();
            const size_t off = tile * 4096 + (size_t)ot * 64 + 32 * fj + of0;
            st4_bf16(p.dXK + off, gather_partial_half(red, ot, of0) - ld4_bf16(p.dXV + off));      // dK -= dt, dt = dV
        }
        __syncthreads();                             // `red` and the tiles are free for the next step
    }
}
}  // namespace b4

// ---------------------------------------------------------------------------------------------------------------------------
// The error word lives in host-mapped memory: the kernels store to it with system scope, the host reads it without a copy -
// at the entry of every TTT-MLP call without synchronising (a hand-over that gave up makes the NEXT call fail), or after a
// device synchronisation when a test / bench asks.
// One word for the process (a time-out on ANY device makes the next call fail), mapped into every device that asks: the host
// allocation is portable, the device pointer is looked up per device.
static unsigned* g_err_host = nullptr;
static unsigned* g_err_dev[16] = {nullptr};
static std::mutex g_err_mutex;
unsigned* sweep_error_word() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    std::lock_guard<std::mutex> lock(g_err_mutex);
    if (!g_err_host) {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return nullptr;
        *(volatile unsigned*)h = 0u;
        g_err_host = (unsigned*)h;
    }
    if (!g_err_dev[dev]) {
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, g_err_host, 0) != hipSuccess) return nullptr;
        g_err_dev[dev] = (unsigned*)d;
    }
    return g_err_dev[dev];
}
unsigned peek_sweep_error() { return g_err_host ? *(volatile unsigned*)g_err_host : 0u; }
unsigned read_sweep_error() {
    (void)hipDeviceSynchronize();
    return peek_sweep_error();
}
void clear_sweep_error() {
    (void)hipDeviceSynchronize();
    if (g_err_host) *(volatile unsigned*)g_err_host = 0u;
}
unsigned read_sweep_fast_count() {      // DEBUG statistic: cluster workgroup launches that published plain (same-XCD) records (synchronises)
    unsigned v = 0;
    (void)hipMemcpyFromSymbol(&v, HIP_SYMBOL(b4::g_fast_count4), sizeof(v));
    return v;
}

// (the owners' partner-independent arithmetic under the record loads - template parameter OVL; one box, NC = 804: with fp32 records
// 14.15 ms per backward against 13.44 without, with bf16 records 11.76 against 12.71 - goes with the bf16 records)
// Shipped instantiation (every parameter decided by an interleaved A/B on one MI355X, profiles/r4b - r4h; the losing
// instantiations and their options were removed in round 5): bf16 hand-over records with the owners' partner-independent arithmetic
// under the record loads (11.82 against 14.16 ms per backward at NC = 804), derivers on waves 2, 3 = SIMDs 2 / 3 beside two owner
// waves (11.48 - 11.62 against 11.79 - 11.94), bf16 inner-LayerNorm owner rows (11.35 against 11.63).

namespace s4 {

void launch_sweep_cluster4(const SweepParams4& bp, int nbh, hipStream_t s) {
    static bool attr[64] = {false};           // per device: a function attribute is a property of the function ON a device
    int dev = 0;
    (void)hipGetDevice(&dev);
    const dim3 grid(nbh * 4), blk(b4::NTC);
    {
        std::lock_guard<std::mutex> lock(g_err_mutex);
        if (dev >= 0 && dev < 64 && !attr[dev]) {
            auto set = [&](auto kern) { (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, b4::LDS_CL); };
            set(b4::mlp_bwd_cluster4_kernel<false, true, true, 2, true>);    set(b4::mlp_bwd_cluster4_kernel<true, true, true, 2, true>);
            (void)hipFuncSetAttribute((const void*)b4::mlp_bwd_tail5_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, b4::LDS_TAIL5);
            attr[dev] = true;
        }
    }
    auto go = [&](auto kern) { hipLaunchKernelGGL(kern, grid, blk, b4::LDS_CL, s, bp); };
    if (bp.dbg != nullptr) go(b4::mlp_bwd_cluster4_kernel<true, true, true, 2, true>);        // (stage stamps: tools/op_bench.py --phases)
    else go(b4::mlp_bwd_cluster4_kernel<false, true, true, 2, true>);
}

void launch_tail5(const Tail5Args& a, int nbh, hipStream_t s) {
    b4::TailParams5 tp = {a.XQ, a.XK, a.dOut, a.eta, a.dXV, a.slots, a.slot_stride_bh, a.W1c, a.wfinal, a.danchor, a.dXQ, a.dXK,
                          a.NC, a.G, a.K, a.chunk_lo, a.group0, a.ngroups};
    hipLaunchKernelGGL(b4::mlp_bwd_tail5_kernel, dim3(nbh * a.ngroups), dim3(b4::NT5), b4::LDS_TAIL5, s, tp);
}
}  // namespace s4

}  // namespace mfma
}  // namespace ttt
