// MFMA TTT-MLP backward for gfx950, revision 2: 8-wave reverse sweep + parallel dK / dQ tail kernel.
//
// Why a second revision (profiles/r1b): the 4-wave / 512-register sweep (ttt_mfma_bwd.hip) spends 70 % of its wave
// cycles waiting (one wave per SIMD: every slot load and every barrier is exposed), 10 % of the occupied SIMDs' MFMA
// time is used, 20 of its MFMAs per wave-step are layout transposes, and it carries work that does not depend on the
// sequential state at all.  Here
//   * the sequential kernel is 8 waves (2 per SIMD, <= 256 VGPRs, VGPR-form MFMA), wave (w, p) owning the 32 hidden
//     units Hp = [64w + 32p, +32): dW1[:, Hp] as tiles (rows = f, lane = n) and dW2[Hp, :] in BOTH orientations
//     (rows = n, lane = f) and (rows = f, lane = n), each updated by its own MFMAs - so every contraction of the step
//     runs over a tile's row index (in-place operand re-use, ttt_mfma_dev.h) with no transposes;
//   * everything elementwise is evaluated in the orientation its consumer contracts over: the (K dW1) and (gZ2 dW2^T)
//     products are formed twice, (rows = n, lane = t) for the token-wise reductions / the contraction over hidden units
//     and (rows = t, lane = n) for the contractions over tokens - an MFMA is cheaper than any cross-lane shuffle;
//   * the contraction over the 256 hidden units of d(gZ2) is reduced over the 4 hidden slices through LDS partials
//     (pairs (w,0),(w,1) first exchange the 32-unit operand fragments they miss: u^T and the packed dW2 block);
//   * dK and dQ leave the sequential kernel: they need the carried dW1 and dZ1 of their step but nothing downstream
//     needs them, so the sweep stores those two (bf16 fragment images) and a fully parallel tail kernel (one workgroup
//     per step) finishes dK = -eta (gZ1 dW1'^T) + dZ1 W1^T - dt and dQ = dOut + dZ1b W1'^T.
// Math: SURVEY.md Appendix A backward; oracle/ttt_oracle.py:_mlp_step_bwd is the executable spec.
//
// Per step i of the sweep (5 workgroup barriers; "step j" = i - 1 is the next one to be processed):
//   S1  E1^T = dW1^T K^T, A2^T = dW2 gZ2^T (rows = n, lane = t); u^T = -eta (E1 + db1) gelu'(Z1);
//       d(eta) partial = -colsum(X2^T A2^T + gZ1^T (E1^T + db1)); first half of d(gZ2)^T = -eta (dW2^T X2^T);
//       publish u^T fragments                                                                              | Ba
//   S2  d(gZ2)^T partial += W2^T u^T (own + partner fragments) -> LDS partials                              | Bb
//   S3  owners: reduce 4 partials, backward of the fused LN / L2 gradient -> dZ2 (LDS), dV, d(eta), dgamma, dbeta;
//       output LayerNorm backward of step j -> dZ2b_j (LDS).  Waves: E1, A2 again in (rows = t, lane = n)  | Bc
//   S4a u = -eta (E1 + db1) gelu'(Z1); dX2 = -eta A2 + dZ2 W2^T; dZ1 = dgZ1 M + dX2 gelu'(Z1);
//       dW1 += K^T dZ1; dW2 += u^T gZ2 + X2^T dZ2 (both orientations); db1, db2; dZ1 -> slot                | Bd
//   S4b step j's output path: dZ1b = (dZ2b W2'^T) gelu'(Z1b); dW1 += Q^T dZ1b; dW2 += X2b^T dZ2b; db1, db2;
//       dZ1b and the now complete dW1 -> slot j (tail kernel + next S1); publish dW2 block; park K, gZ2, eta | Be
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"
#include "ttt_mfma_bwd_dev.h"

namespace ttt {
namespace mfma {
using namespace ttt::mf;

namespace b2 {
// ---- LDS map ------------------------------------------------------------------------------------------------------------
constexpr int TILE_B = TILE_ELEMS * 2;                    // 9216 bytes
constexpr int L_K = 0;                                    // K_i          [t][f]
constexpr int L_G = L_K + TILE_B;                         // gZ2_i        [t][f]
constexpr int L_Q = L_G + TILE_B;                         // Q_j          [t][f]   (j = i - 1)
constexpr int L_A = L_Q + TILE_B;                         // dZ2b_j       [t][f]
constexpr int L_XU = L_A + TILE_B;                        // u^T exchange (8 waves x 4 fragments x 1 KiB); later dZ2_i [t][f]
constexpr int XU_BYTES = 8 * 4 * 1024;
constexpr int L_XD = L_XU + XU_BYTES;                     // dW2 block exchange (8 waves x 2 fragments)
constexpr int XD_BYTES = 8 * 2 * 1024;
constexpr int L_RED = L_XD + XD_BYTES;                    // [4][64][PS] fp32
constexpr int RED_B = 4 * 64 * PS * 4;
constexpr int L_SM = L_RED + RED_B;                       // eta[64], db1[256], db2[64], gamma[64], etaP[8][64]
constexpr int LDS_SWEEP = L_SM + (64 + 256 + 64 + 64 + 8 * 64) * 4;
static_assert(LDS_SWEEP <= 160 * 1024, "LDS budget");
static_assert(TILE_B <= XU_BYTES, "dZ2 tile aliases the u exchange");

#define TTT_STAMP3(k)                                                        \
    if (DBG && p.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {    \
        const unsigned long long _t = __builtin_readcyclecounter();          \
        p.dbg[16 + (k)] += _t - t_last;                                      \
        t_last = _t;                                                         \
    }

struct Stage {            // next step's tiles, register-staged: one 16-byte chunk per thread per tile
    uint4 k, g, q;
    float eta;
};


template <bool DBG>
__global__ __launch_bounds__(NT2) void mlp_bwd_sweep8_kernel(SweepParams2 p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Kt = reinterpret_cast<__bf16*>(smem + L_K);
    __bf16* Gt = reinterpret_cast<__bf16*>(smem + L_G);
    __bf16* Qt = reinterpret_cast<__bf16*>(smem + L_Q);
    __bf16* At = reinterpret_cast<__bf16*>(smem + L_A);
    __bf16* Bt = reinterpret_cast<__bf16*>(smem + L_XU);
    char* exu = smem + L_XU;
    char* exd = smem + L_XD;
    float* red = reinterpret_cast<float*>(smem + L_RED);
    float* etaL = reinterpret_cast<float*>(smem + L_SM);
    float* db1L = etaL + 64;
    float* db2L = db1L + 256;
    float* gamL = db2L + 64;
    float* etaP = gamL + 64;

    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int w = wv >> 1, pp = wv & 1;
    const int nO = 64 * w + 32 * pp;
    const int fO = 32 * pp, fX = 32 * (1 - pp);
    const int bh = blockIdx.x % p.nbh, head = bh % p.NH;
    // ---- prefetch helpers ------------------------------------------------------------------------------------------------
    // A step reads ~430 KiB of slot data and ONE CU sustains only ~10 bytes/cycle of HBM misses (about 64 lines in flight
    // x 900 cycles), which alone is 19 us per step; 208 of the 256 CUs are idle while the 48 scans run.  So `helpers`
    // extra workgroups per (b,h) - launched so that they share the scan's XCD (block % 8, a speed-only assumption) - walk
    // the same slots one step ahead of the scan and pull them into that XCD's L2 with their own miss parallelism; the scan
    // then runs on L2 hits.  Helpers only read; the scan never waits for them; they follow its progress word with relaxed
    // loads and give up after a bounded number of polls, so correctness and termination do not depend on them.
    if (blockIdx.x >= p.nbh) {
        const int hid = blockIdx.x / p.nbh - 1;                    // 0 .. helpers-1
        const char* base = p.slots + (size_t)bh * p.slot_stride_bh;
        const char* kq[3] = {reinterpret_cast<const char*>(p.XK), reinterpret_cast<const char*>(p.XQ), reinterpret_cast<const char*>(p.dOut)};
        constexpr int FR_LINES = 4 * 10 * 64;                      // 4 wave regions x 10 arrays x 64 lines
        constexpr int OWN_LINES = (int)((SLOT_OWN + SLOT_G) / 128);
        constexpr int ALL_LINES = FR_LINES + OWN_LINES + 3 * 64;
        unsigned sink = 0;
        for (int t = p.chunk_hi - 1; t >= p.chunk_lo; --t) {
            int polls = 0;
            while (true) {                                         // stay at most one step ahead of the scan
                const int cur = __hip_atomic_load(p.prog + bh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur <= t + p.lead || ++polls > 200000) break;
                __builtin_amdgcn_s_sleep(8);
            }
            const char* sl = base + (size_t)(t - p.chunk_lo) * SLOT_BYTES;
            for (int ln = hid * NT2 + threadIdx.x; ln < ALL_LINES; ln += p.helpers * NT2) {
                const char* a;
                if (ln < FR_LINES) {
                    const int wr = ln / 640, rem = ln % 640, ai = rem >> 6, li = rem & 63;
                    const int arr = ai < 4 ? ai + 1 : ai == 4 ? 6 : ai + 3;      // FR_W2..FR_D1, FR_GX2, FR_GZ1T..FR_D1N
                    a = sl + (size_t)wr * SLOT_WAVE_FR + (size_t)arr * 8 * FRAG_BYTES + (size_t)li * 128;
                } else if (ln < FR_LINES + OWN_LINES) {
                    a = sl + SLOT_FR + (size_t)(ln - FR_LINES) * 128;
                } else {
                    const int r = ln - FR_LINES - OWN_LINES;
                    a = kq[r >> 6] + (((size_t)bh * p.NC + t) * 4096) * 2 + (size_t)(r & 63) * 128;
                }
                sink ^= *reinterpret_cast<const unsigned*>(a);
            }
        }
        asm volatile("" :: "v"(sink));
        return;
    }
    const int NC = p.NC;
    char* slots = p.slots + (size_t)bh * p.slot_stride_bh;
    float* carry = p.carry + (size_t)bh * CARRY_FLOATS2;
    const __amdgpu_buffer_rsrc_t rS = make_srd(slots, p.slot_stride_bh);                       // this (b,h)'s slot area
    const size_t act_bytes = (size_t)NC * 4096 * 2;
    const __amdgpu_buffer_rsrc_t rK = make_srd(p.XK + (size_t)bh * NC * 4096, act_bytes);
    const __amdgpu_buffer_rsrc_t rQ = make_srd(p.XQ + (size_t)bh * NC * 4096, act_bytes);
    const __amdgpu_buffer_rsrc_t rO = make_srd(p.dOut + (size_t)bh * NC * 4096, act_bytes);
    const __amdgpu_buffer_rsrc_t rV = make_srd(p.dXV + (size_t)bh * NC * 4096, act_bytes);
    const int WREG = w * (int)SLOT_WAVE_FR;                                                     // this wave pair's fragment region
    auto slot_off = [&](int step) { return (step - p.chunk_lo) * (int)SLOT_BYTES; };

    // ---- carried state gradient ---------------------------------------------------------------------------------------
    f32x16 dW1t[2];      // [a]  dW1[f in 32a.., n in Hp]                 (rows = f, lane = n)
    f32x16 dW2t[2];      // [0] dW2[n in Hp, f in Fp], [1] dW2[n in Hp, f in Fx]      (rows = n, lane = f)
    f32x16 dW2Tt[2];     // same blocks transposed                                      (rows = f, lane = n)
    float db1v, db2v = 0.f;   // db1[nO + c] ; db2[fO + c] (waves with w == 0)
    float dgam[8], dbet[8];
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
        if (w == 0) db2v = gb2[fO + c];
        if (p.first) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { dgam[j] = 0.f; dbet[j] = 0.f; }
        } else {
            load8_f32(carry + C_DG + (size_t)tid * 8, dgam);
            load8_f32(carry + C_DBT + (size_t)tid * 8, dbet);
        }
        if (tid < 64) gamL[tid] = p.ln_w[(size_t)head * 64 + tid];
    }
    bf16x8 ONES;
#pragma unroll
    for (int e = 0; e < 8; ++e) ONES[e] = (__bf16)1.0f;

    // ---- staging helpers (lambdas keep the index arithmetic in one place) ---------------------------------------------
    auto stage_issue = [&](Stage& st, int step, bool with_kg, bool with_q) {
        const int t16 = tid * 16;                  // a [64][64] bf16 tile is 512 threads x 16 contiguous bytes
        if (with_kg) {
            st.k = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rK, t16, step * 8192, 0));
            st.g = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rS, t16, slot_off(step) + (int)(SLOT_FR + SLOT_OWN), 0));
            st.eta = tid < 64 ? (float)p.eta[((size_t)bh * NC + step) * 64 + tid] : 0.f;
        }
        if (with_q) st.q = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rQ, t16, step * 8192, 0));
    };
    auto park_kg = [&](const Stage& st) {
        const int prow = tid >> 3, pcol = (tid & 7) * 8;
        *reinterpret_cast<uint4*>(Kt + prow * TS + pcol) = st.k;
        *reinterpret_cast<uint4*>(Gt + prow * TS + pcol) = st.g;
        if (tid < 64) etaL[tid] = st.eta;
    };
    auto park_q = [&](const Stage& st) {
        const int prow = tid >> 3, pcol = (tid & 7) * 8;
        *reinterpret_cast<uint4*>(Qt + prow * TS + pcol) = st.q;
    };
    // owners: backward of the output LayerNorm of step j -> dZ2b_j tile (At), dgamma / dbeta contributions
    auto owner_out_ln = [&](int j) {
        const int ot = tid >> 3, of0 = 8 * (tid & 7);
        const int so = slot_off(j) + (int)SLOT_FR;
        float d[8], xl[8], g[8];
        {
            const bf16x8 dv = bld8(rO, tid * 16, j * 8192);
#pragma unroll
            for (int k = 0; k < 8; ++k) d[k] = (float)dv[k];
        }
        bld8f(rS, tid * 32, so + 2 * (int)SLOT_OWN_ARR, xl);
        const float rstdl = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, ot * 8 + 4, so + 3 * (int)SLOT_OWN_ARR, 0));
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            dgam[k] += d[k] * xl[k];
            dbet[k] += d[k];
            g[k] = d[k] * gamL[of0 + k];
            s1 += g[k]; s2 += g[k] * xl[k];
        }
        s1 = sum8(s1); s2 = sum8(s2);
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = (64.0f * g[k] - s1 - xl[k] * s2) * rstdl * (1.0f / 64.0f);
        store8_bf16(At + ot * TS + of0, g);
    };
    // waves: output path of step j, W2' = state entering step j + 1 (slot j + 1), X2b / gelu'(Z1b) from slot j
    auto add_output_path = [&](int j) {
        const int l = tid & 63, h = l >> 5, c = l & 31;
        const int sj = slot_off(j) + WREG, sn = sj + (int)SLOT_BYTES, l16 = l * 16;
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            f32x16 dz = zero16();
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                dz = mma(row_pi(At, ti, fO, s, l), bld8(rS, l16, sn + fro(FR_W2T, fr_idx(pp, pp, s))), dz);
                dz = mma(row_pi(At, ti, fX, s, l), bld8(rS, l16, sn + fro(FR_W2T, fr_idx(1 - pp, pp, s))), dz);
            }
            const f32x16 d1b = unpack2(bld8(rS, l16, sj + fro(FR_D1B, fr_idx(ti, pp, 0))), bld8(rS, l16, sj + fro(FR_D1B, fr_idx(ti, pp, 1))));
#pragma unroll
            for (int r = 0; r < 16; ++r) dz[r] *= d1b[r];
            db1v += tile_colsum(dz);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 zf = pack(dz, s);                       // dZ1b (k = t rows, j = n lane)
                bst8(rS, l16, sj + fro(FR_DZ1B, fr_idx(ti, pp, s)), zf);
                dW1t[0] = mma(tr_pi(Qt, 32 * ti, s, 0, l), zf, dW1t[0]);
                dW1t[1] = mma(tr_pi(Qt, 32 * ti, s, 32, l), zf, dW1t[1]);
                const bf16x8 xb = bld8(rS, l16, sj + fro(FR_X2B, fr_idx(ti, pp, s)));   // X2b (m = n lane, k = t rows)
                const bf16x8 aO = tr_pi(At, 32 * ti, s, fO, l), aX = tr_pi(At, 32 * ti, s, fX, l);
                dW2t[0] = mma(xb, aO, dW2t[0]);
                dW2t[1] = mma(xb, aX, dW2t[1]);
                dW2Tt[0] = mma(aO, xb, dW2Tt[0]);
                dW2Tt[1] = mma(aX, xb, dW2Tt[1]);
                if (w == 0) {                                         // db2 += column sums of dZ2b (ones MFMA)
                    f32x16 acc = mma(ONES, aO, zero16());
                    db2v += acc[0];
                }
            }
        }
        (void)h; (void)c;
    };
    // publish what the next S1 needs from this wave: complete dW1 (slot j, also the tail kernel's operand), db1, db2,
    // the dW2 block the partner contracts over
    auto publish_state = [&](int j) {
        const int l = tid & 63, h = l >> 5, c = l & 31;
        const int sj = slot_off(j) + WREG;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int s = 0; s < 2; ++s) bst8(rS, l * 16, sj + fro(FR_DW1, fr_idx(a, pp, s)), pack(dW1t[a], s));
#pragma unroll
        for (int s = 0; s < 2; ++s) *reinterpret_cast<bf16x8*>(exd + ((size_t)(wv * 2 + s) * 64 + l) * 16) = pack(dW2t[1], s);
        if (h == 0) db1L[nO + c] = db1v;
        if (w == 0 && h == 0) db2L[fO + c] = db2v;
    };

    // ---- prologue: tiles of the first step, its output path, published state ---------------------------------------------
    const int i0 = p.chunk_hi - 1;
    if (tid == 0 && p.helpers) __hip_atomic_store(p.prog + bh, i0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    Stage st;
    stage_issue(st, i0, true, true);
    park_kg(st);
    park_q(st);
    __syncthreads();
    owner_out_ln(i0);
    __syncthreads();
    add_output_path(i0);
    publish_state(i0);
    __syncthreads();

    unsigned long long t_last = __builtin_readcyclecounter();
    for (int i = i0; i >= p.chunk_lo; --i) {
        int l_op = tid & 63;
        asm volatile("" : "+v"(l_op));           // opaque lane id: keeps address arithmetic inside the loop (no hoist + spill)
        const int l = l_op, h = l >> 5, c = l & 31;
        const bool more = i > p.chunk_lo;
        const size_t tile = (size_t)bh * NC + i;
        const int sI = slot_off(i), sw = sI + WREG, l16 = l * 16;     // byte offsets of slot i / this wave's region in it
        if (tid == 0 && p.helpers) __hip_atomic_store(p.prog + bh, i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (more) stage_issue(st, i - 1, false, true);          // Q_j now (parked after S2); K, gZ2, eta of step j at S3
        // (An L2 prefetch of step j's slot was tried here in two forms - dword touches kept in registers and LDS-DMA touches
        // without destination registers - and both lose: a step reads ~430 KiB of slot data, and ONE CU sustains only ~10
        // bytes/cycle of HBM misses (~64 lines in flight x ~900 cycles), so the prefetch itself costs ~40 k cycles per step.
        // The sweep is bound by per-CU miss parallelism, not by exposed latency: see DESIGN.md 4.)

        // ================= S1 : (rows = n, lane = t) products, u^T, d(eta) partial, first half of d(gZ2)^T ==============
        // Three operand-set blocks, each walking both token tiles.  (Scheduling fences between them, and the output path of step
        // i - 1 moved before barrier Bd, were A/B-ed in round 1 - 9.6 / 8.32 vs 8.26 ms per backward - and removed in round 2.)
        f32x16 P[2];                       // [ti]  d(gZ2)^T partial (rows = f in Fp, lane = t)
        bf16x8 uN[2][2];                   // [ti][s]  u^T (k = n rows, j = t lane)
        float se2[2];
        {
            const f32x16 db1R = rows_from_lds(db1L + nO, 0, h);
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
                for (int s = 0; s < 2; ++s) {      // fragment by fragment: 8 rows of the tile at a time
                    const bf16x8 g1 = bld8(rS, l16, sw + fro(FR_GZ1T, fr_idx(pp, ti, s)));
                    const bf16x8 d1 = bld8(rS, l16, sw + fro(FR_D1N, fr_idx(pp, ti, s)));
                    bf16x8 uf;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float ev = e1[8 * s + e];
                        se += (float)g1[e] * ev;
                        uf[e] = (__bf16)(ec * ev * (float)d1[e]);
                    }
                    uN[ti][s] = uf;
                    *reinterpret_cast<bf16x8*>(exu + ((size_t)(wv * 4 + ti * 2 + s) * 64 + l) * 16) = uf;
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
                    const bf16x8 x2 = bld8(rS, l16, sw + fro(FR_XT, fr_idx(pp, ti, s)));
#pragma unroll
                    for (int e = 0; e < 8; ++e) se += (float)x2[e] * a2[8 * s + e];
                }
                se = xor_add(se, 32);
                if (h == 0) etaP[wv * 64 + 32 * ti + c] = -se;
            }
        }
        {
            const bf16x8 D2o0 = pack(dW2t[0], 0), D2o1 = pack(dW2t[0], 1);
            const bf16x8 D2x0 = *reinterpret_cast<const bf16x8*>(exd + ((size_t)((wv ^ 1) * 2 + 0) * 64 + l) * 16);
            const bf16x8 D2x1 = *reinterpret_cast<const bf16x8*>(exd + ((size_t)((wv ^ 1) * 2 + 1) * 64 + l) * 16);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 pa = zero16();
                pa = mma(D2o0, bld8(rS, l16, sw + fro(FR_XT, fr_idx(pp, ti, 0))), pa);
                pa = mma(D2o1, bld8(rS, l16, sw + fro(FR_XT, fr_idx(pp, ti, 1))), pa);
                pa = mma(D2x0, bld8(rS, l16, sw + fro(FR_XT, fr_idx(1 - pp, ti, 0))), pa);
                pa = mma(D2x1, bld8(rS, l16, sw + fro(FR_XT, fr_idx(1 - pp, ti, 1))), pa);
                const float ec = -etaL[32 * ti + c];
#pragma unroll
                for (int r = 0; r < 16; ++r) pa[r] *= ec;
                P[ti] = pa;
            }
        }
        TTT_STAMP3(0)
        __syncthreads();                   // Ba: u^T fragments visible
        TTT_STAMP3(1)

        // ================= S2 : second half of d(gZ2)^T partial -> LDS ===============================================
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 ux = *reinterpret_cast<const bf16x8*>(exu + ((size_t)((wv ^ 1) * 4 + ti * 2 + s) * 64 + l) * 16);
                P[ti] = mma(bld8(rS, l16, sw + fro(FR_W2, fr_idx(pp, pp, s))), uN[ti][s], P[ti]);
                P[ti] = mma(bld8(rS, l16, sw + fro(FR_W2, fr_idx(1 - pp, pp, s))), ux, P[ti]);
            }
            write_partial2(red + (size_t)w * 64 * PS, P[ti], ti, pp, h, c);
        }
        if (more) park_q(st);              // Q_j: its buffer was last read in the previous S4b
        TTT_STAMP3(2)
        __syncthreads();                   // Bb: partials visible; every read of the u exchange is done

        TTT_STAMP3(3)
        // ================= S3 : owners ====================================================================================
        {
            const int ot = tid >> 3, of0 = 8 * (tid & 7);
            const int so = sI + (int)SLOT_FR;
            float G_[8], xh[8], go[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) G_[k] = 0.f;
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) add8_f32(red + ((size_t)ww * 64 + ot) * PS + of0, G_);
            bld8f(rS, tid * 32, so, xh);
            bld8f(rS, tid * 32, so + (int)SLOT_OWN_ARR, go);
            const float r = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, ot * 8, so + 3 * (int)SLOT_OWN_ARR, 0));
            const float eta_t = etaL[ot];
            float gxh[8], gz[8];
            float s1g = 0.f, s2g = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                gxh[k] = go[k] * gamL[of0 + k];
                s1g += gxh[k]; s2g += gxh[k] * xh[k];
            }
            s1g = sum8(s1g); s2g = sum8(s2g);
            float se = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                gz[k] = (64.0f * gxh[k] - s1g - xh[k] * s2g) * r * (1.0f / 64.0f);      // gZ2 (fp32)
                const float db2 = db2L[of0 + k];
                se += gz[k] * db2;
                G_[k] -= eta_t * db2;                                                    // d(gZ2) complete
                const float m = -G_[k] * r;
                s1 += m; s2 += m * xh[k];
            }
            se = sum8(se); s1 = sum8(s1); s2 = sum8(s2);
            float a1 = 0.f, a2 = 0.f, dxh[8], dyv[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float g = gamL[of0 + k];
                const float m = -G_[k] * r;
                const float dgxh = r * G_[k] + (s1 + xh[k] * s2) * (1.0f / 64.0f);
                const float dy = g * dgxh;
                dgam[k] += go[k] * dgxh + dy * xh[k];
                dbet[k] += dy;
                dyv[k] = -dy;                                                            // dt = -dy ; dV = dt
                dxh[k] = dy * g + (gxh[k] * s2 + s2g * m) * (1.0f / 64.0f);
                const float dstd = -dxh[k] * xh[k] * r - G_[k] * gz[k] * r;
                a1 += dxh[k]; a2 += dstd;
            }
            a1 = sum8(a1); a2 = sum8(a2);
#pragma unroll
            for (int k = 0; k < 8; ++k) G_[k] = dxh[k] * r - a1 * r * (1.0f / 64.0f) + a2 * xh[k] * (1.0f / 64.0f);   // dZ2
            store8_bf16(Bt + ot * TS + of0, G_);
            {
                bf16x8 dv;
#pragma unroll
                for (int k = 0; k < 8; ++k) dv[k] = (__bf16)dyv[k];
                bst8(rV, tid * 16, i * 8192, dv);
            }
            if ((tid & 7) == 0) {
                float de = -se;
#pragma unroll
                for (int ww = 0; ww < 8; ++ww) de += etaP[ww * 64 + ot];
                p.deta[tile * 64 + ot] = (__bf16)de;
            }
            if (more) owner_out_ln(i - 1);
        }
        TTT_STAMP3(4)
        __syncthreads();                   // Bc: dZ2 (Bt) and dZ2b_j (At) visible
        TTT_STAMP3(5)

        // ================= S4a : first-layer gradients and this step's state updates ====================================
        // Per token tile: (rows = t, lane = n) products E1 = K dW1, A2 = gZ2 dW2^T (operands packed just in time), the
        // elementwise chain, then every MFMA that consumes this tile's u / dZ1 / X2.  db1 is read (old value) by both tiles
        // before either adds to it.
        if (more) stage_issue(st, i - 1, true, false);           // K_j, gZ2_j, eta_j (L2 hits: touched in S1); parked in S4b
        {
            const float db1_old = db1v;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const f32x16 etaR = rows_from_lds(etaL, 32 * ti, h);
                f32x16 e1 = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    e1 = mma(row_pi(Kt, ti, 0, s, l), pack(dW1t[0], s), e1);
                    e1 = mma(row_pi(Kt, ti, 32, s, l), pack(dW1t[1], s), e1);
                }
                bf16x8 d1f[2], uf[2];                                       // gelu'(Z1) fragments ; u (m = n lane, k = t rows)
                f32x16 dz;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    d1f[s] = bld8(rS, l16, sw + fro(FR_D1, fr_idx(ti, pp, s)));
                    const bf16x8 mm = bld8(rS, l16, sw + fro(FR_GX2, fr_idx(ti, pp, s)));
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
                        dx = mma(row_pi(Gt, ti, fO, s, l), pack(dW2Tt[0], s), dx);
                        dx = mma(row_pi(Gt, ti, fX, s, l), pack(dW2Tt[1], s), dx);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dx[r] *= -etaR[r];          // -eta A2
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        dx = mma(row_pi(Bt, ti, fO, s, l), bld8(rS, l16, sw + fro(FR_W2T, fr_idx(pp, pp, s))), dx);
                        dx = mma(row_pi(Bt, ti, fX, s, l), bld8(rS, l16, sw + fro(FR_W2T, fr_idx(1 - pp, pp, s))), dx);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) dz[r] += dx[r] * (float)d1f[r >> 3][r & 7];     // dZ1
                }
                db1v += tile_colsum(dz);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 zf = pack(dz, s);                        // dZ1 (k = t rows, j = n lane)
                    bst8(rS, l16, sw + fro(FR_DZ1, fr_idx(ti, pp, s)), zf);
                    dW1t[0] = mma(tr_pi(Kt, 32 * ti, s, 0, l), zf, dW1t[0]);
                    dW1t[1] = mma(tr_pi(Kt, 32 * ti, s, 32, l), zf, dW1t[1]);
                    const bf16x8 gO = tr_pi(Gt, 32 * ti, s, fO, l), gX = tr_pi(Gt, 32 * ti, s, fX, l);
                    dW2t[0] = mma(uf[s], gO, dW2t[0]);
                    dW2t[1] = mma(uf[s], gX, dW2t[1]);
                    dW2Tt[0] = mma(gO, uf[s], dW2Tt[0]);
                    dW2Tt[1] = mma(gX, uf[s], dW2Tt[1]);
                    const bf16x8 xf = bld8(rS, l16, sw + fro(FR_X2, fr_idx(ti, pp, s)));     // X2 (m = n lane, k = t rows)
                    const bf16x8 zO = tr_pi(Bt, 32 * ti, s, fO, l), zX = tr_pi(Bt, 32 * ti, s, fX, l);
                    dW2t[0] = mma(xf, zO, dW2t[0]);
                    dW2t[1] = mma(xf, zX, dW2t[1]);
                    dW2Tt[0] = mma(zO, xf, dW2Tt[0]);
                    dW2Tt[1] = mma(zX, xf, dW2Tt[1]);
                    if (w == 0) {
                        f32x16 acc = mma(ONES, zO, zero16());
                        db2v += acc[0];
                    }
                }
            }
        }
        TTT_STAMP3(6)
        __syncthreads();                   // Bd: every read of K_i, gZ2_i, dZ2_i, eta_i is done
        TTT_STAMP3(7)

        // ================= S4b : (output path of step j,) publish, park ====================================================
        if (more) {
            add_output_path(i - 1);
            publish_state(i - 1);
            park_kg(st);
        }
        TTT_STAMP3(8)
        __syncthreads();                   // Be
        TTT_STAMP3(9)
    }

    if (tid == 0 && p.helpers) __hip_atomic_store(p.prog + bh, -(1 << 30), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // release the helpers
    // ---- hand the state gradient to the next chunk, or emit the final results ---------------------------------------------
    {
        const int l = tid & 63, h = l >> 5, c = l & 31;
        float* o1 = p.last ? p.dW1 + (size_t)bh * 64 * 256 : carry + C_DW1;
        float* o2 = p.last ? p.dW2 + (size_t)bh * 256 * 64 : carry + C_DW2;
        float* ob1 = p.last ? p.db1 + (size_t)bh * 256 : carry + C_DB1;
        float* ob2 = p.last ? p.db2 + (size_t)bh * 64 : carry + C_DB2;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = row_of(r, h);
            o1[(size_t)ro * 256 + nO + c] = dW1t[0][r];
            o1[(size_t)(32 + ro) * 256 + nO + c] = dW1t[1][r];
            o2[(size_t)(nO + ro) * 64 + fO + c] = dW2t[0][r];
            o2[(size_t)(nO + ro) * 64 + fX + c] = dW2t[1][r];
        }
        if (h == 0) ob1[nO + c] = db1v;
        if (w == 0 && h == 0) ob2[fO + c] = db2v;
        if (!p.last) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                f32x4 a = {dgam[4 * q], dgam[4 * q + 1], dgam[4 * q + 2], dgam[4 * q + 3]};
                f32x4 b = {dbet[4 * q], dbet[4 * q + 1], dbet[4 * q + 2], dbet[4 * q + 3]};
                *reinterpret_cast<f32x4*>(carry + C_DG + (size_t)tid * 8 + 4 * q) = a;
                *reinterpret_cast<f32x4*>(carry + C_DBT + (size_t)tid * 8 + 4 * q) = b;
            }
        } else {
            // dgamma / dbeta: thread (token ot, octet o) holds features 8 o .. 8 o + 7: reduce over the 64 tokens
            float* sg = red;                 // [512][8]
            float* sb = red + NT2 * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) { sg[tid * 8 + k] = dgam[k]; sb[tid * 8 + k] = dbet[k]; }
            __syncthreads();
            if (tid < 64) {
                const int o = tid >> 3, k = tid & 7;
                float a = 0.f, b = 0.f;
                for (int t = 0; t < 64; ++t) { a += sg[(t * 8 + o) * 8 + k]; b += sb[(t * 8 + o) * 8 + k]; }
                p.dlnw[(size_t)bh * 64 + tid] = a;
                p.dlnb[(size_t)bh * 64 + tid] = b;
            }
        }
    }
}

// =========================================================================================================================
// Tail kernel: one workgroup (4 waves, wave w <-> hidden slice H_w as in the slot images) per (b, h, step of the chunk):
//   dK = -eta (gZ1 dW1'^T) + dZ1 W1^T - dV        dQ = dOut + dZ1b W1'^T      (W1' = state entering the next step)
struct TailParams {
    const __bf16 *dOut, *eta, *dXV;
    char* slots; size_t slot_stride_bh;
    __bf16 *dXQ, *dXK;
    int NC, chunk_lo, chunk_n;
};
constexpr int LDS_TAIL = 4 * 64 * PS * 4;

__global__ __launch_bounds__(NT) void mlp_bwd_tail_kernel(TailParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, h = l >> 5, c = l & 31;
    const int bh = blockIdx.x / p.chunk_n, si = blockIdx.x % p.chunk_n;
    const int i = p.chunk_lo + si;
    const size_t tile = (size_t)bh * p.NC + i;
    const char* slot_w = p.slots + (size_t)bh * p.slot_stride_bh + (size_t)si * SLOT_BYTES + (size_t)w * SLOT_WAVE_FR;
    const char* next_w = slot_w + SLOT_BYTES;
    const bf16x8 I0 = ident_pi(0, h, c), I1 = ident_pi(1, h, c);
    const int ot = 16 * w + (l & 15), of0 = 16 * (l >> 4);

    for (int pass = 0; pass < 2; ++pass) {           // 0: dK, 1: dQ
        f32x16 PA[2][2];                             // [fj][ti]  partial (rows = f, lane = t) over this wave's hidden slice
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) PA[a][b] = zero16();
        if (pass == 0) {
            // -eta * (dW1'^T)^T-contraction: A = dW1'^T tile (rows = n, lane = f) in place, B = gZ1^T (k = n, j = t)
#pragma unroll
            for (int nj = 0; nj < 2; ++nj) {
                bf16x8 dWt[2][2];
#pragma unroll
                for (int fj = 0; fj < 2; ++fj) {
                    const f32x16 t = transpose_tile(ld_frag(slot_w, FR_DW1, fr_idx(fj, nj, 0), l), ld_frag(slot_w, FR_DW1, fr_idx(fj, nj, 1), l), I0, I1);
                    dWt[fj][0] = pack(t, 0);
                    dWt[fj][1] = pack(t, 1);
                }
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 gt = ld_frag(slot_w, FR_GZ1T, fr_idx(nj, ti, s), l);
                        PA[0][ti] = mma(dWt[0][s], gt, PA[0][ti]);
                        PA[1][ti] = mma(dWt[1][s], gt, PA[1][ti]);
                    }
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const float el = -(float)p.eta[tile * 64 + 32 * ti + c];
#pragma unroll
                for (int r = 0; r < 16; ++r) { PA[0][ti][r] *= el; PA[1][ti][r] *= el; }
            }
        }
        // + W^T-contraction with dZ^T:  pass 0: W1 (entering state), dZ1 ; pass 1: W1' (next slot), dZ1b
        const char* wsrc = pass == 0 ? slot_w : next_w;
        const int zarr = pass == 0 ? FR_DZ1 : FR_DZ1B;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
            bf16x8 W1T[2][2];
#pragma unroll
            for (int fj = 0; fj < 2; ++fj) {
                const f32x16 t = transpose_tile(ld_frag(wsrc, FR_W1, fr_idx(fj, nj, 0), l), ld_frag(wsrc, FR_W1, fr_idx(fj, nj, 1), l), I0, I1);
                W1T[fj][0] = pack(t, 0);
                W1T[fj][1] = pack(t, 1);
            }
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                const f32x16 zt = transpose_tile(ld_frag(slot_w, zarr, fr_idx(ti, nj, 0), l), ld_frag(slot_w, zarr, fr_idx(ti, nj, 1), l), I0, I1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 zb = pack(zt, s);
                    PA[0][ti] = mma(W1T[0][s], zb, PA[0][ti]);
                    PA[1][ti] = mma(W1T[1][s], zb, PA[1][ti]);
                }
            }
        }
        if (pass == 1) __syncthreads();              // owners of pass 0 finished reading `red`
        write_partial(red + (size_t)w * 64 * PS, PA, h, c);
        __syncthreads();
        {
            float z[16], d[16];
            gather_partial(red, nullptr, ot, of0, z);
            const size_t off = tile * 4096 + (size_t)ot * 64 + of0;
            if (pass == 0) {
                load16_bf16(p.dXV + off, d);
#pragma unroll
                for (int j = 0; j < 16; ++j) z[j] -= d[j];                // dK -= dt, dt = dV
                store16_bf16(p.dXK + off, z);
            } else {
                load16_bf16(p.dOut + off, d);
#pragma unroll
                for (int j = 0; j < 16; ++j) z[j] += d[j];
                store16_bf16(p.dXQ + off, z);
            }
        }
    }
}

}  // namespace b2

// ---------------------------------------------------------------------------------------------------------------------------
// The sweep of one (b,h) on a cluster of four workgroups (ttt_mfma_bwd3.hip): whenever the four-fold grid fits the chip.
static int g_cluster = -1;            // -1 = automatic (default), 0 = never (single-workgroup sweep below)
void set_debug_cluster(int v) { g_cluster = v; }
static bool use_cluster(int nbh) { return g_cluster != 0 && nbh * 4 <= 256; }
static size_t align128(size_t v) { return (v + 127) & ~(size_t)127; }

size_t workspace_bytes_v2(const ttt_dims* d) {
    const size_t nbh = (size_t)d->B * d->NH;
    const size_t slots = (size_t)groups_per_chunk(d) * d->G + 1;
    const size_t nbuf = get_debug_overlap() ? 2 : 1;      // a second slot buffer only when recompute and sweep overlap
    // slot buffer(s) + carry + progress words + (cluster form) exchange records and flag lines
    return nbh * (nbuf * slots * SLOT_BYTES + b2::CARRY_FLOATS2 * sizeof(float)) + align128(nbh * 64) +
           nbh * (b2::XCH_BH_BYTES + 4 * b2::FLAG_STRIDE * sizeof(unsigned));
}

// Side stream for the group recompute of the NEXT chunk: it needs only the forward checkpoints, so it runs beside the
// sweep of the current chunk on the ~110 CUs the 48 scans and their prefetch helpers leave idle (two slot buffers).  Fork /
// join with events, so for the caller everything is ordered on `stream`; lower priority than the caller's stream so the
// sweep (the critical path) gets its CUs first.  Handles are created once per process (one process per GPU).
struct SideStream {
    hipStream_t s2 = nullptr;
    hipEvent_t in = nullptr, rec[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    bool ok = false;
    SideStream() {}
    explicit SideStream(int) {
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        ok = hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo) == hipSuccess;
        ok = ok && hipEventCreateWithFlags(&in, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 2 && ok; ++i)
            ok = hipEventCreateWithFlags(&rec[i], hipEventDisableTiming) == hipSuccess &&
                 hipEventCreateWithFlags(&done[i], hipEventDisableTiming) == hipSuccess;
    }
};
static SideStream& side_stream() { static SideStream ss(1); return ss; }

void mlp_backward_v2(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, hipStream_t s) {
    const int nbh = d->B * d->NH, G = d->G, NC = d->NC;
    const int K = (NC + G - 1) / G;
    const int gpc = groups_per_chunk(d);
    const size_t slot_stride = ((size_t)gpc * G + 1) * SLOT_BYTES;
    char* slots0 = (char*)ws;                                   // two slot buffers, used alternately by the chunks
    const size_t buf_bytes = (size_t)nbh * slot_stride;
    float* carry = (float*)(slots0 + (get_debug_overlap() ? 2 : 1) * buf_bytes);
    int* prog = (int*)(carry + (size_t)nbh * b2::CARRY_FLOATS2);
    char* xch = (char*)prog + align128((size_t)nbh * 64);
    unsigned* flags = (unsigned*)(xch + (size_t)nbh * b2::XCH_BH_BYTES);
    const size_t flag_bytes = (size_t)nbh * 4 * b2::FLAG_STRIDE * sizeof(unsigned);
    char* slots = slots0;
    const bool cluster = use_cluster(nbh);
    // prefetch helpers (single-workgroup form): only when they can share the scans' XCDs (nbh % 8 == 0) and everything is co-resident
    int helpers = get_debug_helpers();
    if (helpers < 0) helpers = (nbh % 8 == 0 && nbh * 3 <= 256) ? 2 : 0;      // measured: 2 helpers 9.70 ms, 4 helpers 9.97 ms, none 11.68 ms (3 s geometry)
    if (cluster || nbh % 8 != 0 || nbh * (1 + helpers) > 256) helpers = 0;

    ScanParams sp = {};
    sp.XQ = (const __bf16*)a->XQ; sp.XK = (const __bf16*)a->XK; sp.XV = (const __bf16*)a->XV; sp.eta = (const __bf16*)a->last_eta;
    sp.ln_w = a->ttt_norm_weight; sp.ln_b = a->ttt_norm_bias;
    sp.W1c = const_cast<float*>(a->W1_checkpoints); sp.b1c = const_cast<float*>(a->b1_checkpoints);
    sp.W2c = const_cast<float*>(a->W2_checkpoints); sp.b2c = const_cast<float*>(a->b2_checkpoints);
    sp.NH = d->NH; sp.NC = NC; sp.G = G; sp.K = K; sp.eps = d->eps;
    sp.slots = slots; sp.slot_stride_bh = slot_stride; sp.slot_v2 = 1;

    b2::SweepParams2 bp = {};
    bp.XQ = (const __bf16*)a->XQ; bp.XK = (const __bf16*)a->XK; bp.dOut = (const __bf16*)a->grad_L_XQW; bp.eta = (const __bf16*)a->last_eta;
    bp.ln_w = a->ttt_norm_weight;
    bp.uW1 = a->grad_L_W1_last; bp.ub1 = a->grad_L_b1_last; bp.uW2 = a->grad_L_W2_last; bp.ub2 = a->grad_L_b2_last;
    bp.slots = slots; bp.slot_stride_bh = slot_stride; bp.carry = carry;
    bp.dXV = (__bf16*)a->grad_L_XV; bp.deta = (__bf16*)a->grad_L_last_eta;
    bp.dW1 = a->grad_L_W1_init; bp.db1 = a->grad_L_b1_init; bp.dW2 = a->grad_L_W2_init; bp.db2 = a->grad_L_b2_init;
    bp.dlnw = a->grad_L_ttt_norm_weight; bp.dlnb = a->grad_L_ttt_norm_bias;
    bp.NH = d->NH; bp.NC = NC;

    b2::TailParams tp = {};
    tp.dOut = (const __bf16*)a->grad_L_XQW; tp.eta = (const __bf16*)a->last_eta; tp.dXV = (const __bf16*)a->grad_L_XV;
    tp.slots = slots; tp.slot_stride_bh = slot_stride;
    tp.dXQ = (__bf16*)a->grad_L_XQ; tp.dXK = (__bf16*)a->grad_L_XK; tp.NC = NC;

    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)b2::mlp_bwd_sweep8_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, b2::LDS_SWEEP);
        (void)hipFuncSetAttribute((const void*)b2::mlp_bwd_sweep8_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, b2::LDS_SWEEP);
        (void)hipFuncSetAttribute((const void*)b2::mlp_bwd_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, b2::LDS_TAIL);
        attr = true;
    }
    const int nchunks = (K + gpc - 1) / gpc;
    static SideStream no_side;                                   // handles are only created when the overlap is requested
    SideStream& ss = get_debug_overlap() != 0 ? side_stream() : no_side;
    const bool overlap = get_debug_overlap() != 0 && ss.ok && nchunks > 1;
    auto chunk_range = [&](int ch, int& g0, int& ng) { g0 = ch * gpc; ng = (K - g0 < gpc) ? K - g0 : gpc; };
    auto recompute = [&](int ch, hipStream_t st) {
        int g0, ng;
        chunk_range(ch, g0, ng);
        sp.slots = slots0 + (overlap ? (size_t)(ch & 1) * buf_bytes : 0);
        sp.chunk_group0 = g0; sp.chunk_groups = ng; sp.chunk_lo = g0 * G;
        launch_group_recompute(sp, nbh, st);
    };
    if (overlap) {
        (void)hipEventRecord(ss.in, s);                           // fork: the side stream sees the caller's inputs
        (void)hipStreamWaitEvent(ss.s2, ss.in, 0);
        recompute(nchunks - 1, ss.s2);
        (void)hipEventRecord(ss.rec[(nchunks - 1) & 1], ss.s2);
    }
    for (int ch = nchunks - 1; ch >= 0; --ch) {
        int g0, ng;
        chunk_range(ch, g0, ng);
        const int buf = ch & 1;
        slots = slots0 + (overlap ? (size_t)buf * buf_bytes : 0);
        if (overlap) (void)hipStreamWaitEvent(s, ss.rec[buf], 0);
        else recompute(ch, s);
        bp.slots = slots; tp.slots = slots;
        bp.chunk_lo = g0 * G;
        bp.chunk_hi = ((g0 + ng) * G < NC) ? (g0 + ng) * G : NC;
        bp.first = (ch == nchunks - 1);
        bp.last = (ch == 0);
        bp.dbg = get_debug_timing();
        bp.prog = prog; bp.nbh = nbh; bp.helpers = helpers; bp.lead = get_debug_lead();
        bp.xch = xch; bp.flags = flags;
        if (cluster) {
            (void)hipMemsetAsync(flags, 0, flag_bytes, s);        // hand-over flags restart at 0 for every launch (memset node)
            launch_sweep_cluster(bp, nbh, s);
        } else {
            const dim3 grid(nbh * (1 + helpers)), blk(b2::NT2);
            if (bp.dbg) hipLaunchKernelGGL((b2::mlp_bwd_sweep8_kernel<true>), grid, blk, b2::LDS_SWEEP, s, bp);
            else hipLaunchKernelGGL((b2::mlp_bwd_sweep8_kernel<false>), grid, blk, b2::LDS_SWEEP, s, bp);
        }
        if (overlap && ch > 0) {
            // next chunk's recompute goes to the other buffer, free once the sweep + tail of chunk ch + 1 are done; enqueued
            // AFTER this chunk's sweep so that the sweep's workgroups are dispatched first
            if (ch + 1 <= nchunks - 1) (void)hipStreamWaitEvent(ss.s2, ss.done[buf ^ 1], 0);
            recompute(ch - 1, ss.s2);
            (void)hipEventRecord(ss.rec[buf ^ 1], ss.s2);
        }
        tp.chunk_lo = bp.chunk_lo; tp.chunk_n = bp.chunk_hi - bp.chunk_lo;
        hipLaunchKernelGGL(b2::mlp_bwd_tail_kernel, dim3(nbh * tp.chunk_n), dim3(NT), b2::LDS_TAIL, s, tp);
        if (overlap) (void)hipEventRecord(ss.done[buf], s);
    }
    // join: every side-stream launch was consumed by a wait on `s` above (its last event is rec[0 or 1] of chunk 0)
}

}  // namespace mfma
}  // namespace ttt
