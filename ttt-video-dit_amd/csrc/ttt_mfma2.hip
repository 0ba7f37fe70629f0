// MFMA TTT-MLP forward scan for gfx950, revision 2: 8 waves (2 per SIMD), <= 256 registers per wave.
//
// Why a second revision: the 4-wave / 512-register kernel (ttt_mfma.hip) makes hipcc select AGPR-form
// MFMAs (every accumulator the VALU touches is copied with v_accvgpr_*), spills, runs one wave per SIMD
// (every LDS / MFMA latency exposed) and spends 32 of its 144 MFMAs per step on layout transposes.
// Here a workgroup is 8 waves so each wave fits the 256 architectural VGPRs (VGPR-form MFMA, no copies,
// no spills), two waves share a SIMD (one wave's MFMAs run under the other's VALU / LDS waits), and
// every orientation change goes through LDS transposed reads (ds_read_b64_tr_b16) instead of MFMAs.
//
// Work split.  Wave (w, p), w = hidden slice [64w, 64w+64) of the TTT-MLP, p = 0/1:
//   layer 1 : owns W1[:, Hp] and b1[Hp], Hp = [64w+32p, +32) (tiles rows=f, lane=n): f1, GELU, f3, f4, f6
//             are local to the wave - the hidden units are never contracted there;
//   layer 2 : owns W2[H_w, Fp], Fp = [32p, +32) (tiles rows=n, lane=f) for the two contractions over the
//             hidden units (f2, f7: partial sums over the 4 hidden slices meet in LDS, exactly 4 partials
//             per element), plus W2^T[:, Hp] (tiles rows=f, lane=n) - a second fp32 accumulator copy of
//             its W2 rows, updated by its own MFMAs, because f3 (gZ2 W2^T) contracts over f.
// Per step (primal form), B* = workgroup barriers:
//   A1  Z1 = K W1 + b1 -> X2 = gelu, D1 = gelu'            (rows=t, lane=n);  X2 -> LDS image [n][t]
//   B0
//   A2  partial Z2^T[Fp, t] = W2[H_w,Fp]^T X2[:,H_w]^T     (X2^T by transposed LDS reads) -> LDS partials
//   B1
//   P3  owners (8 lanes x 8 features per token): sum 4 partials + b2, fused LN / L2 backward,
//       Gs = -eta * gZ2 -> LDS image [t][f] (bf16)         (the row scaling commutes with f3)
//   B2
//   C   W2 += X2^T Gs  (f5; partner half of X2 from the image) ; gX2s = Gs W2^T (f3) ; gZ1s = gX2s*D1 ;
//       W1 += K^T gZ1s (f4; K^T by transposed reads) ; W2^T += Gs^T X2 ; b1, b2 ;
//       Z1b^T = W1'^T Q^T + b1' (f6) ; X2b = gelu
//   B3  (the X2 image is dead; its space becomes the X2b exchange buffer)
//       publish X2b^T fragments for the partner wave ; park next K, V, eta
//   B4
//   E   partial Z2b^T[Fp, t] = W2'[H_w,Fp]^T X2b^T          -> LDS partials
//   B5
//   P6  owners: sum partials + b2', LayerNorm, + Q -> XQW (one 16-byte store per lane)
// Q/K/V of step i+1 are fetched into registers at the top of step i.
#include "ttt_mfma.h"
#include "ttt_mfma_dev.h"
#include "ttt_mfma_int.h"
#include "once_per_device.h"

namespace ttt {
namespace mfma {
using namespace ttt::mf;

namespace v2 {

constexpr int NT2 = 512;
constexpr int L_K = 0;
constexpr int L_Q = L_K + TILE_ELEMS * 2;
constexpr int L_V = L_Q + TILE_ELEMS * 2;
constexpr int L_G = L_V + TILE_ELEMS * 2;
constexpr int L_X2 = L_G + TILE_ELEMS * 2;                    // [256][TS] bf16 image, later the X2b exchange
constexpr int X2IMG_BYTES = 256 * TS * 2;
constexpr int L_RED = L_X2 + X2IMG_BYTES;                     // [4][64][PS] fp32
constexpr int RED_BYTES = 4 * 64 * PS * 4;
constexpr int L_SMALL = L_RED + RED_BYTES;                    // eta[64], b1[256], b2[64], gamma[64], beta[64]
constexpr int LDS_V2 = L_SMALL + (64 + 256 + 64 + 64 + 64) * 4;
static_assert(LDS_V2 <= 160 * 1024, "LDS budget");
static_assert(8 * 4 * 1024 <= X2IMG_BYTES, "exchange buffer fits the X2 image");

typedef short s16x4 __attribute__((ext_vector_type(4)));

// ---- DPP reductions over the 8 lanes of an owner group ---------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float sum8(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    return v;
}

// ---- transposed LDS read: operand fragment (outer = column, contract = row) of a row-major bf16 image ----
// rows r0..r0+3 and r1..r1+3 (8 k-slots), 32 outer columns starting at col0; `img` row stride = stride elems.
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* img, int stride, int r0, int r1, int col0, int l) {
    const int i = l & 15, g1 = (l >> 4) & 1;
    const int off = (i >> 2) * stride + col0 + 16 * g1 + 4 * (i & 3);
    // NB: no per-element __builtin_bit_cast on vector elements (it reads element 0 for every index): use the
    // bf16-typed builtin and concatenate whole vectors.
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + r0 * stride + off));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + r1 * stride + off));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// same, in the pi k-slot order of an in-place C tile fragment s of the 32-row block starting at row0
__device__ __forceinline__ bf16x8 tr_frag_pi(const __bf16* img, int stride, int row0, int s, int col0, int l) {
    const int h = l >> 5;
    return tr_frag(img, stride, row0 + 16 * s + 4 * h, row0 + 16 * s + 8 + 4 * h, col0, l);
}

// ---- half-chunk swap (template parameter SW of the scan) ------------------------------------------------------------------
// With the 144-byte row stride every 8-byte access that walks ROWS at a fixed column (pi_read, st_image) is a 2-way bank
// conflict: rows r and r + 16 (reads: 32 lanes over 64 banks) or r and r + 8 (writes: 16 lanes over 32 banks) meet in the same
// banks (tools/lds_bank_model.py: 1 280 of a step's 7 576 LDS passes).  Under SW the two 8-byte units of every 16-byte chunk
// of a tile row are stored swapped in rows with  x(r) = bit 3 ^ bit 4 of r  = 1, which sends the colliding rows to
// different banks.  The price is address selection only: for the row walkers x is a lane constant folded into `h`, for the
// transposed reads it is a compile-time constant per instruction, and the 16-byte accessors (tile parking, the owners' rows)
// swap the halves of their value in registers (4 v_cndmask).  The data are the same: SW on / off give identical bits.
__device__ __forceinline__ int sw_x(int r) { return ((r >> 3) ^ (r >> 4)) & 1; }
template <bool SW>
__device__ __forceinline__ uint4 sw16(uint4 v, int x) {
    if (!SW) return v;
    return x ? uint4{v.z, v.w, v.x, v.y} : v;
}
template <bool SW>
__device__ __forceinline__ bf16x8 tr_frag_pi_sw(const __bf16* img, int stride, int row0, int s, int col0, int l) {
    if (!SW) return tr_frag_pi(img, stride, row0, s, col0, l);
    // rows row0 + 16 s + 4 h + (0 | 8) + (i >> 2), row0 a multiple of 32: bit 4 = s, bit 3 = (0 | 1)  ->  x = s for the
    // first read, s ^ 1 for the second; the lane's unit (i & 3) of its 16-column group flips its low bit when x = 1
    const int h = l >> 5, i = l & 15, g1 = (l >> 4) & 1;
    const int base = (i >> 2) * stride + col0 + 16 * g1;
    const int off0 = base + 4 * (i & 3), off1 = base + 4 * ((i & 3) ^ 1);
    const int r0 = row0 + 16 * s + 4 * h;
    typedef __attribute__((address_space(3))) bf16x4 lds_b4;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + r0 * stride + ((s & 1) ? off1 : off0)));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(img + (r0 + 8) * stride + ((s & 1) ? off0 : off1)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
template <bool SW>
__device__ __forceinline__ void load8_bf16_sw(const __bf16* p, int x, float (&o)[8]) {
    const uint4 raw = sw16<SW>(*reinterpret_cast<const uint4*>(p), x);
    const bf16x8 a = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)a[j];
}

__device__ __forceinline__ void load8_bf16(const __bf16* p, float (&o)[8]) {
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (float)a[j];
}
__device__ __forceinline__ void load8_f32(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3]; o[4] = b[0]; o[5] = b[1]; o[6] = b[2]; o[7] = b[3];
}
__device__ __forceinline__ void add8_f32(const float* p, float (&o)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    o[0] += a[0]; o[1] += a[1]; o[2] += a[2]; o[3] += a[3]; o[4] += b[0]; o[5] += b[1]; o[6] += b[2]; o[7] += b[3];
}

// write one wave's partial tile (rows = f in Fp, lane = t of tile ti) to red[w][t][f]
__device__ __forceinline__ void write_partial2(float* redw, const f32x16& P, int ti, int p, int h, int c) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        f32x4 v = {P[4 * q], P[4 * q + 1], P[4 * q + 2], P[4 * q + 3]};
        *reinterpret_cast<f32x4*>(redw + (32 * ti + c) * PS + 32 * p + 8 * q + 4 * h) = v;
    }
}

#define TTT_STAMP2(k)                                                        \
    if (DBG && p.dbg != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {    \
        const unsigned long long _t = __builtin_readcyclecounter();          \
        p.dbg[k] += _t - t_last;                                             \
        t_last = _t;                                                         \
    }

template <bool DBG, bool SW>
__global__ __launch_bounds__(NT2) void mlp_scan8_kernel(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* Kt = reinterpret_cast<__bf16*>(smem + L_K);
    __bf16* Qt = reinterpret_cast<__bf16*>(smem + L_Q);
    __bf16* Vt = reinterpret_cast<__bf16*>(smem + L_V);
    __bf16* Gs = reinterpret_cast<__bf16*>(smem + L_G);
    __bf16* X2img = reinterpret_cast<__bf16*>(smem + L_X2);
    char* exch = smem + L_X2;
    float* red = reinterpret_cast<float*>(smem + L_RED);
    float* etaL = reinterpret_cast<float*>(smem + L_SMALL);
    float* b1L = etaL + 64;
    float* b2L = b1L + 256;
    float* gamL = b2L + 64;
    float* betL = gamL + 64;

    const int tid = threadIdx.x;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, provably (scalar branches below)
    const int l = tid & 63, h = l >> 5, c = l & 31;
    const int w = wv >> 1, pp = wv & 1;                       // hidden slice, half
    // "own" / "other" halves: hidden units [nO, +32) are this wave's (Hp), [nX, +32) its partner's; output
    // features [fO, +32) are this wave's (Fp), [fX, +32) the partner's.  Index 0 = own, 1 = other everywhere.
    const int nO = 64 * w + 32 * pp, nX = 64 * w + 32 * (1 - pp);
    const int fO = 32 * pp, fX = 32 * (1 - pp);
    const int NC = p.NC, G = p.G;
    const int bh = blockIdx.x, head = bh % p.NH;

    // ---- state ------------------------------------------------------------------------------------
    f32x16 W1t[2];      // [a]  W1[f in 32a.., n in Hp]                      (rows=f, lane=n)   a absolute
    f32x16 W2t[2];      // [0] W2[n in Hp, f in Fp], [1] W2[n in partner's, f in Fp]   (rows=n, lane=f)
    f32x16 W2Tt[2];     // [0] W2[n in Hp, f in Fp]^T, [1] W2[n in Hp, f in partner's]^T (rows=f, lane=n)
    float b1v;          // b1[nO + c]
    float b2v = 0.f;    // b2[fO + c]   (kept by the waves with w == 0)
    {
        const float* W1g = p.W1 + (size_t)bh * 64 * 256;
        const float* W2g = p.W2 + (size_t)bh * 256 * 64;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = row_of(r, h);
            W1t[0][r] = W1g[(size_t)ro * 256 + nO + c];
            W1t[1][r] = W1g[(size_t)(32 + ro) * 256 + nO + c];
            W2t[0][r] = W2g[(size_t)(nO + ro) * 64 + fO + c];
            W2t[1][r] = W2g[(size_t)(nX + ro) * 64 + fO + c];
            W2Tt[0][r] = W2g[(size_t)(nO + c) * 64 + fO + ro];
            W2Tt[1][r] = W2g[(size_t)(nO + c) * 64 + fX + ro];
        }
        b1v = p.b1[(size_t)bh * 256 + nO + c];
        if (w == 0) b2v = p.b2[(size_t)bh * 64 + fO + c];
        if (tid < 64) {
            b2L[tid] = p.b2[(size_t)bh * 64 + tid];
            gamL[tid] = p.ln_w[(size_t)head * 64 + tid];
            betL[tid] = p.ln_b[(size_t)head * 64 + tid];
        }
    }
    bf16x8 ONES;
#pragma unroll
    for (int e = 0; e < 8; ++e) ONES[e] = (__bf16)1.0f;

    // owner geometry: token ot, features of0 .. of0+7
    const int ot = tid >> 3, of0 = 8 * (tid & 7);

    // ---- first tiles ------------------------------------------------------------------------------
    const size_t tile0 = (size_t)bh * (p.NCs ? p.NCs : NC);
    const int prow = tid >> 3, pcol = (tid & 7) * 8;          // one 16-byte chunk per thread per tile
    uint4 pfK, pfQ, pfV;
    unsigned short pfE = 0;      // eta row of the next step as raw bf16 bits, every wave its own copy: converted when it is parked -
                                 // a conversion at the load sits behind the K / V / Q loads issued with it and waits vmcnt(0) for all of them
    {
        const size_t off = tile0 * 4096 + (size_t)prow * 64 + pcol;
        pfK = *reinterpret_cast<const uint4*>(p.XK + off);
        pfV = *reinterpret_cast<const uint4*>(p.XV + off);
        pfE = reinterpret_cast<const unsigned short*>(p.eta)[tile0 * 64 + (tid & 63)];
        *reinterpret_cast<uint4*>(Kt + prow * TS + pcol) = sw16<SW>(pfK, sw_x(prow));
        *reinterpret_cast<uint4*>(Vt + prow * TS + pcol) = sw16<SW>(pfV, sw_x(prow));
        unsigned pfEu = pfE;
        asm volatile("" : "+v"(pfEu));     // every wave consumes its load HERE (left to the compiler the conversion sinks into the branch below,
                                          // the register stays pending in the other waves, and its pairing with b1v in A1 waits vmcnt(0) there)
        const float pfEf = __builtin_bit_cast(float, pfEu << 16);
        if (tid < 64) etaL[tid] = pfEf;
    }
    // packed operands of the entering state (re-made after every update, carried across steps)
    bf16x8 W1F[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int s = 0; s < 2; ++s) W1F[a][s] = pack(W1t[a], s);
    __syncthreads();

    unsigned long long t_last = __builtin_readcyclecounter();
    for (int i = 0; i < NC; ++i) {
        const size_t tile = tile0 + i;
        const bool more = (i + 1 < NC);
        // Per-iteration opaque copies of the lane / thread index: every LDS and global address below is a pure
        // function of them, and hipcc otherwise hoists ~80 loop-invariant address registers out of the loop,
        // spills them, and each reload's vmcnt(0) then waits for the prefetch loads in flight.  Recomputing an
        // address costs one VALU op; a scratch reload costs an HBM round trip.
        int l_op = tid & 63, tid_op = tid;
        asm volatile("" : "+v"(l_op), "+v"(tid_op));
        const int l = l_op, h = l >> 5, c = l & 31;
        const int tid = tid_op;
        const int ot = tid >> 3, of0 = 8 * (tid & 7);
        const int prow = tid >> 3, pcol = (tid & 7) * 8;
        const int hs = SW ? (h ^ sw_x(c)) : h;                // the row walkers' half selector (rows = 32 k + c): see sw_x
        const int xo = SW ? sw_x(ot) : 0;                     // the 16-byte accessors' swap (row ot == prow)

        if (i % G == 0) {   // checkpoint: state entering step i (mlp_tk.py:95-98)
            const size_t ck = (size_t)bh * p.K + p.ck0 + i / G;
            float* W1g = p.W1c + ck * 64 * 256;
            float* W2g = p.W2c + ck * 256 * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = row_of(r, h);
                W1g[(size_t)ro * 256 + nO + c] = W1t[0][r];
                W1g[(size_t)(32 + ro) * 256 + nO + c] = W1t[1][r];
                W2g[(size_t)(nO + ro) * 64 + fO + c] = W2t[0][r];
                W2g[(size_t)(nX + ro) * 64 + fO + c] = W2t[1][r];
            }
            if (h == 0) p.b1c[ck * 256 + nO + c] = b1v;
            if (w == 0 && h == 0) p.b2c[ck * 64 + fO + c] = b2v;
        }
        // Tile traffic.  Register-staged prefetch across a whole step is not affordable at 256 VGPRs (hipcc spills
        // the staged tiles at once, i.e. waits for HBM at the top of every step), so: (1) here, one dword per 128-byte
        // line of the NEXT step's K / V / Q tiles is touched to pull them into L2 (waves 0..2, one tile each; the
        // value is only "used" after B3); (2) the real 16-byte loads are issued after B3 - L2 hits by then - and
        // parked before B5 (K, V, eta) or after the next B0 (Q), live only across the low-pressure tail of the step.
        unsigned touch = 0;
        if (more && wv < 3) {
            const __bf16* src = (wv == 0 ? p.XK : wv == 1 ? p.XV : p.XQ) + (tile + 1) * 4096 + (size_t)l * 64;
            touch = *reinterpret_cast<const unsigned*>(src);
        }
        if (i == 0) pfQ = *reinterpret_cast<const uint4*>(p.XQ + tile * 4096 + (size_t)prow * 64 + pcol);

        // ================= A1: Z1 = K W1 + b1 ; X2, D1 ; X2 image ================================
        f32x16 D1[2];                 // gelu'(Z1)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti) {
            f32x16 Z = zero16();
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    Z = mma(pi_read(Kt + (32 * ti + c) * TS, 32 * a, s, hs), W1F[a][s], Z);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float y, dy;
                gelu_fwd_grad(Z[r] + b1v, y, dy);
                Z[r] = y;
                D1[ti][r] = dy;
            }
            // pin gelu' here: hipcc otherwise sinks half of its arithmetic into phase C and keeps Z1 (+ temporaries,
            // ~56 registers) alive across A2 / P3 instead of these 16
            asm volatile("" : "+v"(D1[ti]));
#pragma unroll
            for (int s = 0; s < 2; ++s) st_image(X2img + (nO + c) * TS, 32 * ti, s, hs, pack(Z, s));
        }
        TTT_STAMP2(0)
        __syncthreads();              // B0: X2 image complete; every P6 read of step i-1 (red, Qt, b2L) is done
        TTT_STAMP2(8)
        if (DBG && p.dump && blockIdx.x == 0 && i == 0)
            for (int e = tid; e < 256 * 64; e += NT2) p.dump[e] = (float)X2img[(e >> 6) * TS + ((e & 63) ^ (SW ? 4 * sw_x(e >> 6) : 0))];

        // ================= A2: partial Z2^T[Fp, t] over the hidden slice ==========================
        {
            bf16x8 W2F[2][2];         // operands of the entering W2 (packed here and again in E: not worth 16 live registers)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W2F[a][s] = pack(W2t[a], s);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 P = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    P = mma(W2F[0][s], tr_frag_pi_sw<SW>(X2img, TS, nO, s, 32 * ti, l), P);
                    P = mma(W2F[1][s], tr_frag_pi_sw<SW>(X2img, TS, nX, s, 32 * ti, l), P);
                }
                write_partial2(red + (size_t)w * 64 * PS, P, ti, pp, h, c);
            }
        }
        *reinterpret_cast<uint4*>(Qt + prow * TS + pcol) = sw16<SW>(pfQ, xo);   // Q of this step (read only after B2)
        TTT_STAMP2(1)
        __syncthreads();              // B1: partials visible
        TTT_STAMP2(9)

        // ================= P3: owners - reduce, fused LN / L2 backward -> Gs = -eta gZ2 ===========
        {
            float z[8], kk[8], vv[8];
            load8_f32(b2L + of0, z);
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) add8_f32(red + ((size_t)ww * 64 + ot) * PS + of0, z);
            if (DBG && p.dump && blockIdx.x == 0 && i == 0) {
                for (int j = 0; j < 8; ++j) p.dump[16384 + ot * 64 + of0 + j] = z[j];
                for (int ww = 0; ww < 4; ++ww)
                    for (int j = 0; j < 8; ++j) p.dump[45376 + (ww * 64 + ot) * 64 + of0 + j] = red[((size_t)ww * 64 + ot) * PS + of0 + j];
            }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += z[j];
            const float mu = sum8(s) * (1.0f / 64.0f);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = z[j] - mu; v += d * d; }
            const float rstd = __builtin_amdgcn_rsqf(sum8(v) * (1.0f / 64.0f) + p.eps);
            if (DBG && p.dump && blockIdx.x == 0 && i == 0 && (tid & 7) == 0) { p.dump[60000 + ot] = mu; p.dump[60064 + ot] = rstd; }
            load8_bf16_sw<SW>(Kt + ot * TS + of0, xo, kk);
            load8_bf16_sw<SW>(Vt + ot * TS + of0, xo, vv);
            float s1 = 0.f, s2 = 0.f, gx[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (z[j] - mu) * rstd;
                const float g = gamL[of0 + j];
                gx[j] = (g * xh + betL[of0 + j] - (vv[j] - kk[j])) * g;
                z[j] = xh;
                s1 += gx[j]; s2 += gx[j] * xh;
            }
            s1 = sum8(s1);
            s2 = sum8(s2);
            const float sc = -etaL[ot] * rstd * (1.0f / 64.0f);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)((64.0f * gx[j] - s1 - z[j] * s2) * sc);
            *reinterpret_cast<uint4*>(Gs + ot * TS + of0) = sw16<SW>(__builtin_bit_cast(uint4, o), xo);
        }
        TTT_STAMP2(2)
        __syncthreads();              // B2: Gs visible
        TTT_STAMP2(10)
        if (DBG && p.dump && blockIdx.x == 0 && i == 0)
            for (int e = tid; e < 64 * 64; e += NT2) p.dump[20480 + e] = (float)Gs[(e >> 6) * TS + ((e & 63) ^ (SW ? 4 * sw_x(e >> 6) : 0))];

        // ================= C: state updates, f3, f4 ===============================================
        {
            // operands of the ENTERING W2^T for f3, packed before the accumulator copy is updated
            bf16x8 W2TF[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W2TF[a][s] = pack(W2Tt[a], s);
            // b2' = b2 + colsum_t Gs (ones MFMA: every row of the product is the column sum); two waves only
            if (w == 0) {
                f32x16 acc = zero16();
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int s = 0; s < 2; ++s) acc = mma(ONES, tr_frag_pi_sw<SW>(Gs, TS, 32 * ti, s, fO, l), acc);
                b2v += acc[0];
            }
            // f5 + W2^T update.  Gs^T fragments (outer=f, k=t) by transposed reads: own half of f (also f5's B
            // operand), then the partner's half.  X2 (outer=n, k=t) is re-read from the image for both halves
            // (cheaper than keeping the in-place fragments of A1 alive across A2 / P3).
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 gO = tr_frag_pi_sw<SW>(Gs, TS, 32 * ti, s, fO, l);
                    const bf16x8 xO = pi_read(X2img + (nO + c) * TS, 32 * ti, s, hs);
                    W2t[0] = mma(xO, gO, W2t[0]);                                                  // f5, own hidden half
                    W2t[1] = mma(pi_read(X2img + (nX + c) * TS, 32 * ti, s, hs), gO, W2t[1]);       // f5, partner's half
                    W2Tt[0] = mma(gO, xO, W2Tt[0]);
                    W2Tt[1] = mma(tr_frag_pi_sw<SW>(Gs, TS, 32 * ti, s, fX, l), xO, W2Tt[1]);
                }
            // f3: gX2s = Gs W2^T ; gZ1s = gX2s * D1   (rows=t, lane=n) ; f4: W1[f, n in Hp] += K[:, f]^T gZ1s
            float sb = 0.f;
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 gx = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    gx = mma(pi_read(Gs + (32 * ti + c) * TS, fO, s, hs), W2TF[0][s], gx);
                    gx = mma(pi_read(Gs + (32 * ti + c) * TS, fX, s, hs), W2TF[1][s], gx);
                }
#pragma unroll
                for (int r = 0; r < 16; r += 2) {      // aligned register pairs (see gelu_fwd_tile_pk)
                    const f32x2 g = f32x2{gx[r], gx[r + 1]} * f32x2{D1[ti][r], D1[ti][r + 1]};
                    gx[r] = g[0];
                    gx[r + 1] = g[1];
                    sb += g[0];
                    sb += g[1];
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 gz = pack(gx, s);           // (outer=n, k=t)
                    W1t[0] = mma(tr_frag_pi_sw<SW>(Kt, TS, 32 * ti, s, 0, l), gz, W1t[0]);
                    W1t[1] = mma(tr_frag_pi_sw<SW>(Kt, TS, 32 * ti, s, 32, l), gz, W1t[1]);
                }
            }
            b1v += xor_add(sb, 32);   // b1' = b1 - sum_t eta gZ1
        }
        if (DBG && p.dump && blockIdx.x == 0 && i == 0) {
            if (h == 0) p.dump[24576 + nO + c] = b1v;
            if (w == 0 && h == 0) p.dump[24832 + fO + c] = b2v;
            for (int r = 0; r < 16; ++r) {
                const int ro = row_of(r, h);
                p.dump[65536 + (size_t)ro * 256 + nO + c] = W1t[0][r];
                p.dump[65536 + (size_t)(32 + ro) * 256 + nO + c] = W1t[1][r];
                p.dump[81920 + (size_t)(nO + ro) * 64 + fO + c] = W2t[0][r];
                p.dump[81920 + (size_t)(nX + ro) * 64 + fO + c] = W2t[1][r];
                p.dump[98304 + (size_t)(nO + c) * 64 + fO + ro] = W2Tt[0][r];
                p.dump[98304 + (size_t)(nO + c) * 64 + fX + ro] = W2Tt[1][r];
            }
        }
        TTT_STAMP2(3)
        // ---- f6: Z1b^T = W1'^T Q^T + b1' (rows=n, lane=t) ; X2b = gelu ---------------------------
        bf16x8 X2bF[2][2];            // [ti][s]  (outer=t, k=n in Hp)
        {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W1F[a][s] = pack(W1t[a], s);
            if (h == 0) b1L[nO + c] = b1v;                                    // this wave's private 32 floats
            const f32x16 bias = rows_from_lds(b1L + nO, 0, h);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 zb = bias;
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        zb = mma(W1F[a][s], pi_read(Qt + (32 * ti + c) * TS, 32 * a, s, hs), zb);
                gelu_fwd_tile_pk(zb);
                if (DBG && p.dump && blockIdx.x == 0 && i == 0)
                    for (int r = 0; r < 16; ++r) p.dump[28992 + (size_t)(nO + row_of(r, h)) * 64 + 32 * ti + c] = zb[r];
                X2bF[ti][0] = pack(zb, 0);
                X2bF[ti][1] = pack(zb, 1);
            }
        }
        TTT_STAMP2(4)
        __syncthreads();              // B3: every read of the X2 image, of Kt and of Vt / etaL is done
        TTT_STAMP2(11)
        asm volatile("" :: "v"(touch));
        if (more) {
            const size_t off = (tile + 1) * 4096 + (size_t)prow * 64 + pcol;
            pfK = *reinterpret_cast<const uint4*>(p.XK + off);
            pfV = *reinterpret_cast<const uint4*>(p.XV + off);
            pfQ = *reinterpret_cast<const uint4*>(p.XQ + off);
            pfE = reinterpret_cast<const unsigned short*>(p.eta)[(tile + 1) * 64 + (tid & 63)];
        }
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int s = 0; s < 2; ++s)
                *reinterpret_cast<bf16x8*>(exch + ((size_t)(wv * 4 + ti * 2 + s) * 64 + l) * 16) = X2bF[ti][s];
        if (w == 0 && h == 0) b2L[fO + c] = b2v;
        __syncthreads();              // B4: exchange visible
        TTT_STAMP2(12)

        // ================= E: partial Z2b^T[Fp, t] ================================================
        {
            bf16x8 W2F[2][2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int s = 0; s < 2; ++s) W2F[a][s] = pack(W2t[a], s);
#pragma unroll
            for (int ti = 0; ti < 2; ++ti) {
                f32x16 P = zero16();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 xo = *reinterpret_cast<const bf16x8*>(exch + ((size_t)((wv ^ 1) * 4 + ti * 2 + s) * 64 + l) * 16);
                    P = mma(W2F[0][s], X2bF[ti][s], P);
                    P = mma(W2F[1][s], xo, P);
                }
                write_partial2(red + (size_t)w * 64 * PS, P, ti, pp, h, c);
            }
        }
        if (more) {                   // next step's K, V, eta (their last readers finished before B3)
            *reinterpret_cast<uint4*>(Kt + prow * TS + pcol) = sw16<SW>(pfK, xo);
            *reinterpret_cast<uint4*>(Vt + prow * TS + pcol) = sw16<SW>(pfV, xo);
            unsigned pfEu = pfE;
            asm volatile("" : "+v"(pfEu));     // every wave consumes its load HERE (left to the compiler the conversion sinks into the branch below,
                                              // the register stays pending in the other waves, and its pairing with b1v in A1 waits vmcnt(0) there)
            const float pfEf = __builtin_bit_cast(float, pfEu << 16);
            if (tid < 64) etaL[tid] = pfEf;
        }
        TTT_STAMP2(5)
        __syncthreads();              // B5
        TTT_STAMP2(13)

        // ================= P6: owners - reduce, LayerNorm, residual -> XQW ========================
        {
            float z[8], q[8];
            load8_f32(b2L + of0, z);
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) add8_f32(red + ((size_t)ww * 64 + ot) * PS + of0, z);
            if (DBG && p.dump && blockIdx.x == 0 && i == 0)
                for (int j = 0; j < 8; ++j) p.dump[24896 + ot * 64 + of0 + j] = z[j];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) s += z[j];
            const float mu = sum8(s) * (1.0f / 64.0f);
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = z[j] - mu; v += d * d; }
            const float rstd = __builtin_amdgcn_rsqf(sum8(v) * (1.0f / 64.0f) + p.eps);
            load8_bf16_sw<SW>(Qt + ot * TS + of0, xo, q);
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (__bf16)(q[j] + gamL[of0 + j] * ((z[j] - mu) * rstd) + betL[of0 + j]);
            *reinterpret_cast<bf16x8*>(p.out + tile * 4096 + (size_t)ot * 64 + of0) = o;
        }
        TTT_STAMP2(6)
    }
    if (p.W1f) {        // the state after the last step of this launch: what the next part of the sequence starts from (fp32, exact)
        float* W1g = p.W1f + (size_t)bh * 64 * 256;
        float* W2g = p.W2f + (size_t)bh * 256 * 64;
        int l_op = threadIdx.x & 63;          // an opaque lane index of its own: addresses formed from the function-scope one would be
        asm volatile("" : "+v"(l_op));        // hoisted in front of the step loop and spilled across it (see the loop's own copies)
        const int l = l_op, h = l >> 5, c = l & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ro = row_of(r, h);
            W1g[(size_t)ro * 256 + nO + c] = W1t[0][r];
            W1g[(size_t)(32 + ro) * 256 + nO + c] = W1t[1][r];
            W2g[(size_t)(nO + ro) * 64 + fO + c] = W2t[0][r];
            W2g[(size_t)(nX + ro) * 64 + fO + c] = W2t[1][r];
        }
        if (h == 0) p.b1f[(size_t)bh * 256 + nO + c] = b1v;
        if (w == 0 && h == 0) p.b2f[(size_t)bh * 64 + fO + c] = b2v;
    }
}

static void set_attr_once() {
    static ttt::OncePerDevice done;
    done.run([&] {
        (void)hipFuncSetAttribute((const void*)mlp_scan8_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_V2);
        (void)hipFuncSetAttribute((const void*)mlp_scan8_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_V2);
    });
}

}  // namespace v2

static float* g_dump = nullptr;
void set_debug_dump(float* buf) { g_dump = buf; }
// (the half-chunk swap of the LDS tile rows - template parameter SW, sw_x above - is always on since round 5: 6.06 against 6.22 ms at
// NC = 804, 2.14 against 2.19 at NC = 282, identical bits, profiles/r4m_*)
void launch_scan_forward_v2(const ScanParams& p0, int n_bh, unsigned long long* dbg, hipStream_t s) {
    ScanParams p = p0;
    p.dbg = dbg;
    p.dump = g_dump;
    v2::set_attr_once();
    const bool dbg_build = p.dbg || p.dump;
    if (dbg_build) hipLaunchKernelGGL((v2::mlp_scan8_kernel<true, true>), dim3(n_bh), dim3(v2::NT2), v2::LDS_V2, s, p);
    else hipLaunchKernelGGL((v2::mlp_scan8_kernel<false, true>), dim3(n_bh), dim3(v2::NT2), v2::LDS_V2, s, p);
}

}  // namespace mfma
}  // namespace ttt
