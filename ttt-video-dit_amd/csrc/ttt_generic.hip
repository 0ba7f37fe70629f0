// Generic TTT-MLP / TTT-Linear scan kernels for gfx950: fp32 arithmetic on the vector ALUs,
// any CS in {16,32,64}, F a multiple of 16 (<= 64), act dtype bf16 or fp32.
//
// Role: (1) the always-correct GPU path for geometries the MFMA kernels do not cover (eval
// configs use CS=16, configs/eval/ttt-mlp/*.toml:9; fp32 activations of the TritonLinear
// contract), (2) the on-device high-accuracy reference the MFMA kernels are validated against
// at full size.  One workgroup (256 threads) per (batch, head) - the scan is sequential over NC
// (reference grid (B,NH): linear_triton.py:96).  All matrices live in an fp32 per-workgroup
// workspace in global memory (L2 resident, ~1.3 MB per workgroup); GEMMs are 4x4
// register-tiled FMA loops.  The arithmetic follows SURVEY.md Appendix A (primal form), which
// tests/test_oracle_golden.py pins to the reference's ops path.
#include "ttt_common.h"
#include "ttt_generic.h"

namespace ttt {
namespace generic {

constexpr int NT = 256;  // threads per workgroup

// ------------------------------------------------------------------------------------------
// cooperative GEMM: for all (i,j): epi(i, j, sum_k a(i,k) * b(k,j)); 4x4 micro-tiles
template <class FA, class FB, class FE>
__device__ __forceinline__ void mm(int M, int N, int K, FA a, FB b, FE epi) {
    const int tn = N >> 2;
    const int tiles = (M >> 2) * tn;
    for (int t = threadIdx.x; t < tiles; t += NT) {
        const int i0 = (t / tn) << 2, j0 = (t % tn) << 2;
        float acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
        for (int k = 0; k < K; ++k) {
            float av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) av[i] = a(i0 + i, k);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[j] = b(k, j0 + j);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], bv[j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) epi(i0 + i, j0 + j, acc[i][j]);
    }
}

// sum over the TPR consecutive lanes that share a row (TPR is a power of two <= 64)
__device__ __forceinline__ float row_sum(float v, int TPR) {
    for (int o = TPR >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// column sums over rows: out[j] (op)= sum_t f(t, j)
template <class FV, class FO>
__device__ __forceinline__ void col_sum(int rows, int cols, FV f, FO out) {
    for (int j = threadIdx.x; j < cols; j += NT) {
        float s = 0.f;
        for (int t = 0; t < rows; ++t) s += f(t, j);
        out(j, s);
    }
}

struct Dims {
    int B, NH, NC, CS, F, H, G, K;
    float eps;
};

// per-workgroup fp32 workspace carve
struct Ws {
    // state ping-pong / scratch state
    float *W1a, *b1a, *W2a, *b2a, *W1b, *b1b, *W2b, *b2b;
    // step intermediates
    float *Q, *K, *T, *eta;
    float *Z1, *X2, *D1, *gX2, *gZ1, *Z1b, *X2b;  // [CS*H]
    float *xh, *go, *gxh, *gZ2, *xhl;             // [CS*F]
    float *std_, *stdl;                           // [CS]
    // backward temporaries
    float *dOut, *dZ2b, *tF1, *tF2, *tF3, *tF4;   // [CS*F]
    float *tH1, *tH2, *tH3, *tH4;                 // [CS*H]
    float *dW1, *db1, *dW2, *db2;                 // state gradient accumulators
    float *dgam, *dbet, *deta;                    // [F],[F],[CS]
};

__host__ __device__ inline size_t ws_floats(int CS, int F, int H, bool mlp) {
    size_t st = (size_t)F * H + H + (mlp ? ((size_t)H * F + F) : 0);
    size_t n = 0;
    n += 2 * st;                        // ping-pong state
    n += 3 * (size_t)CS * F + CS;       // Q K T eta
    n += 7 * (size_t)CS * H;            // Z1 X2 D1 gX2 gZ1 Z1b X2b
    n += 5 * (size_t)CS * F + 2 * CS;   // xh go gxh gZ2 xhl std stdl
    n += 6 * (size_t)CS * F;            // dOut dZ2b tF1..4
    n += 4 * (size_t)CS * H;            // tH1..4
    n += st;                            // dW1 db1 dW2 db2
    n += 2 * (size_t)F + CS;            // dgam dbet deta
    return (n + 63) & ~(size_t)63;
}

__device__ inline Ws carve(float* p, int CS, int F, int H, bool mlp) {
    Ws w;
    auto take = [&](size_t n) { float* r = p; p += n; return r; };
    size_t FH = (size_t)F * H, HF = mlp ? (size_t)H * F : 0, Fb2 = mlp ? F : 0;
    w.W1a = take(FH); w.b1a = take(H); w.W2a = take(HF); w.b2a = take(Fb2);
    w.W1b = take(FH); w.b1b = take(H); w.W2b = take(HF); w.b2b = take(Fb2);
    w.Q = take((size_t)CS * F); w.K = take((size_t)CS * F); w.T = take((size_t)CS * F); w.eta = take(CS);
    w.Z1 = take((size_t)CS * H); w.X2 = take((size_t)CS * H); w.D1 = take((size_t)CS * H);
    w.gX2 = take((size_t)CS * H); w.gZ1 = take((size_t)CS * H); w.Z1b = take((size_t)CS * H); w.X2b = take((size_t)CS * H);
    w.xh = take((size_t)CS * F); w.go = take((size_t)CS * F); w.gxh = take((size_t)CS * F);
    w.gZ2 = take((size_t)CS * F); w.xhl = take((size_t)CS * F);
    w.std_ = take(CS); w.stdl = take(CS);
    w.dOut = take((size_t)CS * F); w.dZ2b = take((size_t)CS * F);
    w.tF1 = take((size_t)CS * F); w.tF2 = take((size_t)CS * F); w.tF3 = take((size_t)CS * F); w.tF4 = take((size_t)CS * F);
    w.tH1 = take((size_t)CS * H); w.tH2 = take((size_t)CS * H); w.tH3 = take((size_t)CS * H); w.tH4 = take((size_t)CS * H);
    w.dW1 = take(FH); w.db1 = take(H); w.dW2 = take(HF); w.db2 = take(Fb2);
    w.dgam = take(F); w.dbet = take(F); w.deta = take(CS);
    return w;
}

__device__ __forceinline__ void copyf(float* dst, const float* src, size_t n) {
    for (size_t i = threadIdx.x; i < n; i += NT) dst[i] = src[i];
}

// LN statistics + fused-L2 backward for one tile.  z holds raw Z (CS x F) on entry and x_hat on
// exit.  go/gxh/gout may be null when only LN forward is wanted.
// mode 0: fused l2 bwd (writes xh,std,go,gxh,gout=gZ)    (ops/utils.py:21-48)
// mode 1: LN forward, y = q + gam*xh + bet               (ops/utils.py:4-18, ops/ttt_mlp.py:54-56)
template <typename TA>
__device__ __forceinline__ void ln_rows(const Dims& d, int mode, float* z, float* stdv, const float* gam, const float* bet,
                                        const float* tgt, float* go, float* gxh, float* gout, const float* q, TA* out) {
    const int TPR = NT / d.CS, row = threadIdx.x / TPR, sub = threadIdx.x % TPR, F = d.F;
    float* zr = z + (size_t)row * F;
    float s = 0.f;
    for (int j = sub; j < F; j += TPR) s += zr[j];
    const float mu = row_sum(s, TPR) / F;
    float v = 0.f;
    for (int j = sub; j < F; j += TPR) { float c = zr[j] - mu; v += c * c; }
    const float sd = sqrtf(row_sum(v, TPR) / F + d.eps);
    const float r = 1.0f / sd;
    if (sub == 0) stdv[row] = sd;
    if (mode == 1) {
        for (int j = sub; j < F; j += TPR) {
            float xh = (zr[j] - mu) * r;
            zr[j] = xh;
            Act<TA>::st(out, (size_t)row * F + j, q[(size_t)row * F + j] + gam[j] * xh + bet[j]);
        }
        return;
    }
    float s1 = 0.f, s2 = 0.f;
    for (int j = sub; j < F; j += TPR) {
        float xh = (zr[j] - mu) * r;
        float g = gam[j] * xh + bet[j] - tgt[(size_t)row * F + j];
        float gx = g * gam[j];
        zr[j] = xh; go[(size_t)row * F + j] = g; gxh[(size_t)row * F + j] = gx;
        s1 += gx; s2 += gx * xh;
    }
    s1 = row_sum(s1, TPR); s2 = row_sum(s2, TPR);
    for (int j = sub; j < F; j += TPR) {
        float xh = zr[j], gx = gxh[(size_t)row * F + j];
        gout[(size_t)row * F + j] = (F * gx - s1 - xh * s2) * r / F;
    }
}

// load one mini-batch of Q,K,V,eta into fp32 workspace (T = V - K)
template <typename TA>
__device__ __forceinline__ void load_tile(const Dims& d, const Ws& w, const TA* XQ, const TA* XK, const TA* XV, const TA* eta, size_t tile) {
    const size_t n = (size_t)d.CS * d.F, base = tile * n;
    for (size_t i = threadIdx.x; i < n; i += NT) {
        float k = Act<TA>::ld(XK, base + i);
        w.Q[i] = Act<TA>::ld(XQ, base + i);
        w.K[i] = k;
        w.T[i] = Act<TA>::ld(XV, base + i) - k;
    }
    for (int i = threadIdx.x; i < d.CS; i += NT) w.eta[i] = Act<TA>::ld(eta, tile * d.CS + i);
}

// ------------------------------------------------------------------------------------------
// One primal step.  State in: (W1,b1,W2,b2) ; out: (W1n,...).  With
// need_out the second half (Z1b, X2b, xhl, stdl, output tile) is produced too.
template <bool MLP, typename TA>
__device__ void step_forward(const Dims& d, const Ws& w, const float* W1, const float* b1, const float* W2, const float* b2,
                             float* W1n, float* b1n, float* W2n, float* b2n, const float* gam, const float* bet,
                             bool need_out, TA* out_tile) {
    const int CS = d.CS, F = d.F, H = d.H;
    if (MLP) {
        // Z1 = K W1 + b1 ; X2 = gelu(Z1) ; D1 = gelu'(Z1)            (ops/ttt_mlp.py:28-29,37)
        mm(CS, H, F, [&](int i, int k) { return w.K[i * F + k]; }, [&](int k, int j) { return W1[k * H + j]; },
           [&](int i, int j, float a) {
               float z = a + b1[j], y, dy;
               gelu_and_grad(z, y, dy);
               w.Z1[i * H + j] = z; w.X2[i * H + j] = y; w.D1[i * H + j] = dy;
           });
        __syncthreads();
        // Z2 = X2 W2 + b2                                             (ops/ttt_mlp.py:30)
        mm(CS, F, H, [&](int i, int k) { return w.X2[i * H + k]; }, [&](int k, int j) { return W2[k * F + j]; },
           [&](int i, int j, float a) { w.xh[i * F + j] = a + b2[j]; });
        __syncthreads();
        ln_rows<TA>(d, 0, w.xh, w.std_, gam, bet, w.T, w.go, w.gxh, w.gZ2, nullptr, nullptr);
        __syncthreads();
        // gZ1 = (gZ2 W2^T) * gelu'(Z1)                                (ops/ttt_mlp.py:37)
        mm(CS, H, F, [&](int i, int k) { return w.gZ2[i * F + k]; }, [&](int k, int j) { return W2[j * F + k]; },
           [&](int i, int j, float a) { w.gX2[i * H + j] = a; w.gZ1[i * H + j] = a * w.D1[i * H + j]; });
        __syncthreads();
        // W1' = W1 - (eta K)^T gZ1 ; b1' = b1 - sum eta gZ1           (ops/ttt_mlp.py:49-50)
        mm(F, H, CS, [&](int i, int k) { return w.eta[k] * w.K[k * F + i]; }, [&](int k, int j) { return w.gZ1[k * H + j]; },
           [&](int i, int j, float a) { W1n[i * H + j] = W1[i * H + j] - a; });
        col_sum(CS, H, [&](int t, int j) { return w.eta[t] * w.gZ1[t * H + j]; }, [&](int j, float s) { b1n[j] = b1[j] - s; });
        // W2' = W2 - (eta X2)^T gZ2 ; b2'                              (ops/ttt_mlp.py:51-52)
        mm(H, F, CS, [&](int i, int k) { return w.eta[k] * w.X2[k * H + i]; }, [&](int k, int j) { return w.gZ2[k * F + j]; },
           [&](int i, int j, float a) { W2n[i * F + j] = W2[i * F + j] - a; });
        col_sum(CS, F, [&](int t, int j) { return w.eta[t] * w.gZ2[t * F + j]; }, [&](int j, float s) { b2n[j] = b2[j] - s; });
        __syncthreads();
        if (!need_out) return;
        // Z1b = Q W1' + b1' ; X2b = gelu                                (primal form of ops/ttt_mlp.py:39-42)
        mm(CS, H, F, [&](int i, int k) { return w.Q[i * F + k]; }, [&](int k, int j) { return W1n[k * H + j]; },
           [&](int i, int j, float a) { float z = a + b1n[j]; w.Z1b[i * H + j] = z; w.X2b[i * H + j] = gelu_only(z); });
        __syncthreads();
        mm(CS, F, H, [&](int i, int k) { return w.X2b[i * H + k]; }, [&](int k, int j) { return W2n[k * F + j]; },
           [&](int i, int j, float a) { w.xhl[i * F + j] = a + b2n[j]; });
        __syncthreads();
        ln_rows<TA>(d, 1, w.xhl, w.stdl, gam, bet, nullptr, nullptr, nullptr, nullptr, w.Q, out_tile);
        __syncthreads();
    } else {
        // TTT-Linear: Z1 = K W1 + b1 ; gZ1 = ln_fused_l2_bwd(Z1, V-K)      (ops/ttt_linear.py:26-32)
        mm(CS, F, F, [&](int i, int k) { return w.K[i * F + k]; }, [&](int k, int j) { return W1[k * F + j]; },
           [&](int i, int j, float a) { w.xh[i * F + j] = a + b1[j]; });
        __syncthreads();
        ln_rows<TA>(d, 0, w.xh, w.std_, gam, bet, w.T, w.go, w.gxh, w.gZ2, nullptr, nullptr);
        __syncthreads();
        mm(F, F, CS, [&](int i, int k) { return w.eta[k] * w.K[k * F + i]; }, [&](int k, int j) { return w.gZ2[k * F + j]; },
           [&](int i, int j, float a) { W1n[i * F + j] = W1[i * F + j] - a; });
        col_sum(CS, F, [&](int t, int j) { return w.eta[t] * w.gZ2[t * F + j]; }, [&](int j, float s) { b1n[j] = b1[j] - s; });
        __syncthreads();
        if (!need_out) return;
        mm(CS, F, F, [&](int i, int k) { return w.Q[i * F + k]; }, [&](int k, int j) { return W1n[k * F + j]; },
           [&](int i, int j, float a) { w.xhl[i * F + j] = a + b1n[j]; });
        __syncthreads();
        ln_rows<TA>(d, 1, w.xhl, w.stdl, gam, bet, nullptr, nullptr, nullptr, nullptr, w.Q, out_tile);
        __syncthreads();
    }
}

// LayerNorm input gradient for the output LN: dz = LNbwd(dOut*gam, xhl, stdl)
__device__ __forceinline__ void ln_bwd_rows(const Dims& d, const float* dout, const float* xh, const float* stdv,
                                            const float* gam, float* dz) {
    const int TPR = NT / d.CS, row = threadIdx.x / TPR, sub = threadIdx.x % TPR, F = d.F;
    float s1 = 0.f, s2 = 0.f;
    for (int j = sub; j < F; j += TPR) {
        float g = dout[(size_t)row * F + j] * gam[j];
        s1 += g; s2 += g * xh[(size_t)row * F + j];
    }
    s1 = row_sum(s1, TPR); s2 = row_sum(s2, TPR);
    const float r = 1.0f / stdv[row];
    for (int j = sub; j < F; j += TPR) {
        float g = dout[(size_t)row * F + j] * gam[j];
        dz[(size_t)row * F + j] = (F * g - s1 - xh[(size_t)row * F + j] * s2) * r / F;
    }
}

// backward of ln_fused_l2_bwd (structure of kernels/linear_backward.py:137-169)
// in : G (dL/d gZ) ; xh,std,go,gxh,gZ ; out: dZ (overwrites G), contrib (for dgam), dy
__device__ __forceinline__ void ln_l2_bwd_bwd_rows(const Dims& d, float* G, const float* xh, const float* stdv, const float* go,
                                                   const float* gxh, const float* gZ, const float* gam, float* contrib, float* dy) {
    const int TPR = NT / d.CS, row = threadIdx.x / TPR, sub = threadIdx.x % TPR, F = d.F;
    const size_t o = (size_t)row * F;
    const float r = 1.0f / stdv[row];
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    for (int j = sub; j < F; j += TPR) {
        float m = -G[o + j] * r;
        s1 += m; s2 += m * xh[o + j]; s3 += gxh[o + j] * xh[o + j];
    }
    s1 = row_sum(s1, TPR); s2 = row_sum(s2, TPR); s3 = row_sum(s3, TPR);
    float a1 = 0.f, a2 = 0.f;
    for (int j = sub; j < F; j += TPR) {
        float g = G[o + j], m = -g * r, x = xh[o + j];
        float dgxh = r * g + s1 / F + x * s2 / F;
        float y = gam[j] * dgxh;
        contrib[o + j] = go[o + j] * dgxh + y * x;
        dy[o + j] = y;
        float dxh = y * gam[j] + gxh[o + j] * s2 / F + s3 * m / F;
        float dstd = -dxh * x * r - g * gZ[o + j] * r;
        G[o + j] = dxh;           // stash dxh
        a1 += dxh; a2 += dstd;
    }
    a1 = row_sum(a1, TPR); a2 = row_sum(a2, TPR);
    for (int j = sub; j < F; j += TPR) G[o + j] = G[o + j] * r - a1 * r / F + a2 * xh[o + j] / F;
}

// ------------------------------------------------------------------------------------------
template <bool MLP, typename TA>
__global__ __launch_bounds__(NT) void fwd_kernel(Dims d, const TA* XQ, const TA* XK, const TA* XV, const TA* eta,
                                                 const float* ln_w, const float* ln_b, const float* W1i, const float* b1i,
                                                 const float* W2i, const float* b2i, float* W1c, float* b1c, float* W2c, float* b2c,
                                                 TA* XQW, float* wsbase, size_t ws_stride) {
    const int bh = blockIdx.x, h = bh % d.NH;
    const int CS = d.CS, F = d.F, H = d.H;
    Ws w = carve(wsbase + (size_t)bh * ws_stride, CS, F, H, MLP);
    const size_t FH = (size_t)F * H, HF = (size_t)H * F;
    const float* gam = ln_w + (size_t)h * F;
    const float* bet = ln_b + (size_t)h * F;
    float *W1 = w.W1a, *b1 = w.b1a, *W2 = w.W2a, *b2 = w.b2a, *W1n = w.W1b, *b1n = w.b1b, *W2n = w.W2b, *b2n = w.b2b;
    copyf(W1, W1i + bh * FH, FH); copyf(b1, b1i + (size_t)bh * H, H);
    if (MLP) { copyf(W2, W2i + bh * HF, HF); copyf(b2, b2i + (size_t)bh * F, F); }
    __syncthreads();
    for (int i = 0; i < d.NC; ++i) {
        if (i % d.G == 0) {
            const size_t c = (size_t)bh * d.K + i / d.G;
            copyf(W1c + c * FH, W1, FH); copyf(b1c + c * H, b1, H);
            if (MLP) { copyf(W2c + c * HF, W2, HF); copyf(b2c + c * F, b2, F); }
        }
        const size_t tile = (size_t)bh * d.NC + i;
        load_tile<TA>(d, w, XQ, XK, XV, eta, tile);
        __syncthreads();
        step_forward<MLP, TA>(d, w, W1, b1, W2, b2, W1n, b1n, W2n, b2n, gam, bet, true, XQW + tile * CS * F);
        float* t;
        t = W1; W1 = W1n; W1n = t; t = b1; b1 = b1n; b1n = t; t = W2; W2 = W2n; W2n = t; t = b2; b2 = b2n; b2n = t;
    }
}

template <bool MLP, typename TA>
__global__ __launch_bounds__(NT) void bwd_kernel(Dims d, const TA* XQ, const TA* XK, const TA* XV, const TA* eta,
                                                 const float* ln_w, const float* ln_b, const float* W1c, const float* b1c,
                                                 const float* W2c, const float* b2c, float* W1g, float* b1g, float* W2g, float* b2g,
                                                 const float* dW1l, const float* db1l, const float* dW2l, const float* db2l,
                                                 const TA* dXQW, float* dlnw, float* dlnb, float* dW1o, float* db1o, float* dW2o,
                                                 float* db2o, TA* deta_o, TA* dXQ, TA* dXK, TA* dXV, float* wsbase, size_t ws_stride) {
    const int bh = blockIdx.x, h = bh % d.NH;
    const int CS = d.CS, F = d.F, H = d.H, G = d.G;
    Ws w = carve(wsbase + (size_t)bh * ws_stride, CS, F, H, MLP);
    const size_t FH = (size_t)F * H, HF = (size_t)H * F, CF = (size_t)CS * F;
    const float* gam = ln_w + (size_t)h * F;
    const float* bet = ln_b + (size_t)h * F;
    // group state buffers (caller scratch, [B,NH,G,...])
    float* gW1 = W1g + (size_t)bh * G * FH;
    float* gb1 = b1g + (size_t)bh * G * H;
    float* gW2 = MLP ? W2g + (size_t)bh * G * HF : nullptr;
    float* gb2 = MLP ? b2g + (size_t)bh * G * F : nullptr;
    // upstream state gradient
    copyf(w.dW1, dW1l + bh * FH, FH); copyf(w.db1, db1l + (size_t)bh * H, H);
    if (MLP) { copyf(w.dW2, dW2l + bh * HF, HF); copyf(w.db2, db2l + (size_t)bh * F, F); }
    for (int j = threadIdx.x; j < F; j += NT) { w.dgam[j] = 0.f; w.dbet[j] = 0.f; }
    __syncthreads();

    for (int k = d.K - 1; k >= 0; --k) {
        const int lo = k * G, hi = min(lo + G, d.NC);
        const size_t c = (size_t)bh * d.K + k;
        copyf(gW1, W1c + c * FH, FH); copyf(gb1, b1c + c * H, H);
        if (MLP) { copyf(gW2, W2c + c * HF, HF); copyf(gb2, b2c + c * F, F); }
        __syncthreads();
        // pass 1: states entering every step of the group
        for (int i = lo; i < hi - 1; ++i) {
            const int s = i - lo;
            load_tile<TA>(d, w, XQ, XK, XV, eta, (size_t)bh * d.NC + i);
            __syncthreads();
            step_forward<MLP, TA>(d, w, gW1 + s * FH, gb1 + (size_t)s * H, MLP ? gW2 + s * HF : nullptr, MLP ? gb2 + (size_t)s * F : nullptr,
                                  gW1 + (s + 1) * FH, gb1 + (size_t)(s + 1) * H, MLP ? gW2 + (s + 1) * HF : nullptr,
                                  MLP ? gb2 + (size_t)(s + 1) * F : nullptr, gam, bet, false, (TA*)nullptr);
        }
        // pass 2: reverse sweep
        for (int i = hi - 1; i >= lo; --i) {
            const int s = i - lo;
            const size_t tile = (size_t)bh * d.NC + i;
            const float *W1 = gW1 + s * FH, *b1 = gb1 + (size_t)s * H;
            const float *W2 = MLP ? gW2 + s * HF : nullptr, *b2 = MLP ? gb2 + (size_t)s * F : nullptr;
            float *W1n = w.W1a, *b1n = w.b1a, *W2n = w.W2a, *b2n = w.b2a;
            load_tile<TA>(d, w, XQ, XK, XV, eta, tile);
            for (size_t e = threadIdx.x; e < CF; e += NT) w.dOut[e] = Act<TA>::ld(dXQW, tile * CF + e);
            __syncthreads();
            // recompute all forward intermediates of this step (the output tile goes to scratch tF4)
            step_forward<MLP, float>(d, w, W1, b1, W2, b2, W1n, b1n, W2n, b2n, gam, bet, true, w.tF4);

            // ---- out = Q + LN(Z2b): dgam/dbet, dZ2b ------------------------------------------
            col_sum(CS, F, [&](int t, int j) { return w.dOut[t * F + j] * w.xhl[t * F + j]; }, [&](int j, float v) { w.dgam[j] += v; });
            col_sum(CS, F, [&](int t, int j) { return w.dOut[t * F + j]; }, [&](int j, float v) { w.dbet[j] += v; });
            ln_bwd_rows(d, w.dOut, w.xhl, w.stdl, gam, w.dZ2b);
            __syncthreads();
            if (MLP) {
                // dW2' += X2b^T dZ2b ; db2' += sum dZ2b
                mm(H, F, CS, [&](int i2, int kk) { return w.X2b[kk * H + i2]; }, [&](int kk, int j) { return w.dZ2b[kk * F + j]; },
                   [&](int i2, int j, float a) { w.dW2[i2 * F + j] += a; });
                col_sum(CS, F, [&](int t, int j) { return w.dZ2b[t * F + j]; }, [&](int j, float v) { w.db2[j] += v; });
                // dZ1b = (dZ2b W2'^T) * gelu'(Z1b)
                mm(CS, H, F, [&](int i2, int kk) { return w.dZ2b[i2 * F + kk]; }, [&](int kk, int j) { return W2n[j * F + kk]; },
                   [&](int i2, int j, float a) { w.tH1[i2 * H + j] = a * gelu_grad(w.Z1b[i2 * H + j]); });
                __syncthreads();
            }
            const float* dZ1b = MLP ? w.tH1 : w.dZ2b;  // linear: Z1b plays the role of Z2b
            const int H1 = MLP ? H : F;                 // width of layer-1 output
            // dW1' += Q^T dZ1b ; db1' += sum dZ1b
            mm(F, H1, CS, [&](int i2, int kk) { return w.Q[kk * F + i2]; }, [&](int kk, int j) { return dZ1b[kk * H1 + j]; },
               [&](int i2, int j, float a) { w.dW1[i2 * H1 + j] += a; });
            col_sum(CS, H1, [&](int t, int j) { return dZ1b[t * H1 + j]; }, [&](int j, float v) { w.db1[j] += v; });
            // dQ = dOut + dZ1b W1'^T
            mm(CS, F, H1, [&](int i2, int kk) { return dZ1b[i2 * H1 + kk]; }, [&](int kk, int j) { return W1n[j * H1 + kk]; },
               [&](int i2, int j, float a) { Act<TA>::st(dXQ, tile * CF + i2 * F + j, w.dOut[i2 * F + j] + a); });
            __syncthreads();

            const float* gZ1p = MLP ? w.gZ1 : w.gZ2;   // gradient used in the W1 update
            // ---- state updates -------------------------------------------------------------
            if (MLP) {
                // A2 = gZ2 dW2'^T ; dX2 = -eta*A2 (tH2) ; deta -= rowsum(X2*A2)
                mm(CS, H, F, [&](int i2, int kk) { return w.gZ2[i2 * F + kk]; }, [&](int kk, int j) { return w.dW2[j * F + kk]; },
                   [&](int i2, int j, float a) { w.tH2[i2 * H + j] = a; });
                // dgZ2 = -(eta X2) dW2' - eta db2'   (tF1)
                mm(CS, F, H, [&](int i2, int kk) { return w.X2[i2 * H + kk]; }, [&](int kk, int j) { return w.dW2[kk * F + j]; },
                   [&](int i2, int j, float a) { w.tF1[i2 * F + j] = -w.eta[i2] * (a + w.db2[j]); });
            }
            // A1 = gZ1 dW1'^T  (tF2)
            mm(CS, F, H1, [&](int i2, int kk) { return gZ1p[i2 * H1 + kk]; }, [&](int kk, int j) { return w.dW1[j * H1 + kk]; },
               [&](int i2, int j, float a) { w.tF2[i2 * F + j] = a; });
            // dgZ1 = -(eta K) dW1' - eta db1'   (tH3 for MLP, tF1 for linear)
            float* dgZ1 = MLP ? w.tH3 : w.tF1;
            mm(CS, H1, F, [&](int i2, int kk) { return w.K[i2 * F + kk]; }, [&](int kk, int j) { return w.dW1[kk * H1 + j]; },
               [&](int i2, int j, float a) { dgZ1[i2 * H1 + j] = -w.eta[i2] * (a + w.db1[j]); });
            __syncthreads();
            {   // deta rows ; dX2 = -eta*A2 ; dK = -eta*A1
                const int TPR = NT / CS, row = threadIdx.x / TPR, sub = threadIdx.x % TPR;
                float s = 0.f;
                if (MLP) {
                    for (int j = sub; j < H; j += TPR) s += w.X2[row * H + j] * w.tH2[row * H + j];
                    for (int j = sub; j < F; j += TPR) s += w.gZ2[row * F + j] * w.db2[j];
                }
                for (int j = sub; j < F; j += TPR) s += w.K[row * F + j] * w.tF2[row * F + j];
                for (int j = sub; j < H1; j += TPR) s += gZ1p[row * H1 + j] * w.db1[j];
                s = row_sum(s, TPR);
                if (sub == 0) Act<TA>::st(deta_o, tile * CS + row, -s);
                const float e = w.eta[row];
                if (MLP) for (int j = sub; j < H; j += TPR) w.tH2[row * H + j] *= -e;
                for (int j = sub; j < F; j += TPR) w.tF2[row * F + j] *= -e;
            }
            __syncthreads();
            if (MLP) {
                // gZ1 = gX2 * D1 :  dZ1 (tH4) = dgZ1*gX2*gelu''(Z1) ; u (tH3) = dgZ1*D1
                for (size_t e = threadIdx.x; e < (size_t)CS * H; e += NT) {
                    float g = w.tH3[e];
                    w.tH4[e] = g * w.gX2[e] * gelu_grad2(w.Z1[e]);
                    w.tH3[e] = g * w.D1[e];
                }
                __syncthreads();
                // dgZ2 += u W2 ; dW2 += u^T gZ2
                mm(CS, F, H, [&](int i2, int kk) { return w.tH3[i2 * H + kk]; }, [&](int kk, int j) { return W2[kk * F + j]; },
                   [&](int i2, int j, float a) { w.tF1[i2 * F + j] += a; });
                mm(H, F, CS, [&](int i2, int kk) { return w.tH3[kk * H + i2]; }, [&](int kk, int j) { return w.gZ2[kk * F + j]; },
                   [&](int i2, int j, float a) { w.dW2[i2 * F + j] += a; });
                __syncthreads();
            }
            // ---- gZ = ln_fused_l2_bwd(Z, V-K): dZ (tF1 in place), contrib (tF3), dy (tF4) ------
            ln_l2_bwd_bwd_rows(d, w.tF1, w.xh, w.std_, w.go, w.gxh, w.gZ2, gam, w.tF3, w.tF4);
            __syncthreads();
            col_sum(CS, F, [&](int t, int j) { return w.tF3[t * F + j]; }, [&](int j, float v) { w.dgam[j] += v; });
            col_sum(CS, F, [&](int t, int j) { return w.tF4[t * F + j]; }, [&](int j, float v) { w.dbet[j] += v; });
            // dV = dt = -dy ; dK (tF2) -= dt
            for (size_t e = threadIdx.x; e < CF; e += NT) {
                float dyv = w.tF4[e];
                Act<TA>::st(dXV, tile * CF + e, -dyv);
                w.tF2[e] += dyv;
            }
            if (MLP) {
                // dX2 += dZ2 W2^T ; dZ1 += dX2 * D1
                mm(CS, H, F, [&](int i2, int kk) { return w.tF1[i2 * F + kk]; }, [&](int kk, int j) { return W2[j * F + kk]; },
                   [&](int i2, int j, float a) { w.tH4[i2 * H + j] += (w.tH2[i2 * H + j] + a) * w.D1[i2 * H + j]; });
                // dW2 += X2^T dZ2 ; db2 += sum dZ2
                mm(H, F, CS, [&](int i2, int kk) { return w.X2[kk * H + i2]; }, [&](int kk, int j) { return w.tF1[kk * F + j]; },
                   [&](int i2, int j, float a) { w.dW2[i2 * F + j] += a; });
                col_sum(CS, F, [&](int t, int j) { return w.tF1[t * F + j]; }, [&](int j, float v) { w.db2[j] += v; });
            }
            __syncthreads();
            const float* dZ1 = MLP ? w.tH4 : w.tF1;
            // dK += dZ1 W1^T ; dW1 += K^T dZ1 ; db1 += sum dZ1
            mm(CS, F, H1, [&](int i2, int kk) { return dZ1[i2 * H1 + kk]; }, [&](int kk, int j) { return W1[j * H1 + kk]; },
               [&](int i2, int j, float a) { Act<TA>::st(dXK, tile * CF + i2 * F + j, w.tF2[i2 * F + j] + a); });
            mm(F, H1, CS, [&](int i2, int kk) { return w.K[kk * F + i2]; }, [&](int kk, int j) { return dZ1[kk * H1 + j]; },
               [&](int i2, int j, float a) { w.dW1[i2 * H1 + j] += a; });
            col_sum(CS, H1, [&](int t, int j) { return dZ1[t * H1 + j]; }, [&](int j, float v) { w.db1[j] += v; });
            __syncthreads();
        }
    }
    copyf(dW1o + bh * FH, w.dW1, FH); copyf(db1o + (size_t)bh * H, w.db1, H);
    if (MLP) { copyf(dW2o + bh * HF, w.dW2, HF); copyf(db2o + (size_t)bh * F, w.db2, F); }
    copyf(dlnw + (size_t)bh * F, w.dgam, F); copyf(dlnb + (size_t)bh * F, w.dbet, F);
}

// ------------------------------------------------------------------------------------------
static Dims make_dims(const ttt_dims* d, bool mlp) {
    Dims x;
    x.B = d->B; x.NH = d->NH; x.NC = d->NC; x.CS = d->CS; x.F = d->F; x.H = mlp ? 4 * d->F : d->F;
    x.G = d->G; x.K = (d->NC + d->G - 1) / d->G; x.eps = d->eps;
    return x;
}

size_t workspace_bytes(const ttt_dims* d, bool mlp) {
    return (size_t)d->B * d->NH * ws_floats(d->CS, d->F, mlp ? 4 * d->F : d->F, mlp) * sizeof(float);
}

bool supports(const ttt_dims* d) {
    return (d->CS == 16 || d->CS == 32 || d->CS == 64) && d->F % 16 == 0 && d->F >= 16 && d->F <= 64;
}

template <bool MLP, typename TA>
static void launch_fwd(const ttt_dims* dd, const void* XQ, const void* XK, const void* XV, const void* eta, const float* lw,
                       const float* lb, const float* W1, const float* b1, const float* W2, const float* b2, float* W1c,
                       float* b1c, float* W2c, float* b2c, void* XQW, void* ws, hipStream_t s) {
    Dims d = make_dims(dd, MLP);
    size_t stride = ws_floats(d.CS, d.F, d.H, MLP);
    hipLaunchKernelGGL((fwd_kernel<MLP, TA>), dim3(d.B * d.NH), dim3(NT), 0, s, d, (const TA*)XQ, (const TA*)XK, (const TA*)XV,
                       (const TA*)eta, lw, lb, W1, b1, W2, b2, W1c, b1c, W2c, b2c, (TA*)XQW, (float*)ws, stride);
}

void mlp_forward(const ttt_dims* d, const ttt_mlp_fwd_args* a, void* ws, hipStream_t s) {
    if (d->act_dtype == TTT_DTYPE_BF16)
        launch_fwd<true, bf16_t>(d, a->XQ, a->XK, a->XV, a->last_eta, a->ttt_norm_weight, a->ttt_norm_bias, a->W1_init, a->b1_init,
                                 a->W2_init, a->b2_init, a->W1_checkpoints, a->b1_checkpoints, a->W2_checkpoints, a->b2_checkpoints, a->XQW, ws, s);
    else
        launch_fwd<true, float>(d, a->XQ, a->XK, a->XV, a->last_eta, a->ttt_norm_weight, a->ttt_norm_bias, a->W1_init, a->b1_init,
                                a->W2_init, a->b2_init, a->W1_checkpoints, a->b1_checkpoints, a->W2_checkpoints, a->b2_checkpoints, a->XQW, ws, s);
}

void linear_forward(const ttt_dims* d, const ttt_linear_fwd_args* a, void* ws, hipStream_t s) {
    if (d->act_dtype == TTT_DTYPE_BF16)
        launch_fwd<false, bf16_t>(d, a->XQ, a->XK, a->XV, a->last_eta, a->ttt_norm_weight, a->ttt_norm_bias, a->W1_init, a->b1_init,
                                  nullptr, nullptr, a->W1_checkpoints, a->b1_checkpoints, nullptr, nullptr, a->XQW, ws, s);
    else
        launch_fwd<false, float>(d, a->XQ, a->XK, a->XV, a->last_eta, a->ttt_norm_weight, a->ttt_norm_bias, a->W1_init, a->b1_init,
                                 nullptr, nullptr, a->W1_checkpoints, a->b1_checkpoints, nullptr, nullptr, a->XQW, ws, s);
}

template <bool MLP, typename TA, class A>
static void launch_bwd(const ttt_dims* dd, const A* a, const float* W2c, const float* b2c, float* W2g, float* b2g,
                       const float* dW2l, const float* db2l, float* dW2o, float* db2o, void* ws, hipStream_t s) {
    Dims d = make_dims(dd, MLP);
    size_t stride = ws_floats(d.CS, d.F, d.H, MLP);
    hipLaunchKernelGGL((bwd_kernel<MLP, TA>), dim3(d.B * d.NH), dim3(NT), 0, s, d, (const TA*)a->XQ, (const TA*)a->XK,
                       (const TA*)a->XV, (const TA*)a->last_eta, a->ttt_norm_weight, a->ttt_norm_bias, a->W1_checkpoints,
                       a->b1_checkpoints, W2c, b2c, a->W1_init_group, a->b1_init_group, W2g, b2g, a->grad_L_W1_last,
                       a->grad_L_b1_last, dW2l, db2l, (const TA*)a->grad_L_XQW, a->grad_L_ttt_norm_weight, a->grad_L_ttt_norm_bias,
                       a->grad_L_W1_init, a->grad_L_b1_init, dW2o, db2o, (TA*)a->grad_L_last_eta, (TA*)a->grad_L_XQ,
                       (TA*)a->grad_L_XK, (TA*)a->grad_L_XV, (float*)ws, stride);
}

void mlp_backward(const ttt_dims* d, const ttt_mlp_bwd_args* a, void* ws, hipStream_t s) {
    if (d->act_dtype == TTT_DTYPE_BF16)
        launch_bwd<true, bf16_t>(d, a, a->W2_checkpoints, a->b2_checkpoints, a->W2_init_group, a->b2_init_group, a->grad_L_W2_last,
                                 a->grad_L_b2_last, a->grad_L_W2_init, a->grad_L_b2_init, ws, s);
    else
        launch_bwd<true, float>(d, a, a->W2_checkpoints, a->b2_checkpoints, a->W2_init_group, a->b2_init_group, a->grad_L_W2_last,
                                a->grad_L_b2_last, a->grad_L_W2_init, a->grad_L_b2_init, ws, s);
}

void linear_backward(const ttt_dims* d, const ttt_linear_bwd_args* a, void* ws, hipStream_t s) {
    if (d->act_dtype == TTT_DTYPE_BF16)
        launch_bwd<false, bf16_t>(d, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, s);
    else
        launch_bwd<false, float>(d, a, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, ws, s);
}

}  // namespace generic
}  // namespace ttt
