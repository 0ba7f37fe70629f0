// Probe: issue cost of v_exp_f32 / v_rcp_f32 / v_fma_f32 / v_pk_fma_f32 on gfx950, one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int N = 256;   // instructions per chain pass (8 independent registers x 32 rounds)
template <int KIND>
__global__ void probe(float* out, unsigned long long* cyc) {
    float v[8];
    f32x2 w[8];
    for (int k = 0; k < 8; ++k) { v[k] = 0.001f * (threadIdx.x + k + 1); w[k] = f32x2{v[k], v[k] + 1.f}; }
    const float c = out[0];
    const f32x2 c2 = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int r = 0; r < 32; ++r) {
#pragma unroll
        for (int rep = 0; rep < N / 8 / 4; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (KIND == 0) v[k] = __builtin_amdgcn_exp2f(v[k]);
                if (KIND == 1) v[k] = __builtin_amdgcn_rcpf(v[k]);
                if (KIND == 2) v[k] = __builtin_fmaf(v[k], c, v[k]);
                if (KIND == 3) w[k] = __builtin_elementwise_fma(w[k], c2, w[k]);
                if (KIND == 4) { v[k] = __builtin_amdgcn_exp2f(v[k]); w[k] = __builtin_elementwise_fma(w[k], c2, w[k]); }   // 1 trans + 1 pk
                if (KIND == 5) { v[k] = __builtin_amdgcn_exp2f(v[k]); v[(k + 4) & 7] = __builtin_fmaf(v[(k + 4) & 7], c, 1.0f); }
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += v[k] + w[k][0] + w[k][1];
    out[1 + blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 4 * (1 + 1024 * 512)); hipMalloc(&cyc, 8);
    hipMemset(out, 0, 4);
    const char* names[] = {"v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_pk_fma_f32", "exp + pk_fma", "exp + fma"};
    for (int waves = 4; waves <= 8; waves += 4)          // 4 waves per workgroup = one per SIMD, 8 = two per SIMD
        for (int kind = 0; kind < 6; ++kind) {
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) probe<0><<<1, 64 * waves>>>(out, cyc);
                if (kind == 1) probe<1><<<1, 64 * waves>>>(out, cyc);
                if (kind == 2) probe<2><<<1, 64 * waves>>>(out, cyc);
                if (kind == 3) probe<3><<<1, 64 * waves>>>(out, cyc);
                if (kind == 4) probe<4><<<1, 64 * waves>>>(out, cyc);
                if (kind == 5) probe<5><<<1, 64 * waves>>>(out, cyc);
            }
            unsigned long long h = 0; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const int n = 32 * (N / 4) ;   // instructions of the named kind per wave (kinds 4, 5: pairs)
            printf("%d waves/SIMD  %-14s %8.2f cycles per instruction%s per wave\n", waves / 4, names[kind], (double)h / n, kind >= 4 ? " pair" : "");
        }
    return 0;
}
