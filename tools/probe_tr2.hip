// Probe 2: the tr_frag / tr_frag_pi helpers of ttt_mfma2.hip on a dynamic-LDS image with stride 72.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int TS = 72;
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
__global__ void k(short* out, int row0, int s, int col0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __bf16* img = reinterpret_cast<__bf16*>(smem + 9216);
    short* raw = reinterpret_cast<short*>(img);
    for (int i = threadIdx.x; i < 64 * TS; i += blockDim.x) raw[i] = (short)((i / TS) * 64 + (i % TS));   // row*64 + col
    __syncthreads();
    const int l = threadIdx.x & 63, h = l >> 5;
    bf16x8 v = tr_frag(img, TS, row0 + 16 * s + 4 * h, row0 + 16 * s + 8 + 4 * h, col0, l);
    if (threadIdx.x < 64)
        *reinterpret_cast<bf16x8*>(out + l * 8) = v;
}
int main() {
    short* d_out;
    (void)hipMalloc(&d_out, 64 * 8 * 2);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 100000);
    int bad = 0;
    for (int row0 = 0; row0 < 64; row0 += 32) for (int s = 0; s < 2; ++s) for (int col0 = 0; col0 < 64; col0 += 32) {
        hipLaunchKernelGGL(k, dim3(1), dim3(128), 100000, 0, d_out, row0, s, col0);
        std::vector<short> o(512);
        (void)hipMemcpy(o.data(), d_out, 1024, hipMemcpyDeviceToHost);
        for (int l = 0; l < 64; ++l) for (int e = 0; e < 8; ++e) {
            const int h = l >> 5, c = l & 31;
            const int row = row0 + 16 * s + 8 * (e >> 2) + 4 * h + (e & 3), col = col0 + c;
            if (o[l * 8 + e] != (short)(row * 64 + col)) {
                if (bad < 10) printf("row0 %d s %d col0 %d lane %d e %d: got (%d,%d) want (%d,%d)\n", row0, s, col0, l, e, o[l*8+e] / 64, o[l*8+e] % 64, row, col);
                ++bad;
            }
        }
    }
    printf("tr_frag_pi probe: %d mismatches\n", bad);
    return 0;
}
