// Probe: semantics of ds_read_b64_tr_b16 on gfx950.  Every lane supplies its own 8-byte-aligned LDS
// address; LDS holds element index i at position i.  Prints, for a few address patterns, which LDS
// element index landed in (lane, j).   hipcc --offload-arch=gfx950 tools/probe_tr.hip -o /tmp/probe_tr
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, const int* addr) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + addr[threadIdx.x]));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d_out; int* d_addr;
    hipMalloc(&d_out, 64 * 4 * 2); hipMalloc(&d_addr, 64 * 4);
    for (int pat = 0; pat < 3; ++pat) {
        std::vector<int> a(64);
        for (int l = 0; l < 64; ++l) {
            if (pat == 0) a[l] = 4 * l;                 // lane-linear: lane l points at elements 4l..4l+3
            if (pat == 1) a[l] = 100 * l;               // widely spaced: identifies the supplying lane
            if (pat == 2) a[l] = ((l & 15) >> 2) * 72 + 4 * (l & 3) + (l >> 4) * 16;  // rows of a stride-72 tile
        }
        hipMemcpy(d_addr, a.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_out, d_addr);
        std::vector<short> o(256);
        hipMemcpy(o.data(), d_out, 512, hipMemcpyDeviceToHost);
        printf("pattern %d\n", pat);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d (addr %4d):", l, a[l]);
            for (int j = 0; j < 4; ++j) printf(" %5d", o[l * 4 + j]);
            printf("\n");
        }
    }
    return 0;
}
