// Stand-alone probe: what does a rendezvous between two workgroups on DIFFERENT CUs cost on MI355X?
// (Design parameter of the planned hidden-dimension split of the TTT-MLP backward sweep over the CUs of one XCD, DESIGN.md
// section 8 item 1: two rendezvous per scan step, each carrying 8 - 16 KiB.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/probe_rendezvous.hip -o tools/_build/probe_rendezvous && tools/_build/probe_rendezvous
//
// Workgroup A writes a payload, publishes a sequence number; workgroup B polls it, reads the payload, writes its own payload
// and sequence number back; A polls.  Two protocols:
//   "fence"  : plain payload stores / loads, release / acquire at agent scope (on gfx950: buffer_wbl2 sc1 before the flag
//              store, buffer_inv sc1 after the poll - an L2 write-back and an L2 invalidate per hand-over);
//   "atomic" : the payload itself moves through relaxed agent-scope atomic stores / loads (sc1: they bypass the CU's L1 and
//              meet in L2 / memory), ordered against the relaxed flag by workgroup-scope fences only (a vmcnt(0) wait, no
//              cache maintenance) - what a same-XCD exchange needs and no more.  Reported: nanoseconds per round trip (= two one-way
// hand-overs) for partners on the same XCD (blocks b and b + 8) and on different XCDs (blocks b and b + 1), for payloads of
// 0 / 4 / 8 / 16 KiB per direction, 256 threads per workgroup, one workgroup per CU (the grid is 16 blocks).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct Channel {
    unsigned seq_ab, pad0[31];       // A -> B sequence number (own 128-byte line)
    unsigned seq_ba, pad1[31];       // B -> A
};

template <bool ATOMIC>
__device__ __forceinline__ void publish(unsigned* flag, unsigned v) {
    if (ATOMIC) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // this thread's payload stores have left the CU
    __syncthreads();                                                       // ... every thread's
    if (threadIdx.x == 0) __hip_atomic_store(flag, v, ATOMIC ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool ATOMIC>
__device__ __forceinline__ void await(unsigned* flag, unsigned v) {
    if (threadIdx.x == 0)
        while (__hip_atomic_load(flag, ATOMIC ? __ATOMIC_RELAXED : __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < v) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (ATOMIC) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
template <bool ATOMIC>
__device__ __forceinline__ void put(u4* buf, int c, u4 v) {
    if (ATOMIC) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(buf + c);
        __hip_atomic_store(q, ((unsigned long long)v[1] << 32) | v[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, ((unsigned long long)v[3] << 32) | v[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        buf[c] = v;
    }
}
template <bool ATOMIC>
__device__ __forceinline__ u4 get(u4* buf, int c) {
    if (ATOMIC) {
        unsigned long long* q = reinterpret_cast<unsigned long long*>(buf + c);
        const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return u4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
    }
    return buf[c];
}

// block_a / block_b: the two participating blocks; every other block exits at once
template <bool ATOMIC>
__global__ __launch_bounds__(256) void pingpong(Channel* ch, u4* buf_ab, u4* buf_ba, int chunks, int iters, int block_a, int block_b,
                                                unsigned long long* out, unsigned* sink) {
    const int b = blockIdx.x, t = threadIdx.x;
    if (b != block_a && b != block_b) return;
    u4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = wall_clock64();
    for (int it = 1; it <= iters; ++it) {
        if (b == block_a) {
            for (int c = t; c < chunks; c += 256) put<ATOMIC>(buf_ab, c, u4{(unsigned)it, (unsigned)c, (unsigned)t, 1u});
            publish<ATOMIC>(&ch->seq_ab, it);
            await<ATOMIC>(&ch->seq_ba, it);
            for (int c = t; c < chunks; c += 256) { const u4 v = get<ATOMIC>(buf_ba, c); acc.x += v.x; acc.y ^= v.y; }
        } else {
            await<ATOMIC>(&ch->seq_ab, it);
            for (int c = t; c < chunks; c += 256) { const u4 v = get<ATOMIC>(buf_ab, c); acc.x += v.x; acc.y ^= v.y; }
            for (int c = t; c < chunks; c += 256) put<ATOMIC>(buf_ba, c, u4{(unsigned)it, (unsigned)c, (unsigned)t, 2u});
            publish<ATOMIC>(&ch->seq_ba, it);
        }
    }
    const unsigned long long t1 = wall_clock64();
    if (b == block_a && t == 0) out[0] = t1 - t0;
    if (acc.x == 0x12345678u) sink[0] = acc.y;           // keep the loads
}

int main() {
    Channel* ch;
    u4 *ab, *ba;
    unsigned long long* out;
    unsigned* sink;
    CHECK(hipMalloc(&ch, sizeof(Channel)));
    CHECK(hipMalloc(&ab, 64 * 1024));
    CHECK(hipMalloc(&ba, 64 * 1024));
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMalloc(&sink, 4));
    int rate_khz = 100000;
    CHECK(hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0));
    const int iters = 2000;
    const int pairs[2][2] = {{0, 8}, {0, 1}};
    const char* names[2] = {"same XCD (blocks 0, 8)", "different XCDs (blocks 0, 1)"};
    for (int mode = 0; mode < 2; ++mode)
        for (int pi = 0; pi < 2; ++pi)
            for (int kib : {0, 4, 8, 16}) {
                CHECK(hipMemset(ch, 0, sizeof(Channel)));
                if (mode == 0) hipLaunchKernelGGL(pingpong<false>, dim3(16), dim3(256), 0, 0, ch, ab, ba, kib * 1024 / 16, iters, pairs[pi][0], pairs[pi][1], out, sink);
                else hipLaunchKernelGGL(pingpong<true>, dim3(16), dim3(256), 0, 0, ch, ab, ba, kib * 1024 / 16, iters, pairs[pi][0], pairs[pi][1], out, sink);
                CHECK(hipDeviceSynchronize());
                unsigned long long ticks = 0;
                CHECK(hipMemcpy(&ticks, out, 8, hipMemcpyDeviceToHost));
                printf("%-6s %-30s payload %2d KiB each way: %8.1f ns per round trip (%.1f ns one way)\n", mode ? "atomic" : "fence", names[pi], kib,
                       1e6 * (double)ticks / rate_khz / iters, 0.5e6 * (double)ticks / rate_khz / iters);
            }
    return 0;
}
