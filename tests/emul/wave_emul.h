// Host-side emulation of gfx950 wavefronts / workgroups for the kernel bodies of csrc/ttt_lin16_body.h, ttt_mlp16_body.h (TEST
// INFRASTRUCTURE: lets the CPU test-suite execute the very same kernel body, lane by lane, and compare it with the oracle
// when no GPU is at hand).  64 host threads play the 64 lanes; every cross-lane primitive (MFMA, transposed LDS read, DPP
// row reduction) is a rendezvous: deposit operands, barrier, compute this lane's share of the result from everybody's
// operands, barrier.  Lane / register layouts are those of the hardware instructions:
//   v_mfma_f32_16x16x32_bf16 : A lane (g,i) = A[i][8g+e], B lane (g,i) = B[8g+e][i], D lane (g,i) = D[4g+r][i]
//   v_mfma_f32_16x16x16_bf16 : A lane (g,i) = A[i][4g+e], B lane (g,i) = B[4g+e][i], D as above
//   v_mfma_f32_32x32x16_bf16 : A lane (h,c) = A[c][8h+e], B lane (h,c) = B[8h+e][c], D lane (h,c) = D[(r&3)+8(r>>2)+4h][c], r < 16
//   ds_read_b64_tr_b16       : within a 16-lane group, lane i receives element (i & 3) of the 8-byte chunks addressed by
//                              lanes 4e + (i >> 2), e = 0..3   (probe: tools/probe_tr.hip)
#pragma once
#include <barrier>
#include <cmath>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "ttt_wave_types.h"

namespace ttt {
namespace emul {
using namespace ttt::wv;

struct WaveShared {                                    // rendezvous state of one wave
    std::barrier<> bar{64};
    bf16x8 a8[64], b8[64];
    bf16x4 a4[64], b4[64];
    float f[64];
    int addr[64];                                      // LDS bank model: this instruction's byte address per lane
};

// LDS bank model (MI355X_MICROARCH.md, "LDS: lane groups and banks"): a wave-wide LDS instruction is served in lane GROUPS; within
// a group, every extra DISTINCT dword address on a busy bank costs one more pass of the LDS array (identical addresses
// broadcast).  passes = sum over groups of max over banks of (distinct dwords on that bank); conflict-free = one pass per group.
//   ds_read_b32 / b64, ds_read_b64_tr_b16 : 2 groups of 32 lanes {0-31}, {32-63}; 32 banks for b32, 64 for the others
//   ds_read_b128                          : 4 groups of 16 lanes {0-3,12-15,20-27}, {4-11,16-19,28-31}, +32 for the upper half; 64 banks
//   ds_write_b32                          : 2 x 32, 32 banks;  ds_write_b64 : 4 x 16 contiguous, 32 banks;  ds_write_b128 : 8 x 8 contiguous, 32 banks
struct BankCount {
    long passes = 0, conflicts = 0, instructions = 0;
};
inline void bank_cost(const int* addr, int bytes, bool write, BankCount& out) {
    static const int G128[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27},
                                    {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                    {32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59},
                                    {36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63}};
    int ngroups, glen, nbanks;
    if (write) { nbanks = 32; ngroups = bytes >= 16 ? 8 : bytes == 8 ? 4 : 2; glen = 64 / ngroups; }
    else if (bytes >= 16) { nbanks = 64; ngroups = 4; glen = 16; }
    else { nbanks = bytes == 4 ? 32 : 64; ngroups = 2; glen = 32; }
    const int dwords = (bytes + 3) / 4;
    for (int g = 0; g < ngroups; ++g) {
        int seen[64][64], n[64];                     // distinct dword addresses per bank
        for (int b = 0; b < nbanks; ++b) n[b] = 0;
        for (int k = 0; k < glen; ++k) {
            const int lane = (!write && bytes >= 16) ? G128[g][k] : g * glen + k;
            for (int d = 0; d < dwords; ++d) {
                const int dw = addr[lane] / 4 + d, b = dw % nbanks;
                bool dup = false;
                for (int j = 0; j < n[b]; ++j) dup = dup || seen[b][j] == dw;
                if (!dup) seen[b][n[b]++] = dw;
            }
        }
        int worst = 1;
        for (int b = 0; b < nbanks; ++b) worst = n[b] > worst ? n[b] : worst;
        out.passes += worst;
        out.conflicts += worst - 1;
    }
    out.instructions += 1;
}

// LDS race detector: every 4-byte word remembers its last tracked write and read as (barrier epoch, wave).  Two accesses
// of DIFFERENT waves to a word within the SAME epoch (no workgroup barrier between them), at least one of them a write,
// are a race on the hardware however the host threads happened to interleave here.  Accesses of one wave are ordered
// (LDS is in order within a wave).  Tracked: lds_load / lds_store / tr_read; untracked: lds<T>() references and lds_ptr().
struct Shadow {
    int w_epoch = -1, w_wave = -1, r_epoch = -1, r_wave = -1;      // r_wave = -2: several waves read it in r_epoch
};
struct GroupShared {                                   // one workgroup: LDS + a barrier over all of its threads
    alignas(16) char lds[160 * 1024];
    std::barrier<> bar;
    std::vector<std::unique_ptr<WaveShared>> waves;
    std::vector<Shadow> shadow;
    std::mutex shadow_mu[256];
    std::mutex report_mu;
    std::string first_race;
    int races = 0;
    bool count_banks = false;                          // LDS bank model on (every lane of a wave must then take part in each LDS access)
    BankCount rd, wr, tr;                              // b128 / b64 / b32 reads, stores, transposed reads (guarded by report_mu)
    explicit GroupShared(int n_waves) : bar(64 * n_waves), shadow(sizeof(lds) / 4) {
        for (int w = 0; w < n_waves; ++w) waves.emplace_back(new WaveShared());
    }
    void report(const char* what, int word, int epoch, int wave, int other) {
        std::lock_guard<std::mutex> g(report_mu);
        if (races++ == 0)
            first_race = std::string(what) + " at LDS byte " + std::to_string(4 * word) + ", epoch " + std::to_string(epoch) + ": wave " +
                         std::to_string(wave) + " vs wave " + std::to_string(other);
    }
    void track(int byte_off, int bytes, bool write, int epoch, int wave) {
        for (int word = byte_off / 4; word <= (byte_off + bytes - 1) / 4; ++word) {
            std::lock_guard<std::mutex> g(shadow_mu[word & 255]);
            Shadow& s = shadow[word];
            if (s.w_epoch == epoch && s.w_wave != wave) report(write ? "write after write" : "read after write", word, epoch, wave, s.w_wave);
            if (write) {
                if (s.r_epoch == epoch && s.r_wave != wave) report("write after read", word, epoch, wave, s.r_wave);
                s.w_epoch = epoch; s.w_wave = wave;
            } else {
                if (s.r_epoch == epoch) { if (s.r_wave != wave) s.r_wave = -2; }
                else { s.r_epoch = epoch; s.r_wave = wave; }
            }
        }
    }
};

struct EmulWave {
    GroupShared* grp;
    WaveShared* sh;
    int l, w;
    int epoch = 0;                                     // workgroup barriers passed so far

    int lane() const { return l; }
    int wave() const { return w; }
    int thread() const { return 64 * w + l; }
    int opaque(int v) const { return v; }
    bf16x8 opaque8(bf16x8 v) const { return v; }
    void store_stream(char* p, bf16x8 v) const { *reinterpret_cast<bf16x8*>(p) = v; }
    void stamp(int) const {}                    // (device: DEBUG cycle stamps inside a body)
    static constexpr bool kPrio = true;         // (attention bodies: the priority calls are no-ops here)
    void setprio(int) const {}
    void sync() { sh->bar.arrive_and_wait(); }
    void barrier() { grp->bar.arrive_and_wait(); ++epoch; }     // __syncthreads()
    void lds_fence() { sync(); }                       // same-wave LDS write -> read ordering point (free on the device)
    template <class T> T& lds(int byte_off) { return *reinterpret_cast<T*>(grp->lds + byte_off); }
    char* lds_ptr(int byte_off) { return grp->lds + byte_off; }
    void bank_account(int byte_off, int bytes, int kind) {      // kind 0 read, 1 write, 2 transposed read
        if (!grp->count_banks) return;
        sh->addr[l] = byte_off;
        sync();
        if (l == 0) {
            BankCount c;
            bank_cost(sh->addr, bytes, kind == 1, c);
            std::lock_guard<std::mutex> g(grp->report_mu);
            BankCount& t = kind == 0 ? grp->rd : kind == 1 ? grp->wr : grp->tr;
            t.passes += c.passes; t.conflicts += c.conflicts; t.instructions += c.instructions;
        }
        sync();
    }
    template <class T> T lds_load(int byte_off) {      // tracked by the race detector
        bank_account(byte_off, (int)sizeof(T), 0);
        grp->track(byte_off, (int)sizeof(T), false, epoch, w);
        return *reinterpret_cast<const T*>(grp->lds + byte_off);
    }
    template <class T> void lds_store(int byte_off, T v) {
        bank_account(byte_off, (int)sizeof(T), 1);
        grp->track(byte_off, (int)sizeof(T), true, epoch, w);
        *reinterpret_cast<T*>(grp->lds + byte_off) = v;
    }
    // element-pointer view of LDS (bf16 elements) used by the attention bodies (csrc/attn_body.h): device = __bf16*
    struct tile_t {
        int e;
        tile_t operator+(int k) const { return tile_t{e + k}; }
    };
    tile_t lds_base() const { return tile_t{0}; }
    template <class T> T ld(tile_t p) { return lds_load<T>(2 * p.e); }
    template <class T> void st(tile_t p, T v) { lds_store<T>(2 * p.e, v); }
    bf16x4 tr(tile_t p) { return tr_read(2 * p.e); }
    float log(float x) const { return std::log(x); }
    float rsq(float x) const { return 1.0f / std::sqrt(x); }
    float exp2(float x) const { return std::exp2(x); }
    float rcp(float x) const { return 1.0f / x; }

    f32x4 mma32(bf16x8 a, bf16x8 b, f32x4 c) {
        sh->a8[l] = a; sh->b8[l] = b;
        sync();
        const int g = l >> 4, i = l & 15;
        for (int r = 0; r < 4; ++r) {
            float acc = 0.f;
            for (int gk = 0; gk < 4; ++gk)
                for (int e = 0; e < 8; ++e) acc += (float)sh->a8[16 * gk + 4 * g + r][e] * (float)sh->b8[16 * gk + i][e];
            c[r] += acc;
        }
        sync();
        return c;
    }
    f32x4 mma16(bf16x4 a, bf16x4 b, f32x4 c) {
        sh->a4[l] = a; sh->b4[l] = b;
        sync();
        const int g = l >> 4, i = l & 15;
        for (int r = 0; r < 4; ++r) {
            float acc = 0.f;
            for (int gk = 0; gk < 4; ++gk)
                for (int e = 0; e < 4; ++e) acc += (float)sh->a4[16 * gk + 4 * g + r][e] * (float)sh->b4[16 * gk + i][e];
            c[r] += acc;
        }
        sync();
        return c;
    }
    // v_mfma_f32_32x32x16_bf16 (the CS = 64 kernels' shape; lane (h, c) = (l >> 5, l & 31))
    f32x16 mma3216(bf16x8 a, bf16x8 b, f32x16 c) {
        sh->a8[l] = a; sh->b8[l] = b;
        sync();
        const int h = l >> 5, col = l & 31;
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            float acc = 0.f;
            for (int hk = 0; hk < 2; ++hk)
                for (int e = 0; e < 8; ++e) acc += (float)sh->a8[32 * hk + row][e] * (float)sh->b8[32 * hk + col][e];
            c[r] += acc;
        }
        sync();
        return c;
    }
    float sum8(float v) {                                // sum over the aligned group of 8 lanes (quad_perm x2 + row_half_mirror)
        sh->f[l] = v;
        sync();
        float s = 0.f;
        for (int j = 0; j < 8; ++j) s += sh->f[(l & ~7) + j];
        sync();
        return s;
    }
    // ds_read_b64_tr_b16 with this lane's byte address into LDS
    bf16x4 tr_read(int byte_addr) {
        bank_account(byte_addr, 8, 2);
        sync();                                          // earlier LDS writes of every lane of the wave have landed
        grp->track(byte_addr, 8, false, epoch, w);
        sh->a4[l] = *reinterpret_cast<const bf16x4*>(grp->lds + byte_addr);
        sync();
        const int base = l & ~15, i = l & 15;
        bf16x4 r;
        for (int e = 0; e < 4; ++e) r[e] = sh->a4[base + 4 * e + (i >> 2)][i & 3];
        sync();
        return r;
    }
    float sum16(float v) {                               // sum over the 16 lanes of this lane's DPP row
        sh->f[l] = v;
        sync();
        float s = 0.f;
        for (int j = 0; j < 16; ++j) s += sh->f[(l & ~15) + j];
        sync();
        return s;
    }
    float xor_read(float v, int mask) {                  // v of lane (l ^ mask)   (__shfl_xor)
        sh->f[l] = v;
        sync();
        const float s = sh->f[l ^ mask];
        sync();
        return s;
    }
    bool any(bool b) {                                   // wave-wide OR (ballot != 0)
        sh->f[l] = b ? 1.0f : 0.0f;
        sync();
        bool r = false;
        for (int j = 0; j < 64; ++j) r = r || sh->f[j] != 0.0f;
        sync();
        return r;
    }
    float xor_add(float v, int mask) {                   // v + v of lane (l ^ mask)
        sh->f[l] = v;
        sync();
        const float s = v + sh->f[l ^ mask];
        sync();
        return s;
    }
};

// run `body(EmulWave&)` on the 64 * n_waves threads of one workgroup
struct RaceReport {
    int races = 0;
    std::string first;
    BankCount rd, wr, tr;                                // LDS bank model totals (when switched on)
};
template <class F>
RaceReport run_group(int n_waves, F body, bool count_banks = false) {
    GroupShared* grp = new GroupShared(n_waves);
    grp->count_banks = count_banks;
    std::vector<std::thread> th;
    for (int w = 0; w < n_waves; ++w)
        for (int l = 0; l < 64; ++l)
            th.emplace_back([grp, w, l, &body] {
                EmulWave bk{grp, grp->waves[w].get(), l, w};
                body(bk);
            });
    for (auto& t : th) t.join();
    RaceReport r{grp->races, grp->first_race, grp->rd, grp->wr, grp->tr};
    delete grp;
    return r;
}
template <class F>
void run_wave(F body) { run_group(1, body); }

}  // namespace emul
}  // namespace ttt
