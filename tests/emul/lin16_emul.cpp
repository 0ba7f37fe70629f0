// Host build of the TTT-Linear (mini-batch 16) wave-level kernel bodies on the wave emulator: TEST INFRASTRUCTURE, compiled
// on the fly by tests/test_emul_cpu.py with the host clang of the ROCm toolchain.  Every (b, h) scan is one emulated wave.
#include <cstdio>

#include "wave_emul.h"

#include "ttt_lin16_body.h"
#include "ttt_mlp16_body.h"

using namespace ttt;

extern "C" {

void emul_lin16_forward(const wv::Lin16Params* p, int n_bh) {
    for (int bh = 0; bh < n_bh; ++bh) emul::run_wave([&](emul::EmulWave& w) { lin16::forward(w, *p, bh); });
}

void emul_lin16_backward(const wv::Lin16Params* p, int n_bh) {
    for (int bh = 0; bh < n_bh; ++bh) emul::run_wave([&](emul::EmulWave& w) {
        lin16::backward(w, *p, bh);
    });
}

int emul_lin16_params_size() { return (int)sizeof(wv::Lin16Params); }

// TTT-MLP forward scan (mini-batch 16): one emulated 8-wave workgroup per (b, h)
// returns the number of LDS races the detector saw (0 expected); the first one is described in `msg`
int emul_mlp16_forward(const wv::Mlp16Params* p, int n_bh, char* msg, int msg_len) {
    int races = 0;
    for (int bh = 0; bh < n_bh; ++bh) {
        const emul::RaceReport r = emul::run_group(8, [&](emul::EmulWave& w) { mlp16::forward(w, *p, bh); });
        if (r.races && !races && msg) snprintf(msg, msg_len, "%s", r.first.c_str());
        races += r.races;
    }
    return races;
}

// a deliberately broken exchange between two waves: mode 0 = correct (barrier between write and read), 1 = the barrier is
// missing (read-after-write race), 2 = both waves write the same words in one epoch, 3 = the buffer is rewritten while the
// other wave may still be reading it (write-after-read: a missing second barrier)
int emul_race_selftest(int mode, char* msg, int msg_len) {
    const emul::RaceReport r = emul::run_group(2, [&](emul::EmulWave& w) {
        const int other = 1 - w.wave();
        w.lds_store<float>((64 * (mode == 2 ? 0 : w.wave()) + w.lane()) * 4, 1.0f);
        if (mode != 1) w.barrier();
        volatile float x = w.lds_load<float>((64 * other + w.lane()) * 4);
        (void)x;
        if (mode != 3) w.barrier();
        w.lds_store<float>((64 * w.wave() + w.lane()) * 4, 2.0f);
        w.barrier();
    });
    if (r.races && msg) snprintf(msg, msg_len, "%s", r.first.c_str());
    return r.races;
}
int emul_mlp16_params_size() { return (int)sizeof(wv::Mlp16Params); }

// self-test of the emulated MFMA shapes: D = A B for A [M x K], B [K x N] given row-major in fp32 (rounded to bf16 inside)
void emul_mfma_selftest(int shape, const float* A, const float* B, float* D) {
    emul::run_wave([&](emul::EmulWave& w) {
        const int l = w.lane();
        if (shape == 0) {            // 16x16x32
            const int g = l >> 4, i = l & 15;
            wv::bf16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[i * 32 + 8 * g + e]; b[e] = (__bf16)B[(8 * g + e) * 16 + i]; }
            wv::f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = w.mma32(a, b, c);
            for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
        } else if (shape == 1) {     // 16x16x16
            const int g = l >> 4, i = l & 15;
            wv::bf16x4 a, b;
            for (int e = 0; e < 4; ++e) { a[e] = (__bf16)A[i * 16 + 4 * g + e]; b[e] = (__bf16)B[(4 * g + e) * 16 + i]; }
            wv::f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = w.mma16(a, b, c);
            for (int r = 0; r < 4; ++r) D[(4 * g + r) * 16 + i] = c[r];
        } else {                     // 32x32x16
            const int h = l >> 5, c0 = l & 31;
            wv::bf16x8 a, b;
            for (int e = 0; e < 8; ++e) { a[e] = (__bf16)A[c0 * 16 + 8 * h + e]; b[e] = (__bf16)B[(8 * h + e) * 32 + c0]; }
            wv::f32x16 c;
            for (int r = 0; r < 16; ++r) c[r] = 0.f;
            c = w.mma3216(a, b, c);
            for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + c0] = c[r];
        }
    });
}
}
