// Host build of the TTT-Linear (mini-batch 16) wave-level kernel bodies on the wave emulator: TEST INFRASTRUCTURE, compiled
// on the fly by tests/test_emul_cpu.py with the host clang of the ROCm toolchain.  Every (b, h) scan is one emulated wave.
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
        if (p->lds_slots > 0) lin16::backward<true>(w, *p, bh);
        else lin16::backward<false>(w, *p, bh);
    });
}

int emul_lin16_params_size() { return (int)sizeof(wv::Lin16Params); }

// TTT-MLP forward scan (mini-batch 16): one emulated 8-wave workgroup per (b, h)
void emul_mlp16_forward(const wv::Mlp16Params* p, int n_bh) {
    for (int bh = 0; bh < n_bh; ++bh) emul::run_group(8, [&](emul::EmulWave& w) { mlp16::forward(w, *p, bh); });
}
int emul_mlp16_params_size() { return (int)sizeof(wv::Mlp16Params); }
}
