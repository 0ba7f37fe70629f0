// Host build of the revision-4 backward's deriver-wave body (csrc/ttt_bwd4_aux_body.h) on the wave emulator: TEST
// INFRASTRUCTURE, compiled on the fly by tests/test_emul_bwd4_cpu.py with the host clang of the ROCm toolchain.
#include <cstdio>
#include <cstring>

#include "wave_emul.h"

#include "ttt_bwd4_aux_body.h"

using namespace ttt;

namespace {
constexpr int TILE_B = 64 * bwd4::TS * 2;
constexpr int L_K = 0, L_G = TILE_B, L_ETA = 2 * TILE_B, L_R1 = 20 * 1024, L_R2 = L_R1 + 24 * 1024, L_R3 = L_R2 + 8 * 1024,
              L_R4 = L_R3 + 24 * 1024, L_END = L_R4 + 24 * 1024;
}

extern "C" {

// One reverse step of one hidden slice (64 units) on two emulated deriver waves (pp = 0, 1).
//   W2 [64 n][64 f]  fp32: state AFTER the step (in) -> state ENTERING the step (out); W1 [64 f][64 n] is not touched
//   z1, z1b: the slice's A_Z1 / A_Z1B fragment arrays (8 fragments x 64 lanes x 8 bf16, as raw 16-bit words)
//   K, G [64 t][64] bf16 (raw words), eta [64] fp32
//   lds_out: the LDS image after the step (R1 | R2 | R3 = D1B, X2B, W2T | R4 = D1, M, X2), L_END - L_R1 bytes
//   gslice: 8 KiB: gZ1 (T) fragments for the tail kernel
// returns the number of LDS races the detector saw
int emul_bwd4_aux_step(float* W1, float* W2, const unsigned short* z1, const unsigned short* z1b, const unsigned short* K,
                       const unsigned short* G, const float* eta, char* lds_out, char* gslice, char* msg, int msg_len) {
    (void)W1;                                  // (rounds 3 - 5: the deriver reversed W1 too; since round 6 the tail kernel does)
    static char park[2 * bwd4::PARK_BYTES];
    const emul::RaceReport rep = emul::run_group(2, [&](emul::EmulWave& w) {
        const int pp = w.wave(), l = w.lane(), h = l >> 5, c = l & 31;
        // stage the tiles (wave 0 only, then a barrier)
        if (pp == 0) {
            for (int t = l; t < 64; t += 64) {
                for (int f = 0; f < 64; ++f) {
                    *reinterpret_cast<unsigned short*>(w.lds_ptr(L_K + (t * bwd4::TS + f) * 2)) = K[t * 64 + f];
                    *reinterpret_cast<unsigned short*>(w.lds_ptr(L_G + (t * bwd4::TS + f) * 2)) = G[t * 64 + f];
                }
                *reinterpret_cast<float*>(w.lds_ptr(L_ETA + t * 4)) = eta[t];
            }
        }
        w.barrier();
        bwd4::AuxState st;
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2) + 4 * h;
            for (int a = 0; a < 2; ++a) {
                st.W2t[a][r] = W2[(32 * pp + ro) * 64 + 32 * a + c];
            }
        }
        bwd4::Frags4 Z1, Z1B;
        for (int ti = 0; ti < 2; ++ti)
            for (int s = 0; s < 2; ++s) {
                std::memcpy(&Z1.f[ti][s], z1 + ((size_t)bwd4::fr_idx(ti, pp, s) * 64 + l) * 8, 16);
                std::memcpy(&Z1B.f[ti][s], z1b + ((size_t)bwd4::fr_idx(ti, pp, s) * 64 + l) * 8, 16);
            }
        bwd4::reverse_step(w, st, pp, L_K, L_G, L_ETA, Z1, L_R1, L_R2, gslice, 0, park + pp * bwd4::PARK_BYTES);
        bwd4::stage_r4(w, pp, L_R4, park + pp * bwd4::PARK_BYTES);
        bwd4::derive_z1b(w, Z1B, pp, L_R3, L_R3 + 8 * 1024);
        bwd4::stage_w2t(w, st, pp, L_R3 + 16 * 1024);
        for (int r = 0; r < 16; ++r) {
            const int ro = (r & 3) + 8 * (r >> 2) + 4 * h;
            for (int a = 0; a < 2; ++a) {
                W2[(32 * pp + ro) * 64 + 32 * a + c] = st.W2t[a][r];
            }
        }
        w.barrier();
        if (pp == 0 && l == 0) std::memcpy(lds_out, w.lds_ptr(L_R1), L_END - L_R1);
    });
    if (rep.races && msg) snprintf(msg, msg_len, "%s", rep.first.c_str());
    return rep.races;
}
int emul_bwd4_lds_bytes() { return L_END - L_R1; }
}
