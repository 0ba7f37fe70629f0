// Host build of the segment-attention workgroup bodies (csrc/attn_body.h) on the wave emulator: TEST INFRASTRUCTURE, compiled on
// the fly by tests/test_emul_cpu.py with the host clang of the ROCm toolchain.  One emulated 8-wave workgroup per block of
// 256 query rows, the same (head, block) decomposition as the device launch.
#include <cstdio>

#include "wave_emul.h"

#include "attn_body.h"

using namespace ttt;

extern "C" {

// returns the number of LDS races the detector saw (0 expected); the first one is described in `msg`
int emul_attn_forward(const attn::FwdParams* p, char* msg, int msg_len) {
    const int nqb = (p->S + attnb::QB - 1) / attnb::QB, nbh = p->B * p->NH;
    int races = 0;
    for (int b = 0; b < nbh * nqb; ++b) {
        int bh, qb;
        attnb::head_of_block(b, nqb, nbh, bh, qb);
        const emul::RaceReport r = emul::run_group(8, [&](emul::EmulWave& w) { attnb::forward(w, *p, bh, qb); });
        if (r.races && !races && msg) snprintf(msg, msg_len, "%s", r.first.c_str());
        races += r.races;
    }
    return races;
}

// nsub = key tiles of 64 per LDS stage (1 = the shipped form, 2 = the opt-in form with half the barriers)
int emul_attn_dq_n(const attn::BwdParams* p, int nsub, char* msg, int msg_len) {
    const int nqb = (p->S + attnb::QB - 1) / attnb::QB, nbh = p->B * p->NH;
    int races = 0;
    for (int b = 0; b < nbh * nqb; ++b) {
        int bh, qb;
        attnb::head_of_block(b, nqb, nbh, bh, qb);
        const emul::RaceReport r = emul::run_group(8, [&](emul::EmulWave& w) {
            if (nsub == 2) attnb::dq_staged<2>(w, *p, bh, qb);
            else if (nsub == -1) attnb::dq_staged<1, true>(w, *p, bh, qb);       // (negative: XOR-swizzled tiles)
            else if (nsub == -2) attnb::dq_staged<2, true>(w, *p, bh, qb);
            else attnb::dq(w, *p, bh, qb);
        });
        if (r.races && !races && msg) snprintf(msg, msg_len, "%s", r.first.c_str());
        races += r.races;
    }
    return races;
}

int emul_attn_dq(const attn::BwdParams* p, char* msg, int msg_len) { return emul_attn_dq_n(p, 1, msg, msg_len); }

// dq_wide: NQ = 2 blocks of 32 query rows per wave (a workgroup covers 512 rows), nsub tiles of 64 keys per LDS stage
int emul_attn_dq_wide(const attn::BwdParams* p, int nsub, char* msg, int msg_len) {
    const int nqb = (p->S + 2 * attnb::QB - 1) / (2 * attnb::QB), nbh = p->B * p->NH;
    int races = 0;
    for (int b = 0; b < nbh * nqb; ++b) {
        int bh, qb;
        attnb::head_of_block(b, nqb, nbh, bh, qb);
        const emul::RaceReport r = emul::run_group(8, [&](emul::EmulWave& w) {
            if (nsub == 2) attnb::dq_wide<2, 2>(w, *p, bh, qb);
            else attnb::dq_wide<1, 2>(w, *p, bh, qb);
        });
        if (r.races && !races && msg) snprintf(msg, msg_len, "%s", r.first.c_str());
        races += r.races;
    }
    return races;
}

// variant 2 = <8 waves, revision 1's arithmetic>, 3 = <8, accumulator-initialised row scalars>, 4 = <12, ...> (attn.h)
int emul_attn_dkdv_n(const attn::BwdParams* p, int variant, int nsub, char* msg, int msg_len) {
    const int nw = variant == 4 ? 12 : 8;
    const int nkb = (p->S + 32 * nw - 1) / (32 * nw), nbh = p->B * p->NH;
    int races = 0;
    for (int b = 0; b < nbh * nkb; ++b) {
        int bh, kvb;
        attnb::head_of_block(b, nkb, nbh, bh, kvb);
        const emul::RaceReport r = emul::run_group(nw, [&](emul::EmulWave& w) {
            if (variant == 2) attnb::dkdv<8, false>(w, *p, bh, kvb);
            else if (variant == 3) attnb::dkdv<8, true>(w, *p, bh, kvb);
            else if (nsub == 2) attnb::dkdv_staged<12, true, 2>(w, *p, bh, kvb);
            else if (nsub == 3) attnb::dkdv_staged<12, true, 3>(w, *p, bh, kvb);
            else if (nsub == 4) attnb::dkdv_staged<12, true, 4>(w, *p, bh, kvb);
            else if (nsub == -1) attnb::dkdv_staged<12, true, 1, true>(w, *p, bh, kvb);
            else if (nsub == -2) attnb::dkdv_staged<12, true, 2, true>(w, *p, bh, kvb);
            else attnb::dkdv<12, true>(w, *p, bh, kvb);
        });
        if (r.races && !races && msg) snprintf(msg, msg_len, "%s", r.first.c_str());
        races += r.races;
    }
    return races;
}

int emul_attn_dkdv(const attn::BwdParams* p, int variant, char* msg, int msg_len) { return emul_attn_dkdv_n(p, variant, 1, msg, msg_len); }

// LDS bank model (wave_emul.h bank_cost) over the first workgroup of the dQ (kernel 0) or 12-wave dK / dV (kernel 1) body:
// out[0..8] = {passes, conflict passes, instructions} of the plain reads (b128 row fragments, row scalars), the stores, and the
// transposed reads.  Returns the number of LDS races (0).
int emul_attn_bank_model(const attn::BwdParams* p, int kernel, long* out) {
    emul::RaceReport r;
    if (kernel == 0) r = emul::run_group(8, [&](emul::EmulWave& w) { attnb::dq(w, *p, 0, 0); }, true);
    else if (kernel == 1) r = emul::run_group(12, [&](emul::EmulWave& w) { attnb::dkdv<12, true>(w, *p, 0, 0); }, true);
    else if (kernel == 2) r = emul::run_group(8, [&](emul::EmulWave& w) { attnb::dq_staged<2, true>(w, *p, 0, 0); }, true);      // swizzled tiles
    else r = emul::run_group(12, [&](emul::EmulWave& w) { attnb::dkdv_staged<12, true, 2, true>(w, *p, 0, 0); }, true);
    const emul::BankCount* c[3] = {&r.rd, &r.wr, &r.tr};
    for (int k = 0; k < 3; ++k) { out[3 * k] = c[k]->passes; out[3 * k + 1] = c[k]->conflicts; out[3 * k + 2] = c[k]->instructions; }
    return r.races;
}

int emul_attn_fwd_params_size() { return (int)sizeof(attn::FwdParams); }
int emul_attn_bwd_params_size() { return (int)sizeof(attn::BwdParams); }
}
