// Plain vector / parameter types shared by the wave-level kernel bodies (ttt_lin16_body.h), the device backend
// (ttt_mfma16.hip) and the host-side wave emulator used by the CPU tests (tests/emul).  No HIP dependency.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace ttt {
namespace wv {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// arguments of the TTT-Linear scans at mini-batches of 16 tokens (F = 64)
struct Lin16Params {
    const __bf16 *XQ, *XK, *XV, *eta;      // [B,NH,NC,16,64] x3, [B,NH,NC,16,1]
    const float *ln_w, *ln_b;               // [NH,64]
    const float *W1, *b1;                   // forward: initial state [B,NH,64,64], [B,NH,1,64]
    float *W1c, *b1c;                       // checkpoints [B,NH,K,64,64], [B,NH,K,1,64] (forward: written; backward: read)
    __bf16* out;                            // forward: XQW
    // backward
    const __bf16* dOut;                     // [B,NH,NC,16,64]
    const float *dW1_last, *db1_last;       // upstream gradient of the final state
    char* scratch_w;                        // [B,NH,G] x 16 KiB: per-step state as packed operands (W1_init_group storage)
    float* scratch_b;                       // [B,NH,G,64]
    float *dln_w, *dln_b;                   // [B,NH,1,64]
    float *dW1, *db1;                       // [B,NH,64,64], [B,NH,1,64]
    __bf16 *deta, *dXQ, *dXK, *dXV;
    int NH, NC, G, K;
    float eps;
};

// arguments of the TTT-MLP forward scan at mini-batches of 16 tokens (F = 64, hidden 256)
struct Mlp16Params {
    const __bf16 *XQ, *XK, *XV, *eta;
    const float *ln_w, *ln_b;               // [NH,64] (also addressed as [1,NH,1,64])
    const float *W1, *b1, *W2, *b2;         // initial state [B,NH,64,256], [B,NH,1,256], [B,NH,256,64], [B,NH,1,64]
    float *W1c, *b1c, *W2c, *b2c;           // checkpoints [B,NH,K,...]
    __bf16* out;
    int NH, NC, G, K;
    float eps;
};

}  // namespace wv
}  // namespace ttt
