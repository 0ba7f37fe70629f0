#!/bin/bash
# Build libttt_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [-j]
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
OUT=../lib
mkdir -p "$OUT" build
pids=()
for src in capi ttt_generic ttt_mfma ttt_mfma2 ttt_mfma16 ttt_mfma_bwd2 ttt_mfma_bwd4 ttt_mfma_rc4 ttt_prepost attn_fwd attn_bwd attn_pre attn_v2; do
  if [ ! -f build/$src.o ] || [ $src.hip -nt build/$src.o ] || [ ttt_common.h -nt build/$src.o ] || [ ../../include/ttt_hip.h -nt build/$src.o ] || [ ttt_mfma_dev.h -nt build/$src.o ] || [ ttt_mfma_int.h -nt build/$src.o ] || [ ttt_mfma_bwd_dev.h -nt build/$src.o ] || [ ttt_bwd4_dev.h -nt build/$src.o ] || [ ttt_bwd4_aux_body.h -nt build/$src.o ] || [ ttt_prepost.h -nt build/$src.o ] || [ ttt_mfma.h -nt build/$src.o ] || [ attn.h -nt build/$src.o ] || [ attn_dev.h -nt build/$src.o ] || [ ttt_lin16_body.h -nt build/$src.o ] || [ ttt_mlp16_body.h -nt build/$src.o ] || [ ttt_wave_types.h -nt build/$src.o ] || [ attn_body.h -nt build/$src.o ] || [ attn_types.h -nt build/$src.o ] || [ once_per_device.h -nt build/$src.o ]; then
    $HIPCC $FLAGS -c $src.hip -o build/$src.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/capi.o build/ttt_generic.o build/ttt_mfma.o build/ttt_mfma2.o build/ttt_mfma16.o build/ttt_mfma_bwd2.o build/ttt_mfma_bwd4.o build/ttt_mfma_rc4.o build/ttt_prepost.o build/attn_fwd.o build/attn_bwd.o build/attn_pre.o build/attn_v2.o -o $OUT/libttt_hip.so
echo "built $OUT/libttt_hip.so"
