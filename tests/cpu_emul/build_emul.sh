#!/bin/bash
# TEST-ONLY: compile the unchanged kernel sources for the host with the emulation shim.
#   libalsep_emul.so       -O2, loaded by pytest (-m "not gpu") through ctypes
#   libalsep_emul_asan.so  -O1 -fsanitize=address, run in a subprocess with the ASan runtime preloaded
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
ROOT="$(cd "$HERE/../.." && pwd)"
CXX="${ALSEP_HOST_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
SRC="$ROOT/audiolab_amd/csrc/fft.hip $ROOT/audiolab_amd/csrc/tdfnet.hip $ROOT/audiolab_amd/csrc/elementwise.hip $ROOT/audiolab_amd/csrc/vrnet.hip $ROOT/audiolab_amd/csrc/nn.hip $ROOT/audiolab_amd/csrc/nn_half.hip $ROOT/audiolab_amd/csrc/nn_conv_half.hip $ROOT/audiolab_amd/csrc/reverb.hip $ROOT/audiolab_amd/csrc/tdfnet_f16.hip $ROOT/audiolab_amd/csrc/fft_f16.hip"
# -DALSEP_EXPERIMENTS: the emulated library keeps the superseded / timing-experiment kernel variants, which the tests use as
# bit-identity cross-checks of the production kernels (the product libalsep.so is built without it)
COMMON="-DALSEP_EXPERIMENTS -std=c++17 -fPIC -shared -pthread -I$HERE -I$ROOT/audiolab_amd/csrc -Wno-unused-value -Wno-pass-failed -Wno-unknown-pragmas"
XS=""
for f in $SRC; do XS="$XS -x c++ $f"; done
$CXX $COMMON -O2 $XS -x c++ "$HERE/emul_runtime.cpp" -o "$HERE/libalsep_emul.so"
if [ "${1:-}" = "asan" ]; then
  $CXX $COMMON -O1 -g -fsanitize=address -fno-omit-frame-pointer $XS -x c++ "$HERE/emul_runtime.cpp" -o "$HERE/libalsep_emul_asan.so"
fi
echo "built: $(ls $HERE/*.so | tr '\n' ' ')"
