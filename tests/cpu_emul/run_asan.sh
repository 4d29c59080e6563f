#!/bin/bash
# TEST-ONLY: AddressSanitizer pass over the emulated kernels (about 10 minutes on 8 cores).
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
CXX="${ALSEP_HOST_CXX:-/opt/rocm/lib/llvm/bin/clang++}"
"$HERE/build_emul.sh" asan > /dev/null
RT="$($CXX -print-file-name=libclang_rt.asan-x86_64.so)"
cd /tmp
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 LD_PRELOAD="$RT" python "$HERE/asan_cases.py"
# the round-2 conv kernels at shapes where every class dispatches (about 4 minutes per mode)
for m in mq mny big; do
  ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 LD_PRELOAD="$RT" python "$HERE/asan_cases_conv.py" $m
done
# the fp32 model families' kernels (tiled GEMM / conv, InstanceNorm statistics, softmax with a leading dimension; about 2 minutes)
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 LD_PRELOAD="$RT" python "$HERE/asan_cases_nn.py"
# the round-4 kernels: persistent f16 GEMM, 64-key attention, split-half contractions (about 2 minutes)
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 LD_PRELOAD="$RT" python "$HERE/asan_cases_half.py"
