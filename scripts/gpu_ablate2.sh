#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
: > gpurun_out/ablate2.log
for a in 0 2 4 6 3; do
  rm -rf gpurun_out/prof_a
  ALSEP_CONV_ABLATE=$a timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_a -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  echo "== ablate $a" | tee -a gpurun_out/ablate2.log
  python scripts/trace_summary.py "$(find gpurun_out/prof_a -name '*kernel_trace.csv' | head -1)" 40 | grep -E "conv3x3" | tee -a gpurun_out/ablate2.log
done
rm -rf gpurun_out/prof_a
