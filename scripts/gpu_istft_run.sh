#!/bin/bash
# iSTFT: hop-blocks per workgroup (run) sweep, three-pass and generic kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/istft_run.log
for m in 1; do for r in 16 18 28; do
  echo "ALSEP_ISTFT_R16=$m ALSEP_ISTFT_RUN=$r" | tee -a gpurun_out/istft_run.log
  ALSEP_ISTFT_R16=$m ALSEP_ISTFT_RUN=$r timeout 300 python scripts/bench_fft.py 2>&1 | grep "istft" | tee -a gpurun_out/istft_run.log
done; done
