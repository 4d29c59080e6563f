#!/bin/bash
# timing-only ablations of the three-pass STFT kernel (bf16, n_fft 6144)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/fft_ablate.log
for a in 0 1 2 3 4; do
  echo "ALSEP_STFT_ABLATE=$a" | tee -a gpurun_out/fft_ablate.log
  ALSEP_STFT_ABLATE=$a timeout 300 python scripts/bench_fft.py 2>&1 | grep "stft n_fft=6144 bfloat16" | grep -v istft | tee -a gpurun_out/fft_ablate.log
done
