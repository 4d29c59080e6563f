#!/bin/bash
# STFT parity + stage micro-benchmark: sliding-window kernel (1) with several run lengths, per-frame kernel (2)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/fft_quick.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stft or istft or demix or golden" 2>&1 | tail -2 | tee -a gpurun_out/fft_quick.log
for r in 0 16 19 32; do
  echo "ALSEP_STFT_R16=1 ALSEP_STFT_RUN=$r" | tee -a gpurun_out/fft_quick.log
  ALSEP_STFT_RUN=$r timeout 300 python scripts/bench_fft.py 2>&1 | grep -v istft | tail -2 | tee -a gpurun_out/fft_quick.log
done
