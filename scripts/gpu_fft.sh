#!/bin/bash
# STFT / iSTFT kernels: parity tests, then the stage micro-benchmark with the three-pass kernels on and off
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stft or istft or demix or golden" 2>&1 | tail -5 | tee gpurun_out/fft_pytest.log
: > gpurun_out/fft_bench.log
for m in 1 0; do
  echo "ALSEP_STFT_R16=$m ALSEP_ISTFT_R16=$m" | tee -a gpurun_out/fft_bench.log
  ALSEP_STFT_R16=$m ALSEP_ISTFT_R16=$m timeout 300 python scripts/bench_fft.py 2>&1 | tail -4 | tee -a gpurun_out/fft_bench.log
done
