#!/bin/bash
# STFT / iSTFT parity + stage micro-benchmark (defaults)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/fft_quick.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "stft or istft or demix or golden" 2>&1 | tail -2 | tee -a gpurun_out/fft_quick.log
timeout 300 python scripts/bench_fft.py 2>&1 | tail -4 | tee -a gpurun_out/fft_quick.log
N_FFT=4096 DIM_F=2048 timeout 300 python scripts/bench_fft.py 2>&1 | tail -4 | tee -a gpurun_out/fft_quick.log
N_FFT=7680 DIM_F=3072 timeout 300 python scripts/bench_fft.py 2>&1 | tail -4 | tee -a gpurun_out/fft_quick.log
