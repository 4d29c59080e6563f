#!/bin/bash
# stage micro-benchmark only (three-pass kernels on)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python scripts/bench_fft.py 2>&1 | tail -4 | tee gpurun_out/fft_quick.log
