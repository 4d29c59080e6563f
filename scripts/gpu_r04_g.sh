#!/bin/bash
# round 4, call g: split contraction of the generic float32 kernels (A/B per family) + the whole GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python3 scripts/bench_contraction.py > gpurun_out/r04_contraction_ab.txt 2>&1; echo "ab rc $?"; grep -v amdgpu.ids gpurun_out/r04_contraction_ab.txt | tail -14
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04_pytest_gpu_full.txt 2>&1; echo "pytest rc $?"; tail -24 gpurun_out/r04_pytest_gpu_full.txt
