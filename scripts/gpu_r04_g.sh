#!/bin/bash
# round 4, call g: split contraction of the generic float32 kernels (A/B per family) + the whole GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python3 scripts/bench_contraction.py > gpurun_out/r04_contraction_ab.txt 2>&1
rc=$?; echo "ab rc $rc"; grep -v amdgpu.ids gpurun_out/r04_contraction_ab.txt | tail -14
[ $rc -eq 124 ] || [ $rc -eq 137 ] && exit $rc          # a killed GPU step: no further GPU step in this call
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --durations=12 > gpurun_out/r04_pytest_gpu_full.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -24 gpurun_out/r04_pytest_gpu_full.txt
exit $rc
