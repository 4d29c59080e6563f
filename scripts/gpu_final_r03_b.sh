#!/bin/bash
# Round-3 evidence, call B: the whole GPU suite in one process
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 1150 python -u -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_gpu_full.log 2>&1
echo "pytest rc=$?"; tail -18 gpurun_out/pytest_gpu_full.log
