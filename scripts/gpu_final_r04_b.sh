#!/bin/bash
# Round-4 evidence, call B: the whole GPU suite in one process
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
step 1150 pytest python3 -u -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_gpu_full.log 2>&1
tail -18 gpurun_out/pytest_gpu_full.log
