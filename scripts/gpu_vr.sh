#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k vrnet 2>&1 | tail -3 | tee gpurun_out/vr_pytest.log
timeout 600 python scripts/bench_vr.py 2>&1 | tail -2 | tee gpurun_out/vr_bench.log
