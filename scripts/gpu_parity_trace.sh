#!/bin/bash
# kernel change: variant bit-identity + network parity tests, then the bench under a kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "identical or streaming or net_ or full_size or tdf" 2>&1 | tail -3 | tee gpurun_out/pt_pytest.log
grep -q passed gpurun_out/pt_pytest.log && ! grep -q failed gpurun_out/pt_pytest.log && bash scripts/gpu_trace.sh
