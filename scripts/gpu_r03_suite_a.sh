#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -q -x --ignore=tests/test_configs.py --ignore=tests/test_gpu_parity.py --durations=8 > $O/r03_suite_a.log 2>&1
echo "suite A rc=$?"; tail -14 $O/r03_suite_a.log
