#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_mdx23c.py tests/test_roformer.py -m gpu -q -x -k "lanes_and_graphs" > $O/r03_p_tests.log 2>&1
echo "tests rc=$?"; tail -12 $O/r03_p_tests.log | cut -c1-300
