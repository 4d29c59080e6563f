#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_mdx23c.py -m gpu -q -x -s > $O/r03_s_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -i "mdx23c\|passed\|failed" $O/r03_s_tests.log | tail -8 | cut -c1-220
[ $rc -eq 0 ] || exit $rc
python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt 2>&1 | grep "ms for"
