#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_mdx23c.py -m gpu -q -x -s > $O/r03_m_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -i "mdx23c\|passed\|failed" $O/r03_m_tests.log | tail -8
[ $rc -eq 0 ] || exit $rc
python3 scripts/bench_conv_h.py > $O/r03_conv_h_microbench.txt 2>&1; grep "level [345]" $O/r03_conv_h_microbench.txt | cut -c1-150
python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt 2>&1 | grep "ms for"
python3 scripts/bench_lanes.py MDX23C-8KFFT-InstVoc_HQ.ckpt 2>&1 | grep "ms for"
