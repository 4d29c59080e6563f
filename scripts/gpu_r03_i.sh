#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_roformer.py tests/test_mdx23c.py -m gpu -q -x > $O/r03_i_tests.log 2>&1
echo "tests rc=$?"; tail -3 $O/r03_i_tests.log
python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt MDX23C-8KFFT-InstVoc_HQ.ckpt 2>&1 | grep "ms for\|Error\|error" | head
ALSEP_RUNNER_GRAPH=0 python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt 2>&1 | grep "ms for"
python3 scripts/bench_lanes.py vocals_mel_band_roformer.ckpt 2>&1 | grep "ms for"
