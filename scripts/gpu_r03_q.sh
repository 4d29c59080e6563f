#!/bin/bash
# no-scratch half kernels: lane tests, model tests, timings
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_mdx23c.py tests/test_roformer.py tests/test_htdemucs.py tests/test_gpu_parity.py -m gpu -q -x --durations=6 > $O/r03_q_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -10 $O/r03_q_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt 2>&1 | grep "ms for"
python3 scripts/bench_lanes.py htdemucs_6s.yaml 2>&1 | grep "ms for"
python3 scripts/bench_gemm_h.py 2>&1 | head -4
python3 scripts/bench_conv_h.py 2>&1 | grep "level [01] tfc"
