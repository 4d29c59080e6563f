#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python3 scripts/bench_gemm_h.py 2>&1 | grep -v amdgpu.ids | tee $O/r03_gemm_h_microbench.txt
timeout -k 10 600 python -m pytest tests/test_roformer.py -m gpu -q -x -k "half" > $O/r03_e_tests.log 2>&1
echo "tests rc=$?"; tail -2 $O/r03_e_tests.log
python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt 2>&1 | grep "ms for"
