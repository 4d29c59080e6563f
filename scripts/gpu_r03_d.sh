#!/bin/bash
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
python3 scripts/bench_gemm_h.py 2>&1 | tee $O/r03_gemm_h_microbench.txt
timeout -k 10 600 python -m pytest tests/test_roformer.py -m gpu -q -s -x -k "half" > $O/r03_d_tests.log 2>&1
echo "tests rc=$?"; grep -E "roformer\[|passed|failed|Error" $O/r03_d_tests.log | cut -c1-300
python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt 2>&1 | grep "ms for"
ALSEP_RUNNER_LANES=1 python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt 2>&1 | grep "ms for"
timeout -k 10 500 python3 bench.py --workload model --model vocals_mel_band_roformer.ckpt --steps 2 --warmup 1 > $O/r03_model_melband.json 2> $O/r03_model_melband.err
echo "model bench rc=$?"; cat $O/r03_model_melband.json | cut -c1-1500; tail -3 $O/r03_model_melband.err
