#!/bin/bash
# round 3: new GPU tests (reverb, loaders-on-GPU paths, configs, half-precision Roformer) + timings / kernel stats of the Roformers in the
# half-precision mode
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out
mkdir -p $O
timeout -k 10 1000 python -m pytest tests/test_reverb.py tests/test_roformer.py tests/test_configs.py tests/test_engine_e2e.py tests/test_gpu_conv_variants.py -m gpu -q -s -x > $O/r03_c_tests.log 2>&1
echo "tests rc=$?"; grep -E "roformer\[|reverb at|configs\[|passed|failed|Error|error" $O/r03_c_tests.log | cut -c1-400 | tail -30
python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt 2>&1 | grep "ms for"
prof() {
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" > $O/prof_$tag.log 2>&1
  echo "prof $tag rc=$?"; grep "ms for" $O/prof_$tag.log | cut -c1-300
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/r03_${tag}_kernel_stats.csv
}
prof melband_half python3 scripts/bench_lanes.py --half vocals_mel_band_roformer.ckpt
