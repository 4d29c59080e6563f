#!/bin/bash
# kernel stats of MDX23C in the half-precision mode (graphs off so that kernels are traced individually), lanes 4 and 1
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
prof() {
  tag=$1; shift
  rm -rf /tmp/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- "$@" > $O/prof_$tag.log 2>&1
  echo "prof $tag rc=$?"; grep "ms for" $O/prof_$tag.log | cut -c1-300
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" $O/r03_${tag}_kernel_stats.csv
}
python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt 2>&1 | grep "ms for"
ALSEP_RUNNER_GRAPH=0 python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt 2>&1 | grep "ms for"
ALSEP_RUNNER_GRAPH=0 ALSEP_RUNNER_LANES=1 python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt 2>&1 | grep "ms for"
export ALSEP_RUNNER_GRAPH=0
prof mdx23c_half python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt
