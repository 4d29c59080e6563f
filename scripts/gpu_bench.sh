#!/bin/bash
# usage: gpu_bench.sh [bench args]   -- bench + rocprof kernel stats on the GPU box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" 2>&1 | tail -3 | tee gpurun_out/bench.log
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/bench_prof.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv && head -14 "$f" | cut -c1-160
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
