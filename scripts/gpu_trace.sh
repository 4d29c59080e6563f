#!/bin/bash
# kernel trace of the bench (2 steps), grouped per (kernel, grid)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -1 gpurun_out/bench_prof.log | cut -c1-200
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv
python scripts/trace_summary.py "$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)" 60 > gpurun_out/trace_summary.txt
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
grep -E "stream|pix_gemm" gpurun_out/trace_summary.txt
