#!/bin/bash
# Round-4 evidence, call C: smoke, default bench line (precision + accuracy objects, PMC traffic from profiles/pmc_traffic.json), rocprof kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
step 300 smoke python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; tail -2 gpurun_out/smoke.log
step 700 bench python3 bench.py > gpurun_out/bench_default.log 2> gpurun_out/bench_default.err; tail -1 gpurun_out/bench_default.log | cut -c1-300
rm -rf /tmp/prof
step 400 prof rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-precision > gpurun_out/bench_prof.log 2>&1
f=$(find /tmp/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv
python3 scripts/trace_summary.py "$(find /tmp/prof -name '*kernel_trace.csv' | head -1)" 60 > gpurun_out/trace_summary.txt; head -5 gpurun_out/trace_summary.txt | cut -c1-200
