#!/bin/bash
# Round-end evidence: GPU tests, smoke, default bench (with cpu baseline), rocprof kernel stats, PMC traffic.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -v --timeout 420 > gpurun_out/pytest_gpu_full.log 2>&1; tail -4 gpurun_out/pytest_gpu_full.log | tee gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.log
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.log
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp "$f" gpurun_out/kernel_stats.csv
python scripts/trace_summary.py "$(find gpurun_out/prof -name '*kernel_trace.csv' | head -1)" 40 > gpurun_out/trace_summary.txt
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
rm -rf gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 1500 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/$c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline > gpurun_out/pmc_$c.log 2>&1
done
python scripts/pmc_summary.py gpurun_out/pmc > /dev/null
python scripts/pmc_traffic.py gpurun_out/pmc/pmc_summary.json gpurun_out/pmc_traffic.json | head -12
find gpurun_out/pmc -name "*counter_collection.csv" -delete
