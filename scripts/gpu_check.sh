#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, smoke, bench, rocprof kernel stats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocm-smi"; rocm-smi --showproductname 2>/dev/null | head -8
echo "== pytest -m gpu"
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -40 | tee gpurun_out/pytest_gpu.log
echo "== smoke"
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== bench"
timeout 900 python bench.py --steps 2 --warmup 1 2>&1 | tail -5 | tee gpurun_out/bench.log
echo "== rocprof kernel trace of bench (1 step, no cpu baseline)"
rm -rf gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
tail -3 gpurun_out/bench_prof.log
find gpurun_out/prof -name "*kernel_stats*" | head -3
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -25 "$f" | cut -c1-200
# keep only the small summaries (the per-dispatch trace can be large)
find gpurun_out/prof -name "*kernel_trace.csv" -size +20M -delete
