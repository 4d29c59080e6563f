#!/bin/bash
# copy the summaries of scripts/gpu_final_r02.sh from gpurun_out/ (scratch) into profiles/ (tracked)
set -eu
cd "$(dirname "$0")/.."
T=${1:-r02_final}
cp gpurun_out/pytest_gpu_full.log profiles/${T}_pytest_gpu.txt
cp gpurun_out/smoke.log profiles/${T}_smoke.txt
cp gpurun_out/bench_default.log profiles/${T}_bench_line.json
cp gpurun_out/kernel_stats.csv profiles/${T}_kernel_stats.csv
cp gpurun_out/trace_summary.txt profiles/${T}_trace_summary.txt
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
cp gpurun_out/pmc_traffic.json profiles/${T}_pmc_traffic.json
cp gpurun_out/sq_table.txt profiles/${T}_sq_counters.txt
ls -la profiles/${T}_*
