#!/bin/bash
# HBM traffic counters for the bench kernels: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), no tracing domains.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/pmc
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc/$c -- python bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 > gpurun_out/pmc_$c.log 2>&1
  tail -2 gpurun_out/pmc_$c.log | cut -c1-300
done
python scripts/pmc_summary.py gpurun_out/pmc | head -60
find gpurun_out/pmc -name "*counter_collection.csv" -size +8M -delete
