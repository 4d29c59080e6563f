#!/bin/bash
# round 4, call d: kernel statistics of the float32 split mode
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "half_range" 2>&1 | tail -3
rm -rf gpurun_out/f32s
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/f32s -- python3 bench.py --dtype f32 --steps 1 --warmup 0 --no-precision --no-cpu-baseline --batch 13 > gpurun_out/r04_d_bench.log 2>&1; echo "rc $?"; tail -1 gpurun_out/r04_d_bench.log | cut -c1-300
python3 scripts/trace_summary.py $(find gpurun_out/f32s -name "*kernel_trace.csv" | head -1) 2>/dev/null | head -24
find gpurun_out/f32s -name "*.csv" -size +4M -delete
