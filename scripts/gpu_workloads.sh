#!/bin/bash
# Supplementary bench lines (BASELINE configs[2], [3], [4]) on one GPU, reduced lengths by default.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
D=${1:-120}; T=${2:-60}; L=${3:-300}
timeout -k 10 400 python bench.py --workload demucs6 --seconds $D --steps 1 --warmup 1 2>&1 | tail -1 | tee gpurun_out/wl_demucs6.json
timeout -k 10 400 python bench.py --workload tracks --tracks 2 --seconds $T --steps 1 --warmup 1 2>&1 | tail -1 | tee gpurun_out/wl_tracks.json
timeout -k 10 500 python bench.py --workload longform --seconds $L --steps 1 --warmup 0 2>&1 | tail -1 | tee gpurun_out/wl_longform.json
