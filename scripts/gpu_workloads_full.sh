#!/bin/bash
# BASELINE configs[2], [3], [4] at their stated sizes on ONE GPU (the 8-GPU forms are the driver's to launch): htdemucs_6s 10 min;
# 8 tracks x 3 min MDX + Demucs per GPU (configs[3] is 64 tracks on 8 GPUs); 60 min 48 kHz 8 channels fp16 overlap 0.75
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 500 python bench.py --workload demucs6 --steps 1 --warmup 1 2>gpurun_out/wl_demucs6.err | tail -1 | tee gpurun_out/wl_demucs6.json | cut -c1-330
timeout -k 10 700 python bench.py --workload tracks --tracks 8 --steps 1 --warmup 1 2>gpurun_out/wl_tracks.err | tail -1 | tee gpurun_out/wl_tracks.json | cut -c1-330
timeout -k 10 500 python bench.py --workload longform --steps 1 --warmup 1 2>gpurun_out/wl_longform.err | tail -1 | tee gpurun_out/wl_longform.json | cut -c1-330
