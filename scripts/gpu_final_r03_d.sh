#!/bin/bash
# Round-3 evidence, call D: the other workloads' bench lines (roofline of the dominant kernel class + CPU baseline each)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in demucs6 tracks longform; do
  timeout -k 10 400 python bench.py --workload $wl --steps 2 --warmup 1 > gpurun_out/r03_workload_$wl.json 2> gpurun_out/r03_workload_$wl.err
  echo "$wl rc=$?"; cut -c1-260 gpurun_out/r03_workload_$wl.json
done
for m in vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt MDX23C-8KFFT-InstVoc_HQ.ckpt; do
  t=$(echo $m | cut -c1-12)
  timeout -k 10 300 python bench.py --workload model --model $m --steps 2 --warmup 1 > gpurun_out/r03_model_$t.json 2> gpurun_out/r03_model_$t.err
  echo "$m rc=$?"; cut -c1-260 gpurun_out/r03_model_$t.json
done
timeout -k 10 300 python bench.py --workload model --model MDX23C-8KFFT-InstVoc_HQ.ckpt --dtype f32 --steps 2 --warmup 1 > gpurun_out/r03_model_MDX23C_f32.json 2> gpurun_out/r03_model_MDX23C_f32.err
echo "mdx23c f32 rc=$?"; cut -c1-260 gpurun_out/r03_model_MDX23C_f32.json
