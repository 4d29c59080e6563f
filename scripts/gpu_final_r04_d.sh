#!/bin/bash
# Round-4 evidence, call D: the other workloads' bench lines, the model-family lines, and the Mel-Band half-precision kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
export TMPDIR=/tmp
for wl in demucs6 tracks longform; do
  step 400 $wl python3 bench.py --workload $wl --steps 2 --warmup 1 > gpurun_out/r04_workload_$wl.json 2> gpurun_out/r04_workload_$wl.err
  tail -1 gpurun_out/r04_workload_$wl.json | cut -c1-260
done
for m in vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt MDX23C-8KFFT-InstVoc_HQ.ckpt; do
  t=$(echo $m | cut -c1-12)
  step 300 $t python3 bench.py --workload model --model $m --dtype f16 --steps 2 --warmup 1 > gpurun_out/r04_model_${t}_half.json 2> gpurun_out/r04_model_${t}_half.err
  tail -1 gpurun_out/r04_model_${t}_half.json | cut -c1-260
done
step 300 mdx23c_f32 python3 bench.py --workload model --model MDX23C-8KFFT-InstVoc_HQ.ckpt --dtype f32 --steps 2 --warmup 1 > gpurun_out/r04_model_MDX23C_f32.json 2> gpurun_out/r04_model_MDX23C_f32.err
tail -1 gpurun_out/r04_model_MDX23C_f32.json | cut -c1-200
rm -rf /tmp/prof_m
step 300 prof_mel rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -- python3 scripts/prof_host_roformer.py vocals_mel_band_roformer.ckpt > gpurun_out/prof_mel_half.log 2>&1
f=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_roformer_mel_half_kernel_stats.csv
rm -rf /tmp/prof_b
step 300 prof_bs rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b -- python3 scripts/prof_host_roformer.py model_bs_roformer_ep_368_sdr_12.9628.ckpt > gpurun_out/prof_bs_half.log 2>&1
f=$(find /tmp/prof_b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_roformer_bs_half_kernel_stats.csv
