#!/bin/bash
# Round-3 evidence, call E: single-model lines again (roofline passes with plain launches), fp32 default line with the warmed-up precision step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
for m in vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt MDX23C-8KFFT-InstVoc_HQ.ckpt; do
  t=$(echo $m | cut -c1-12)
  timeout -k 10 300 python bench.py --workload model --model $m --steps 2 --warmup 1 > gpurun_out/r03_model_$t.json 2> gpurun_out/r03_model_$t.err
  echo "$m rc=$?"; cut -c1-200 gpurun_out/r03_model_$t.json
done
timeout -k 10 300 python bench.py --workload model --model MDX23C-8KFFT-InstVoc_HQ.ckpt --dtype f32 --steps 2 --warmup 1 > gpurun_out/r03_model_MDX23C_f32.json 2> gpurun_out/r03_model_MDX23C_f32.err
echo "mdx23c f32 rc=$?"
timeout -k 10 600 python bench.py 2>gpurun_out/bench_default.err | tail -1 | tee gpurun_out/bench_default.log | cut -c1-200
