#!/bin/bash
# Round-3 evidence, last call: smoke, default bench line, rocprof kernel stats, MDX23C model lines and half-mode kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scripts/gpu_final_r03_c.sh
export TMPDIR=/tmp
for d in "" "--dtype f32"; do
  t=$([ -z "$d" ] && echo half || echo f32)
  timeout -k 10 300 python bench.py --workload model --model MDX23C-8KFFT-InstVoc_HQ.ckpt $d --steps 2 --warmup 1 > gpurun_out/r03_model_mdx23c_$t.json 2> gpurun_out/r03_model_mdx23c_$t.err
  echo "mdx23c $t rc=$?"; cut -c1-200 gpurun_out/r03_model_mdx23c_$t.json
done
rm -rf /tmp/prof_m
ALSEP_RUNNER_GRAPH=0 timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_m -o p -- python3 scripts/bench_lanes.py --half MDX23C-8KFFT-InstVoc_HQ.ckpt > gpurun_out/prof_mdx23c_half.log 2>&1
f=$(find /tmp/prof_m -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_mdx23c_half_kernel_stats.csv
