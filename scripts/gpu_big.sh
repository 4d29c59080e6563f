#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_variants.py -m gpu -q -x 2>&1 | tail -5
: > gpurun_out/big.log
for m in 0 1; do
  ALSEP_CONV_BIG=$m timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('big=$m','value',d['value'],'ms/step',d['ms_per_step'],'plain conv avg_us',r['avg_us'],'TF',r['achieved'])" | tee -a gpurun_out/big.log
done
bash scripts/gpu_bench.sh > /dev/null 2>&1
