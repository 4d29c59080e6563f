#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_variants.py -m gpu -q -x 2>&1 | tail -5
: > gpurun_out/pipe.log
for m in "0 0" "1 0" "1 1"; do
  set -- $m
  ALSEP_CONV_REGW=$1 ALSEP_CONV_PIPE=$2 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('regw,pipe=$m','value',d['value'],'ms/step',d['ms_per_step'],'conv avg_us',r['avg_us'],'TF',r['achieved'])" | tee -a gpurun_out/pipe.log
done
bash scripts/gpu_bench.sh > /dev/null 2>&1
python scripts/trace_summary.py "" 14 2>/dev/null || true
