#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/stagger.log
for s in 0 1 2 4 8; do
  ALSEP_CONV_STAGGER=$s timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('stagger',$s,'value',d['value'],'ms/step',d['ms_per_step'],'conv avg_us',r['avg_us'],'TF',r['achieved'])" | tee -a gpurun_out/stagger.log
done
