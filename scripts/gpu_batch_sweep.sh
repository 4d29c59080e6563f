#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for b in 1 2 8 13; do
  timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch $b 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batch',$b,'value',d['value'],'ms',d['ms_per_step'],'conv TF',d['roofline']['achieved'])" | tee -a gpurun_out/batch_sweep.log
done
