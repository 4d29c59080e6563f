#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
: > gpurun_out/ablate.log
for a in 0 1 2 4 3 6; do
  ALSEP_CONV_ABLATE=$a timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ablate',$a,'ms/step',d['ms_per_step'],'conv avg_us',r['avg_us'],'TF',r['achieved'], 'stft',d['stages']['stft']['achieved'],'istft',d['stages']['istft']['achieved'])" | tee -a gpurun_out/ablate.log
done
