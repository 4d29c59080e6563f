#!/bin/bash
# A/B of an environment switch on the bench: usage gpu_ab.sh VAR v1 v2 ...
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
var=$1; shift
: > gpurun_out/ab.log
for v in "$@"; do
  env $var=$v timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v','value',d['value'],'ms/step',d['ms_per_step'], 'roofline',d['roofline']['kernel'],d['roofline']['avg_us'], {k:v['avg_us'] for k,v in d['kernels'].items()})" | tee -a gpurun_out/ab.log
done
