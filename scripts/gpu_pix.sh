#!/bin/bash
# streaming ds/us kernels for the 96<->144 level: parity (bit identity vs the tile GEMM, network vs torch), then A/B on the bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "streaming or net_ or full_size" 2>&1 | tail -5 | tee gpurun_out/pix_pytest.log
: > gpurun_out/pix.log
for m in 0 1; do
  ALSEP_PIX_STREAM=$m timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pix_stream=$m','value',d['value'],'ms/step',d['ms_per_step'])" | tee -a gpurun_out/pix.log
done
