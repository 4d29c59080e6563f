#!/bin/bash
# conv kernel change: bit-identity of the variants, network parity, then the bench with a kernel trace
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "identical or streaming or net_ or full_size" 2>&1 | tail -5 | tee gpurun_out/conv_pytest.log
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value',d['value'],'ms/step',d['ms_per_step'], d['kernels'].get('conv3x3_bf16_regw_kernel<1>',{}).get('avg_us'), d['roofline']['avg_us'])" | tee gpurun_out/conv.log
