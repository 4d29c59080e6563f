#!/bin/bash
# A/B of environment switch combinations on the bench: usage gpu_ab2.sh "A=1 B=0" "A=0 B=0" ...   (first: bit-identity tests)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_conv_variants.py tests/test_gpu_parity.py -m gpu -q -x -k "identical or streaming or net_ or full_size or tdf" 2>&1 | tail -3 | tee gpurun_out/ab2_pytest.log
: > gpurun_out/ab2.log
for combo in "$@"; do
  env $combo timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$combo','value',d['value'],'ms/step',d['ms_per_step'], 'roofline',d['roofline']['kernel'],d['roofline']['avg_us'],d['roofline']['launches'], {k:(v['avg_us'],v['launches']) for k,v in d['kernels'].items()})" | tee -a gpurun_out/ab2.log
done
