#!/bin/bash
# round 4, call p: persistent fused STFT front end: bit-identity tests + bench stages
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
step 400 pytest python3 -m pytest tests/test_fused_front.py tests/test_gpu_parity.py tests/test_configs.py -m gpu -q -x > gpurun_out/r04_p_pytest.txt 2>&1; tail -3 gpurun_out/r04_p_pytest.txt
step 400 bench python3 bench.py --no-cpu-baseline --no-precision > gpurun_out/r04_p_bench.json 2> gpurun_out/r04_p_bench.err
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/r04_p_bench.json') if l.startswith('{')][-1])
print(d['ms_per_step'])
for k,v in d['stages'].items(): print(k, v.get('frac'), v.get('us_per_launch'))
"
