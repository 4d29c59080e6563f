#!/bin/bash
# round 4, call r: vectorised GroupNorm / activation / scale-add kernels: HTDemucs + nn tests, htdemucs track time, then PMC + bench evidence
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
step 500 pytest python3 -m pytest tests/test_htdemucs.py tests/test_nn_ops.py tests/test_mdx23c.py tests/test_roformer.py tests/test_configs.py -m gpu -q -x > gpurun_out/r04_r_pytest.txt 2>&1; tail -2 gpurun_out/r04_r_pytest.txt
step 300 demucs python3 scripts/prof_host_demucs_runner.py 2>&1 | grep "^lanes"
step 300 demucs6 python3 bench.py --workload demucs6 --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('demucs6', d['ms_per_step'], d.get('realtime_factor'))
"
bash scripts/gpu_final_r04_a.sh > gpurun_out/r04_r_pmc.txt 2>&1; tail -2 gpurun_out/r04_r_pmc.txt | cut -c1-100
