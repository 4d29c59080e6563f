#!/bin/bash
# round 4, call c: the split-half float32 mode -- parity at full size, smoke, and the bench line with precision / accuracy
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -s -k "full_size_mdx_f32 or half_range" > gpurun_out/r04_c_pytest.txt 2>&1; echo "pytest rc $?"; grep -E "full-size|passed|failed|Error" gpurun_out/r04_c_pytest.txt | cut -c1-400
timeout -k 10 120 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 600 python3 bench.py --steps 3 --warmup 1 > gpurun_out/r04_c_bench.json 2> gpurun_out/r04_c_bench.err; echo "bench rc $?"; python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_c_bench.json').read().strip().splitlines()[-1])
print('ms_per_step', d['ms_per_step'], 'value', d['value'])
print('precision', json.dumps(d['precision']))
print('accuracy', json.dumps(d['accuracy']))
PY
