#!/bin/bash
# round 4, call h: fused STFT epilogue with whole-line stores: parity + bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 300 python3 -m pytest tests/test_fused_front.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/r04_h_pytest.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04_h_pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py > gpurun_out/r04_h_bench.json 2> gpurun_out/r04_h_bench.err
rc=$?; echo "bench rc $rc"; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_h_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"])
print(json.dumps(d.get("stages"), indent=0)[:1500])
PY
exit $rc
