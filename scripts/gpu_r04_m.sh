#!/bin/bash
# round 4, call m: the whole library without the SLP vectoriser's packed float32 arithmetic: bit-exact parity of the TFC path + bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_roformer.py -m gpu -q -x > gpurun_out/r04_m_pytest.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -4 gpurun_out/r04_m_pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 400 python3 bench.py --no-cpu-baseline > gpurun_out/r04_m_bench.json 2> gpurun_out/r04_m_bench.err
rc=$?; echo "bench rc $rc"; python3 - <<'PY'
import json
d = json.loads(open("gpurun_out/r04_m_bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["roofline"]["frac"], {k: v.get("ms_per_step") for k, v in d.get("precision", {}).items()})
PY
exit $rc
