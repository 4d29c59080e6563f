#!/bin/bash
# round 4, call n: library built with packed-fp32-ops off (SLP on): determinism of the half kernels, MDX23C A/B against the SLP+packed build, bench, tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
step 200 determinism python3 scripts/dbg/determinism.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_determinism.txt
step 150 mdx23c_product python3 scripts/dbg/mdx23c_slp_ab.py 2>&1 | grep "per 120"
DBG_LIB=libalsep_slp.so step 150 mdx23c_slp python3 scripts/dbg/mdx23c_slp_ab.py 2>&1 | grep "per 120"
step 150 mel_product python3 scripts/dbg/mdx23c_slp_ab.py vocals_mel_band_roformer.ckpt 2>&1 | grep "per 120"
step 150 bs_product python3 scripts/dbg/mdx23c_slp_ab.py model_bs_roformer_ep_368_sdr_12.9628.ckpt 2>&1 | grep "per 120"
step 600 pytest python3 -m pytest tests/test_roformer.py tests/test_mdx23c.py tests/test_gpu_parity.py tests/test_htdemucs.py -m gpu -q -x > gpurun_out/r04_n_pytest.txt 2>&1; tail -3 gpurun_out/r04_n_pytest.txt
step 400 bench python3 bench.py --no-cpu-baseline > gpurun_out/r04_n_bench.json 2> gpurun_out/r04_n_bench.err
python3 - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r04_n_bench.json") if l.startswith("{")][-1])
print(d["ms_per_step"], d["roofline"]["frac"], {k: v.get("ms_per_step") for k, v in d.get("precision", {}).items()}, d["stages"]["stft_first_conv"]["frac"])
PY
