#!/bin/bash
# round 4, call s: iSTFT run length chosen per launch for the generic kernel too: tests, Roformer / MDX23C track times, PMC passes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
step 600 pytest python3 -m pytest tests/test_roformer.py tests/test_mdx23c.py tests/test_gpu_parity.py tests/test_vr_frontend.py tests/test_htdemucs.py tests/test_fused_front.py -m gpu -q -x > gpurun_out/r04_s_pytest.txt 2>&1; tail -2 gpurun_out/r04_s_pytest.txt
step 150 mel python3 scripts/dbg/mdx23c_slp_ab.py vocals_mel_band_roformer.ckpt 2>&1 | grep "per 120"
step 150 bs python3 scripts/dbg/mdx23c_slp_ab.py model_bs_roformer_ep_368_sdr_12.9628.ckpt 2>&1 | grep "per 120"
step 150 mdx23c python3 scripts/dbg/mdx23c_slp_ab.py 2>&1 | grep "per 120"
bash scripts/gpu_final_r04_a.sh > gpurun_out/r04_s_pmc.txt 2>&1; grep "rc=" gpurun_out/r04_s_pmc.txt | tr '\n' ' '
