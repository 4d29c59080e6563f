#!/bin/bash
# round 4, call q: tile-height rule of the persistent GEMM: Roformer tests, the two Roformers' track times, then the PMC passes and the bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
. scripts/gpu_final_common.sh
mkdir -p gpurun_out
step 300 pytest python3 -m pytest tests/test_roformer.py -m gpu -q -x > gpurun_out/r04_q_pytest.txt 2>&1; tail -2 gpurun_out/r04_q_pytest.txt
step 150 mel python3 scripts/dbg/mdx23c_slp_ab.py vocals_mel_band_roformer.ckpt 2>&1 | grep "per 120"
step 150 bs python3 scripts/dbg/mdx23c_slp_ab.py model_bs_roformer_ep_368_sdr_12.9628.ckpt 2>&1 | grep "per 120"
bash scripts/gpu_final_r04_a.sh
