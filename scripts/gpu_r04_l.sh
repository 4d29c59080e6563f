#!/bin/bash
# round 4, call l: restructured attention (64-key chunks, 128 queries per workgroup) + v_rcp GELU: parity, microbench, Roformer workloads
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout -k 10 400 python3 -m pytest tests/test_roformer.py -m gpu -q -x > gpurun_out/r04_l_pytest.txt 2>&1
rc=$?; echo "pytest rc $rc"; tail -5 gpurun_out/r04_l_pytest.txt
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python3 scripts/bench_gemm_h.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_gemm_h_microbench.txt
rc=$?; cat gpurun_out/r04_gemm_h_microbench.txt
[ $rc -eq 0 ] || exit $rc
for m in vocals_mel_band_roformer.ckpt model_bs_roformer_ep_368_sdr_12.9628.ckpt; do
  timeout -k 10 300 python3 bench.py --workload model --model $m --dtype f16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04_l_$m.json 2> gpurun_out/r04_l_$m.err
  rc=$?; echo "bench $m rc $rc"
  [ $rc -eq 0 ] || exit $rc
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/r04_l_$m.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d.get('realtime_factor'))
"
done
