#!/bin/bash
# MDX23C half mode, VR resamplers / BVE entry, Demucs; MDX23C 120 s in both precisions
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_mdx23c.py tests/test_vr_frontend.py tests/test_htdemucs.py -m gpu -q -x -s --durations=5 > $O/r03_k_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; grep -i "mdx23c\|passed\|failed" $O/r03_k_tests.log | tail -12
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --workload model --model MDX23C-8KFFT-InstVoc_HQ.ckpt --steps 2 --warmup 1 > $O/r03_model_mdx23c_f16.json 2> $O/r03_model_mdx23c_f16.err; echo "bench f16 rc=$?"
timeout -k 10 300 python bench.py --workload model --model MDX23C-8KFFT-InstVoc_HQ.ckpt --dtype f32 --steps 2 --warmup 1 > $O/r03_model_mdx23c_f32.json 2> $O/r03_model_mdx23c_f32.err; echo "bench f32 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r03_model_mdx23c_f16.json","gpurun_out/r03_model_mdx23c_f32.json"):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'], d.get('roofline',{}).get('kernel'), d.get('roofline',{}).get('frac'))
    except Exception as e: print(f, "ERR", e)
PY
