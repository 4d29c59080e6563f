#!/bin/bash
# same-box A/B of the level-0 conv: register-weight kernel, LDS-resident-weight kernel (ALSEP_CONV_M0=1) without / with the younger half's
# epilogue deferred behind the next barrier, then the phase stamps of both forms
for m in "ALSEP_CONV_M0=0" "ALSEP_CONV_M0=1 ALSEP_CONV_M0_DEFER=0" "ALSEP_CONV_M0=1 ALSEP_CONV_M0_DEFER=1" "ALSEP_CONV_M0=0" "ALSEP_CONV_M0=1 ALSEP_CONV_M0_DEFER=0" "ALSEP_CONV_M0=1 ALSEP_CONV_M0_DEFER=1"; do
  env $m timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; r=d['roofline']; print('$m', d['ms_per_step'], r['kernel'], r['avg_us'], {n:v['avg_us'] for n,v in k.items()})" || exit 1
done
for d in 0 1; do
ALSEP_CONV_M0=1 ALSEP_CONV_M0_DEFER=$d ALSEP_CONV_M0_STAMP=2 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 2>&1 >/dev/null | grep -A3 "m0.*stamp\]" | tail -4 | cut -c1-330
done
