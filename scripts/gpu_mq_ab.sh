#!/bin/bash
# same-box A/B of the level-1 conv kernels: merged (mny), double-buffered (mq), mq with progress priority
for m in "ALSEP_CONV_MNY=1" "ALSEP_CONV_MQ=1" "ALSEP_CONV_MQ=1 ALSEP_CONV_MQ_PRIO=1" "ALSEP_CONV_MNY=1" "ALSEP_CONV_MQ=1" "ALSEP_CONV_MQ=1 ALSEP_CONV_MQ_PRIO=1"; do
  env $m timeout -k 10 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels']; print('$m', d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_us'], {n:v['avg_us'] for n,v in k.items()})" || exit 1
done
for p in 0 1; do
ALSEP_CONV_MQ=1 ALSEP_CONV_MQ_PRIO=$p ALSEP_CONV_BIG_STAMP=2 timeout -k 10 300 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 2>&1 >/dev/null | grep -A3 "mq.*stamp\]" | tail -4 | cut -c1-330
done
