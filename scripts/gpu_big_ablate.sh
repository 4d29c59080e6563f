#!/bin/bash
# in-kernel phase stamps of conv3x3_bf16_big_kernel<2> with parts of the k-loop removed in turn (timing experiment, wrong results)
mkdir -p gpurun_out
for a in 0 1 2 4 6 7 8 16 32 100; do
  ALSEP_CONV_BIG_STAMP=3 ALSEP_CONV_BIG_ABL=$a timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 > /dev/null 2> gpurun_out/abl_$a.txt || true
  grep "big<2> stamp" gpurun_out/abl_$a.txt | tail -1
done
