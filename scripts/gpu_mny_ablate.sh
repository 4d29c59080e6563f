#!/bin/bash
# in-kernel phase stamps of conv3x3_bf16_mny_kernel with its LDS-DMA streams / stores removed in turn (timing experiment, wrong results)
mkdir -p gpurun_out
for a in 0 1 2 3 4 7; do
  ALSEP_CONV_MNY=3 ALSEP_CONV_BIG_STAMP=2 ALSEP_CONV_BIG_ABL=$a timeout -k 10 200 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --seconds 60 > /dev/null 2> gpurun_out/mabl_$a.txt || true
  grep -A2 "stamp\]" gpurun_out/mabl_$a.txt | grep -A2 "mny<2>" | tail -3
done
