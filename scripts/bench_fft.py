#!/usr/bin/env python3
"""Micro-benchmark of the STFT / iSTFT stages alone (bench workload geometry: 52 model windows of
dim_t 256 x dim_f 3072, n_fft 6144, hop 1024), HIP-event timed through the library's own profile hooks.
Prints one line per (stage, dtype): us per launch, algorithmic GB/s (SURVEY 8(d) bytes), fraction of 8 TB/s."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiolab_amd import _lib  # noqa: E402
from audiolab_amd.mdx import StftPlan  # noqa: E402
from audiolab_amd.synth import synth_mix  # noqa: E402


def main():
    n_fft = int(os.environ.get("N_FFT", "6144"))
    dim_f = int(os.environ.get("DIM_F", "3072"))
    nb, dim_t, hop = 52, 256, 1024
    ctx = _lib.Context("cuda:0")
    plan = StftPlan(ctx, n_fft, hop, dim_f, dim_t)
    gen = plan.gen_size
    pad_len = plan.trim * 2 + nb * gen + plan.chunk_size
    mix = torch.from_numpy(synth_mix(pad_len)).cuda()
    outb = torch.empty((2, nb * gen), device="cuda")
    for dtype in (torch.bfloat16, torch.float32):
        es = 2 if dtype == torch.bfloat16 else 4
        spec = plan.stft_strided(mix, pad_len, gen, nb, dtype, _lib.LAYOUT_NHWC)
        torch.cuda.synchronize()
        for name, cat in (("stft", _lib.PROF_STFT), ("istft", _lib.PROF_ISTFT)):
            reps = 20
            for timed in (False, True):
                if timed:
                    ctx.profile_begin(cat)
                for _ in range(reps):
                    if name == "stft":
                        plan.stft_strided(mix, pad_len, gen, nb, dtype, _lib.LAYOUT_NHWC, out=spec)
                    else:
                        plan.istft_strided(spec, _lib.LAYOUT_NHWC, outb, nb * gen, gen, plan.trim,
                                           plan.chunk_size - plan.trim, nb * gen)
                torch.cuda.synchronize()
            ms, launches = ctx.profile_end()
            spec_bytes = 4 * dim_f * dim_t * es
            alg = 2 * plan.chunk_size * 4 + spec_bytes if name == "stft" else spec_bytes + 3 * 2 * plan.chunk_size * 4
            gbs = alg * nb * reps / (ms * 1e-3) / 1e9
            print(f"{name} n_fft={n_fft} {str(dtype).split('.')[-1]}: {ms * 1e3 / launches:.1f} us/launch "
                  f"({ms * 1e3 / launches / nb:.2f} us/chunk), {gbs:.0f} GB/s = {gbs / 80:.1f} % of 8 TB/s", flush=True)


if __name__ == "__main__":
    main()
