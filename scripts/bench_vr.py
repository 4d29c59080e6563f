#!/usr/bin/env python3
"""Timing of the VR-architecture network at a production window: n_fft 2048 (1024 bins), 768 frames (512 + 2 x 128 offset),
nets_61968KB widths, random weights.  Prints ms per forward and the convolution TFLOP/s (fp32 MFMA peak 157 TFLOP/s)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from audiolab_amd import _lib  # noqa: E402
from audiolab_amd.vrnet import WIDTHS, VRNet, _Conv, random_state_dict  # noqa: E402


def main():
    variant, n_fft, frames = "nets_61968KB", 2048, 768
    ctx = _lib.Context("cuda:0")
    net = VRNet(n_fft, random_state_dict(WIDTHS[variant], seed=0), variant=variant, ctx=ctx)
    flops = [0.0]
    orig = net._conv

    def counted(L, x, y=None, c0=0):
        ho, wo = L.out_hw(x.shape[1], x.shape[2])
        flops[0] += 2.0 * x.shape[0] * ho * wo * L.cout * L.cin * L.kh * L.kw
        return orig(L, x, y, c0)
    net._conv = counted
    x = torch.rand((1, n_fft // 2 + 1, frames, 2), device="cuda") * 3
    net.forward_nhwc(x)
    torch.cuda.synchronize()
    per = flops[0]
    net._conv = orig
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        y = net.forward_nhwc(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert bool(torch.isfinite(y).all())
    print(f"VRNet {variant} n_fft={n_fft} frames={frames}: {dt * 1e3:.1f} ms/forward, {per / 1e12:.2f} TFLOP conv -> "
          f"{per / dt / 1e12:.1f} TFLOP/s = {per / dt / 157.3e12 * 100:.1f} % of the fp32 MFMA peak; "
          f"{frames * 1024 / 44100 / dt:.0f}x realtime per stem pair at hop 1024")


if __name__ == "__main__":
    main()
