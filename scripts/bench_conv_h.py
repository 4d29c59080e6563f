"""Micro-benchmark of alsep_nn_conv2d_f16 / alsep_nn_instnorm_f16 at MDX23C-8KFFT-InstVoc_HQ's shapes (one 5.9 s chunk: 256 frames x 1024
sub-band bins at level 0, halved per level; 128 (level + 1) channels)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from audiolab_amd import _lib
ctx = _lib.Context("cuda:0")
lib, h = ctx.lib, ctx.handle


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


for lvl in range(6):
    H, W, C = 256 >> lvl, 1024 >> lvl, 128 * (lvl + 1)
    for (cin, cout, k, st, tag) in ((C, C, 3, 1, "tfc 3x3"), (C, C, 1, 1, "shortcut 1x1"), (2 * C, C, 3, 1, "decoder tfc1 3x3 (cat)"), (C, C + 128, 2, 2, "down 2x2/2"),
                                    (C, 4 * max(C - 128, 128), 1, 1, "up (1x1 to 4 Cout)")):
        if lvl == 5 and "cat" in tag or lvl == 5 and "down" in tag or (lvl == 0 and "up" in tag):
            continue
        pad = 1 if k == 3 else 0
        Ho, Wo = (H + 2 * pad - k) // st + 1, (W + 2 * pad - k) // st + 1
        x = torch.randn(H * W, cin, device="cuda").half()
        w = (torch.randn(cout, k, k, cin, device="cuda") / (k * k * cin) ** 0.5).half()
        y = torch.empty(Ho * Wo, cout, device="cuda")
        need = int(lib.alsep_nn_conv2d_f16_workspace_bytes(1, H, W, cin, cout, k, k, st, st, pad, pad))
        cws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
        dt = timed(lambda: ctx.check(lib.alsep_nn_conv2d_f16(h, _lib.ptr(x), _lib.ptr(w), _lib.ptr(y), None, cout, 1, H, W, cin, cout, k, k, st, st, pad, pad,
                                                             cout, 0, _lib.ptr(cws) if need else None, need), "conv"))
        fl = 2.0 * Ho * Wo * cout * k * k * cin
        by = 2.0 * H * W * cin + 2.0 * cout * k * k * cin + 4.0 * Ho * Wo * cout
        wgs = -(-cout // 128) * -(-(Ho * Wo) // 128)
        print(f"conv_hh level {lvl} {tag:26s} {H:3d}x{W:4d} {cin:4d}->{cout:4d}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:7.1f} TFLOP/s  {by / dt / 1e9:6.0f} GB/s  "
              f"{wgs:5d} tiles, {k * k * cin // 64:4d} K slices, split workspace {need >> 20} MiB")
    P = H * W
    x = torch.randn(P, C, device="cuda"); g = torch.ones(C, device="cuda"); b = torch.zeros(C, device="cuda")
    ws = torch.empty(int(lib.alsep_nn_instnorm_workspace_bytes(P, C)), dtype=torch.uint8, device="cuda")
    y16 = torch.empty(P, C, device="cuda", dtype=torch.float16)
    dt = timed(lambda: ctx.check(lib.alsep_nn_instnorm_f16(h, _lib.ptr(x), _lib.ptr(y16), _lib.ptr(g), _lib.ptr(b), P, C, 1e-5, 3, _lib.ptr(ws)), "in"))
    print(f"instnorm_f16 level {lvl} {P:6d} x {C:4d}: {dt * 1e6:8.1f} us  {P * C * 10 / dt / 1e9:6.0f} GB/s (two reads + half write)")
