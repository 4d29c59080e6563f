"""Direct tests of the generic fp32 kernels behind the Roformer / MDX23C / HTDemucs / VR host modules (csrc/nn.hip, vrnet.hip conv):
the strided batched GEMM in every operand layout its dispatch distinguishes (tiled [N][K] / [K][N] forms, float4 or scalar epilogue,
ragged K, edge tiles, the straight-from-L1 fallback) and the implicit-GEMM convolution (both tile sizes, the fallback for Cin % 16 != 0,
stride / dilation / padding, channel-slice outputs) against torch on the CPU in float64.  Same bodies on the emulation and on cuda:0."""
import ctypes as C

import numpy as np
import pytest
import torch

from audiolab_amd import _lib
from tests.conftest import host, on


def _arr(*v):
    return (C.c_int64 * 4)(*v)


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / 2 ** 0.5))


@pytest.mark.parametrize("case", [
    # (nb1, nb2, M, N, K, B layout "nk" | "kn", C row-major?, bias, act, A row padding)
    dict(nb=(1, 1), M=300, N=200, K=64, b="nk", ct=True, bias=True, act=3),          # tiled, float4 epilogue, edge tiles in M and N
    dict(nb=(2, 3), M=130, N=64, K=48, b="nk", ct=True, bias=False, act=0),           # batched, K = 3 slices
    dict(nb=(1, 2), M=200, N=132, K=801, b="kn", ct=True, bias=False, act=0, pad=3),  # P V: B = [K][N], ragged K, padded A rows
    dict(nb=(1, 1), M=96, N=70, K=40, b="nk", ct=False, bias=True, act=5),            # N % 4 != 0: scalar epilogue; K tail of 8
    dict(nb=(1, 1), M=257, N=128, K=36, b="kn", ct=False, bias=False, act=0),         # [K][N] with a transposed C
    dict(nb=(1, 1), M=50, N=30, K=33, b="nk", ct=True, bias=True, act=0),             # small product / unaligned K: the fallback kernel
    dict(nb=(1, 4), M=100, N=100, K=60, b="nk", ct=True, bias=False, act=0, pad=1),   # A rows not 16-byte aligned: fallback
])
def test_bgemm_layouts_vs_torch(dev, case):
    nb1, nb2 = case["nb"]
    M, N, K = case["M"], case["N"], case["K"]
    pad = case.get("pad", 0)
    g = torch.Generator().manual_seed(M * 7 + N)
    lda = K + pad
    A = torch.randn(nb1, nb2, M, lda, generator=g)
    Bm = torch.randn(nb1, nb2, N, K, generator=g) if case["b"] == "nk" else torch.randn(nb1, nb2, K, N, generator=g)
    bias = torch.randn(N, generator=g) if case["bias"] else None
    a64 = A[..., :K].double()
    b64 = Bm.double() if case["b"] == "nk" else Bm.double().transpose(-1, -2)
    want = 0.7 * (a64 @ b64.transpose(-1, -2))
    if bias is not None:
        want = want + bias.double()
    if case["act"] == 3:
        want = _gelu(want)
    elif case["act"] == 5:
        want = torch.tanh(want)
    Ad, Bd = on(dev, A), on(dev, Bm)
    if case["ct"]:
        Cd = dev.empty((nb1, nb2, M, N))
        sc = _arr(nb2 * M * N, M * N, N, 1)
    else:                                                       # C stored transposed: [N][M]
        Cd = dev.empty((nb1, nb2, N, M))
        sc = _arr(nb2 * M * N, M * N, 1, M)
    sa = _arr(nb2 * M * lda, M * lda, lda, 1)
    sb = _arr(nb2 * N * K, N * K, K, 1) if case["b"] == "nk" else _arr(nb2 * N * K, N * K, 1, N)
    bd = on(dev, bias) if bias is not None else None
    # both contractions of the float32 entry points: exact f32 MFMA, and split-half products on the f16 pipe (nn_f32s.h)
    try:
        for split in (False, True):
            dev.set_nn_contraction(split)
            dev.launch_counts_reset()
            Cd.zero_()
            dev.check(dev.lib.alsep_nn_bgemm_bias(dev.handle, _lib.ptr(Ad), _lib.ptr(Bd), _lib.ptr(Cd), nb1, nb2, M, N, K, sa, sb, sc, 0.7,
                                                  _lib.ptr(bd) if bd is not None else None, case["act"]), "alsep_nn_bgemm_bias")
            got = torch.from_numpy(host(Cd)).double()
            if not case["ct"]:
                got = got.transpose(-1, -2)
            scale = float(want.abs().max())
            assert float((got - want).abs().max()) < 2e-5 * max(1.0, scale), f"split={split}"
            tiled = dev.launch_count("nn_gemm_tn_kernel") + dev.launch_count("nn_gemm_split_kernel") > 0
            assert (dev.launch_count("nn_gemm_split_kernel") > 0) == (split and tiled)      # the split kernel takes every tiled case
            assert not dev.nn_range_exceeded()
    finally:
        dev.set_nn_contraction(False)


@pytest.mark.parametrize("case", [
    dict(B=2, H=20, W=24, cin=32, cout=48, k=(3, 3), s=(1, 1), p=(1, 1), d=(1, 1), act=1),              # tiled, small grid -> 64 x 64 tiles
    dict(B=1, H=64, W=96, cin=16, cout=200, k=(3, 3), s=(1, 1), p=(2, 2), d=(2, 2), act=3),             # 128 x 128 tiles, dilation, edge tile in N
    dict(B=2, H=33, W=1, cin=48, cout=64, k=(8, 1), s=(4, 1), p=(2, 0), d=(1, 1), act=0),               # Demucs Conv1d: stride 4 along H, W = 1
    dict(B=1, H=16, W=40, cin=64, cout=32, k=(1, 1), s=(1, 1), p=(0, 0), d=(1, 1), act=2, slice=(80, 8)),  # Linear into a channel slice (float4 stores)
    dict(B=1, H=16, W=40, cin=64, cout=32, k=(1, 1), s=(1, 1), p=(0, 0), d=(1, 1), act=0, slice=(70, 6)),  # ... an unaligned slice (scalar stores)
    dict(B=2, H=18, W=18, cin=4, cout=16, k=(3, 3), s=(2, 2), p=(1, 1), d=(1, 1), act=1),               # Cin % 16 != 0: the straight-from-L1 kernel
])
def test_conv2d_shapes_vs_torch(dev, case):
    B, H, W, cin, cout = case["B"], case["H"], case["W"], case["cin"], case["cout"]
    (kh, kw), (sh, sw), (ph, pw), (dh, dw) = case["k"], case["s"], case["p"], case["d"]
    g = torch.Generator().manual_seed(H * 31 + cout)
    x = torch.randn(B, H, W, cin, generator=g)
    w = torch.randn(cout, cin, kh, kw, generator=g) / (cin * kh * kw) ** 0.5
    scale = 1.0 + 0.2 * torch.randn(cout, generator=g)
    shift = 0.1 * torch.randn(cout, generator=g)
    y = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), stride=(sh, sw), padding=(ph, pw), dilation=(dh, dw))
    y = y * scale.double()[None, :, None, None] + shift.double()[None, :, None, None]
    act = case["act"]
    if act == 1:
        y = torch.relu(y)
    elif act == 2:
        y = torch.nn.functional.leaky_relu(y, 0.01)
    elif act == 3:
        y = _gelu(y)
    want = y.permute(0, 2, 3, 1)                                # [B, Ho, Wo, cout]
    Ho, Wo = want.shape[1], want.shape[2]
    ct, c0 = case.get("slice", (cout, 0))
    out = dev.zeros((B, Ho, Wo, ct))
    xd, wd = on(dev, x), on(dev, w.permute(2, 3, 1, 0).contiguous())       # [KH][KW][Cin][Cout]
    sd, hd = on(dev, scale), on(dev, shift)
    try:
        for split in (False, True):                            # exact f32 MFMA, then the split-half contraction (nn_f32s.h)
            dev.set_nn_contraction(split)
            dev.launch_counts_reset()
            out.zero_()
            dev.check(dev.lib.alsep_nn_conv2d(dev.handle, _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(sd), _lib.ptr(hd), _lib.ptr(out), B, H, W, cin, cout,
                                              kh, kw, sh, sw, ph, pw, dh, dw, act, ct, c0), "alsep_nn_conv2d")
            got = torch.from_numpy(host(out)).double()
            assert float((got[..., c0:c0 + cout] - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max())), f"split={split}"
            if ct != cout:                                      # nothing written outside the slice
                assert float(got[..., :c0].abs().max()) == 0.0 and float(got[..., c0 + cout:].abs().max()) == 0.0
            tiled = dev.launch_count("nn_conv2d_tiled_kernel") > 0             # (the tiled path's launch site counts under this name in both modes)
            assert (dev.launch_count("nn_conv2d_split_kernel") > 0) == (split and tiled)
            assert not dev.nn_range_exceeded()
    finally:
        dev.set_nn_contraction(False)


def test_split_contraction_raises_the_range_word(dev):
    """an operand beyond the half range (|x| > 65504) raises the context's range word in the split kernels, and only there"""
    g = torch.Generator().manual_seed(3)
    A, Bm = torch.randn(1, 1, 128, 64, generator=g), torch.randn(1, 1, 64, 64, generator=g)
    A[0, 0, 5, 7] = 1.0e5
    Ad, Bd, Cd = on(dev, A), on(dev, Bm), dev.empty((1, 1, 128, 64))
    args = (_lib.ptr(Ad), _lib.ptr(Bd), _lib.ptr(Cd), 1, 1, 128, 64, 64, _arr(128 * 64, 128 * 64, 64, 1), _arr(64 * 64, 64 * 64, 64, 1),
            _arr(128 * 64, 128 * 64, 64, 1), 1.0, None, 0)
    try:
        dev.set_nn_contraction(True)
        dev.check(dev.lib.alsep_nn_bgemm_bias(dev.handle, *args), "alsep_nn_bgemm_bias")
        assert dev.nn_range_exceeded() and not dev.nn_range_exceeded()          # read clears it
        dev.set_nn_contraction(False)
        dev.check(dev.lib.alsep_nn_bgemm_bias(dev.handle, *args), "alsep_nn_bgemm_bias")
        assert not dev.nn_range_exceeded()
        want = A[0, 0].double() @ Bm[0, 0].double().t()
        assert float((torch.from_numpy(host(Cd))[0, 0].double() - want).abs().max()) < 1e-1 * 2e-5 * float(want.abs().max()) + 2e-2
    finally:
        dev.set_nn_contraction(False)


def test_softmax_rows_with_leading_dimension(dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 804, generator=g) * 3
    want = torch.softmax(x[:, :801].double(), dim=-1).numpy()
    xd = on(dev, x.clone())                                     # in place (on the emulation ``on`` shares the host tensor's memory)
    dev.check(dev.lib.alsep_nn_softmax_rows_ld(dev.handle, _lib.ptr(xd), 37, 801, 804), "alsep_nn_softmax_rows_ld")
    got = host(xd)
    assert np.max(np.abs(got[:, :801] - want)) < 1e-6
    assert np.array_equal(got[:, 801:], x[:, 801:].numpy())     # the padding is left alone


@pytest.mark.parametrize("P,Cn,act", [(5000, 16, 3), (777, 128, 0), (3000, 96, 3), (1200, 384, 3)])
def test_instnorm_vs_torch(dev, P, Cn, act):
    """InstanceNorm2d (+ GELU) on channels-last data: channel counts that divide 256 take the all-threads statistics kernel, the
    others the thread-per-channel one"""
    g = torch.Generator().manual_seed(P + Cn)
    x = torch.randn(P, Cn, generator=g) * 2 + 0.3
    gamma, beta = 1 + 0.1 * torch.randn(Cn, generator=g), 0.1 * torch.randn(Cn, generator=g)
    want = torch.nn.functional.instance_norm(x.double().t()[None], weight=gamma.double(), bias=beta.double(), eps=1e-5)[0].t()
    if act == 3:
        want = _gelu(want)
    ws = dev.empty((int(dev.lib.alsep_nn_instnorm_workspace_bytes(P, Cn)),), torch.uint8)
    xd, y = on(dev, x), dev.empty((P, Cn))
    gd, bd = on(dev, gamma), on(dev, beta)
    dev.check(dev.lib.alsep_nn_instnorm(dev.handle, _lib.ptr(xd), _lib.ptr(y), _lib.ptr(gd), _lib.ptr(bd), P, Cn, 1e-5, act, _lib.ptr(ws)),
              "alsep_nn_instnorm")
    assert float((torch.from_numpy(host(y)).double() - want).abs().max()) < 2e-5


@pytest.mark.parametrize("rows,n", [(130, 60), (9, 1024), (5, 1500)])
def test_softmax_rows_sizes(dev, rows, n):
    """short rows (wave-per-row kernel, also a partial last workgroup), the longest such row, and a longer one (workgroup per row)"""
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, n, generator=g) * 4
    want = torch.softmax(x.double(), dim=-1).numpy()
    xd = on(dev, x.clone())
    dev.check(dev.lib.alsep_nn_softmax_rows(dev.handle, _lib.ptr(xd), rows, n), "alsep_nn_softmax_rows")
    assert np.max(np.abs(host(xd) - want)) < 1e-6
