"""TEST-ONLY: AddressSanitizer pass over the round-4 kernels on the emulation's ASan build: the persistent half-precision GEMM (both tile
heights, ragged M / N / K, batches, residual), the 64-key attention (one and two query blocks per wave, ragged lengths, both layouts), the
split-half generic GEMM / convolution (one small HTDemucs forward with the contraction switched on) and the split-half TFC-TDF network.
    LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tests/cpu_emul/asan_cases_half.py"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiolab_amd import _lib  # noqa: E402

_lib._LIB = _lib.bind(os.path.join(ROOT, "tests", "cpu_emul", "libalsep_emul_asan.so"))
_lib.DEVICE_TYPE = "cpu"
ctx = _lib.Context("cpu")
lib, h = ctx.lib, ctx.handle
g = torch.Generator().manual_seed(0)
# persistent GEMM: (M, N, K, batches, act, residual, half out); 2049 x 384 picks the 192-row tiles (33 against 27 of 256: fewer idle rows)
for M, N, K, nb, act, res, c16 in ((2100, 136, 200, 1, 3, False, True), (2049, 384, 64, 1, 0, True, False), (2304, 72, 128, 2, 5, False, False),
                                   (2050, 200, 72, 3, 0, True, True), (4000, 128, 64, 1, 0, False, True)):
    lda, ldc = K + 8, N + 8
    a = torch.randn(nb, M, lda, generator=g).half(); w = (torch.randn(nb, N, K, generator=g) / K ** 0.5).half()
    bias = torch.randn(nb, N, generator=g); r = torch.randn(nb, M, ldc, generator=g)
    c = torch.zeros((nb, M, ldc), dtype=torch.float16 if c16 else torch.float32)
    ctx.launch_counts_reset()
    ctx.check(lib.alsep_nn_gemm_f16(h, _lib.ptr(a), lda, M * lda, _lib.ptr(w), K, N * K, _lib.ptr(c), 1 if c16 else 0, ldc, M * ldc, _lib.ptr(bias), N,
                                    _lib.ptr(r) if res else None, ldc, M * ldc, nb, M, N, K, 0.5, act, None), "gemm")
    assert ctx.launch_count("nn_gemm_h2_kernel") == 1 and torch.isfinite(c.float()).all()
    print("gemm_h2", M, N, K, nb, "ok", flush=True)
# attention: (over_time, L, sequences)
for over_time, L, n_seq in ((True, 301, 2), (True, 70, 3), (False, 33, 5), (False, 130, 2), (True, 1, 2)):
    heads, d = 2, 64
    inner, rows, ld = heads * d, n_seq * L, 3 * heads * d
    qkv = torch.randn(rows, ld, generator=g).half(); out = torch.zeros(rows, inner, dtype=torch.float16)
    if over_time: ss, rs, os_, or_ = ld, n_seq * ld, inner, n_seq * inner
    else: ss, rs, os_, or_ = L * ld, ld, L * inner, inner
    table = torch.zeros(L, d // 2, 2); ctx.check(lib.alsep_nn_rotary_table(h, _lib.ptr(table), L, d), "table")
    gates = torch.randn(rows, heads, generator=g)
    ctx.check(lib.alsep_nn_attention_f16(h, _lib.ptr(qkv), _lib.ptr(out), n_seq, L, heads, d, ss, rs, os_, or_, 0.125, _lib.ptr(table), _lib.ptr(gates),
                                         heads if over_time else L * heads, n_seq * heads if over_time else heads), "attention")
    assert torch.isfinite(out.float()).all()
    print("attention", over_time, L, n_seq, "ok", flush=True)
# RMSNorm: the four-rows-per-wave kernel (rows % 16 != 0: the last wave's rows beyond the end read the last row) and the generic one
for rows, C in ((1001, 384), (130, 512), (3, 384), (37, 96)):
    x = torch.randn(rows, C, generator=g); gm = torch.ones(C); y = torch.zeros(rows, C, dtype=torch.float16)
    ctx.check(lib.alsep_nn_rmsnorm_f16(h, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gm), rows, C, C, C), "rmsnorm")
    assert torch.isfinite(y.float()).all()
    print("rmsnorm", rows, C, "ok", flush=True)
# the persistent 7680 fused front end: 528 frames over 512 workgroups (16 of them walk two frames)
from audiolab_amd.mdx import StftPlan  # noqa: E402
from audiolab_amd.synth import synthetic_state_dict as _ssd  # noqa: E402
from audiolab_amd.tdfnet import TDFNet as _TDFNet, TDFNetConfig as _Cfg  # noqa: E402
c7 = _Cfg(dim_f=80, dim_t=8, n_fft=7680, hop=1024, num_blocks=1, g=48, bn=8)
n7 = _TDFNet(c7, _ssd(c7, seed=2, calib="noise"), ctx=ctx, dtype=torch.float16, max_batch=66)
p7 = StftPlan(ctx, c7.n_fft, c7.hop, c7.dim_f, c7.dim_t)
tot7 = 65 * 1500 + p7.chunk_size + 17
out7 = n7.forward_pcm(p7, torch.randn(2 * tot7, generator=g), tot7, 1500, 66)          # flat buffer: channel stride tot7, chunk stride 1500
assert out7 is not None and torch.isfinite(out7.float()).all()
print("fused 7680 persistent ok", flush=True)
# split-half generic GEMM / convolution: a small HTDemucs forward with the contraction on
from audiolab_amd.htdemucs import HTDemucs, HTDemucsConfig  # noqa: E402
from oracle import htdemucs_oracle as ho  # noqa: E402
ocfg = ho.HTDemucsConfig(sources=("drums", "bass", "other"), channels=16, nfft=256, depth=2, dconv_comp=4, bottom_channels=32, t_layers=3,
                         t_heads=4, segment_samples=2560, samplerate=4000)
net = HTDemucs(HTDemucsConfig(**dataclasses.asdict(ocfg)), ho.synthetic_state_dict(ocfg, 1), ctx=ctx)
ctx.set_nn_contraction(True)
ctx.launch_counts_reset()
y = net.forward(torch.randn(2, ocfg.segment_samples) * 0.2)
ctx.set_nn_contraction(False)
assert torch.isfinite(y).all()
print("htdemucs split", {n: ctx.launch_count(n) for n in ("nn_gemm_split_kernel", "nn_conv2d_split_kernel") if ctx.launch_count(n)}, flush=True)
# split-half TFC-TDF network (float32 storage)
from audiolab_amd.synth import synthetic_state_dict  # noqa: E402
from audiolab_amd.tdfnet import TDFNet, TDFNetConfig  # noqa: E402
cfg = TDFNetConfig(dim_f=96, dim_t=16, n_fft=256, hop=64, num_blocks=3, g=48, bn=8)
tn = TDFNet(cfg, synthetic_state_dict(cfg, seed=1, calib="noise"), ctx=ctx, dtype=torch.float32, max_batch=2)
x = torch.randn(2, cfg.dim_t, cfg.dim_f, 4) * 0.1
ctx.launch_counts_reset()
y = tn.forward_nhwc(x)
assert torch.isfinite(y).all()
print("tdfnet split", {n: ctx.launch_count(n) for n in ("conv3x3_f32s_kernel", "tdf_gemm_f32s_kernel", "pix_gemm_f32s_kernel") if ctx.launch_count(n)}, flush=True)
print("half / split asan ok")
