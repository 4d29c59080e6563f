"""Micro-benchmark of the half-precision kernels of csrc/nn_half.hip at the Mel-Band Roformer's shapes (one 8 s chunk: 801 frames x 60 bands)."""
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


for M, N, K, nb, c16, tag in ((48060, 1536, 384, 1, 1, "qkv / ffn l1 (half out)"), (48060, 384, 1536, 1, 0, "ffn l2 (+ residual)"),
                              (48060, 384, 512, 1, 0, "attn out (+ residual)"), (48060, 8, 384, 1, 0, "gates"),
                              (801, 1536, 384, 60, 1, "mask layer 1 (batched over bands)"), (801, 1040, 1536, 60, 0, "mask layer 2 (batched, padded)"),
                              (801, 384, 520, 60, 0, "band split (batched, padded)")):
    a = torch.randn(nb, M, K, device="cuda").half()
    w = torch.randn(nb, N, K, device="cuda").half()
    c = torch.empty(nb, M, N, device="cuda", dtype=torch.float16 if c16 else torch.float32)
    bias = torch.randn(nb, N, device="cuda")
    res = torch.randn(nb, M, N, device="cuda") if "residual" in tag else None
    dt = timed(lambda: ctx.check(lib.alsep_nn_gemm_f16(h, _lib.ptr(a), K, M * K, _lib.ptr(w), K, N * K, _lib.ptr(c), c16, N, M * N, _lib.ptr(bias), N,
                                                       _lib.ptr(res) if res is not None else None, N, M * N, nb, M, N, K, 1.0, 0, None), "gemm"))
    fl = 2.0 * nb * M * N * K
    by = nb * (2.0 * M * K + 2.0 * N * K + (2 if c16 else 4) * M * N + (4.0 * M * N if res is not None else 0))
    print(f"gemm_hh {tag:34s} M {M:6d} N {N:5d} K {K:5d} nb {nb:3d}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:7.1f} TFLOP/s ({fl / dt / 2.5e15 * 100:4.1f} % of 2.5 PF)  "
          f"{by / dt / 1e9:7.0f} GB/s")
x = torch.randn(48060, 384, device="cuda"); gm = torch.ones(384, device="cuda"); y = torch.empty(48060, 384, device="cuda", dtype=torch.float16)
dt = timed(lambda: ctx.check(lib.alsep_nn_rmsnorm_f16(h, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gm), 48060, 384, 384, 384), "rms"))
print(f"rmsnorm_f16 48060 x 384: {dt * 1e6:8.1f} us  {48060 * 384 * 6 / dt / 1e9:7.0f} GB/s")
for over_time, n_seq, L in ((True, 60, 801), (False, 801, 60)):
    heads, d = 8, 64
    inner = heads * d
    rows = n_seq * L
    qkv = torch.randn(rows, 3 * inner, device="cuda").half()
    out = torch.empty(rows, inner, device="cuda", dtype=torch.float16)
    ld = 3 * inner
    if over_time:
        ss, rs, os_, or_ = ld, n_seq * ld, inner, n_seq * inner
    else:
        ss, rs, os_, or_ = L * ld, ld, L * inner, inner
    table = torch.zeros(L, d // 2, 2, device="cuda")
    ctx.check(lib.alsep_nn_rotary_table(h, _lib.ptr(table), L, d), "table")
    gates = torch.randn(rows, heads, device="cuda")
    dt = timed(lambda: ctx.check(lib.alsep_nn_attention_f16(h, _lib.ptr(qkv), _lib.ptr(out), n_seq, L, heads, d, ss, rs, os_, or_, 0.125, _lib.ptr(table),
                                                            _lib.ptr(gates), heads if over_time else L * heads, n_seq * heads if over_time else heads), "attn"))
    fl = 4.0 * n_seq * heads * L * L * d
    print(f"attention {'time' if over_time else 'freq'}: {n_seq} x {heads} sequences of {L}: {dt * 1e6:8.1f} us  {fl / dt / 1e12:6.1f} TFLOP/s")
