"""run-to-run determinism of the half-precision kernels at the Mel-Band shapes: every op 12 times on the same operands, outputs compared bit for bit"""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
ctx = _lib.Context("cuda:0")
lib, h = ctx.lib, ctx.handle
def rep(name, fn, out, n=12):
    fn(); torch.cuda.synchronize(); ref = out.clone(); bad = 0; worst = 0.0
    for _ in range(n):
        out.zero_(); fn(); torch.cuda.synchronize()
        if not torch.equal(ref, out):
            bad += 1; worst = max(worst, float((ref.float() - out.float()).abs().max()))
    print(f"{name:60s} {'deterministic' if bad == 0 else 'DIFFERS in %d of %d runs, max |delta| %.3g (peak %.3g)' % (bad, n, worst, float(ref.float().abs().max()))}", flush=True)
for M, N, K, nb, c16, act, res, bias_on, tag in ((48060, 1536, 384, 1, 1, 0, 0, 0, "qkv"), (48060, 1536, 384, 1, 1, 3, 0, 1, "ffn l1 gelu"), (48060, 384, 1536, 1, 0, 0, 1, 1, "ffn l2 + res"),
                                             (48060, 384, 512, 1, 0, 0, 1, 0, "attn out + res"), (48060, 8, 384, 1, 0, 0, 0, 1, "gates"), (801, 1536, 384, 60, 1, 5, 0, 1, "mask l1")):
    a = torch.randn(nb, M, K, device="cuda").half(); w = (torch.randn(nb, N, K, device="cuda") / K ** 0.5).half()
    c = torch.empty(nb, M, N, device="cuda", dtype=torch.float16 if c16 else torch.float32)
    bias = torch.randn(nb, N, device="cuda"); r = torch.randn(nb, M, N, device="cuda")
    rep(f"gemm {tag} M {M} N {N} K {K}", lambda: ctx.check(lib.alsep_nn_gemm_f16(h, _lib.ptr(a), K, M * K, _lib.ptr(w), K, N * K, _lib.ptr(c), c16, N, M * N, _lib.ptr(bias) if bias_on else None, N,
                                                   _lib.ptr(r) if res else None, N, M * N, nb, M, N, K, 1.0, act, None), "gemm"), c)
x = torch.randn(48060, 384, device="cuda"); gm = torch.ones(384, device="cuda"); y = torch.empty(48060, 384, device="cuda", dtype=torch.float16)
rep("rmsnorm", lambda: ctx.check(lib.alsep_nn_rmsnorm_f16(h, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gm), 48060, 384, 384, 384), "rms"), y)
for over_time, n_seq, L in ((True, 60, 801), (False, 801, 60)):
    heads, d = 8, 64; inner = heads * d; rows = n_seq * L
    qkv = torch.randn(rows, 3 * inner, device="cuda").half(); out = torch.empty(rows, inner, device="cuda", dtype=torch.float16); ld = 3 * inner
    if over_time: ss, rs, os_, or_ = ld, n_seq * ld, inner, n_seq * inner
    else: ss, rs, os_, or_ = L * ld, ld, L * inner, inner
    table = torch.zeros(L, d // 2, 2, device="cuda"); ctx.check(lib.alsep_nn_rotary_table(h, _lib.ptr(table), L, d), "table")
    gates = torch.randn(rows, heads, device="cuda")
    rep(f"attention {'time' if over_time else 'freq'} L {L}", lambda: ctx.check(lib.alsep_nn_attention_f16(h, _lib.ptr(qkv), _lib.ptr(out), n_seq, L, heads, d, ss, rs, os_, or_, 0.125, _lib.ptr(table),
                                             _lib.ptr(gates), heads if over_time else L * heads, n_seq * heads if over_time else heads), "attn"), out)
