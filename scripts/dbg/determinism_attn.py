import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
ctx = _lib.Context("cuda:0")
lib, h = ctx.lib, ctx.handle
for over_time, n_seq, L, rot_on in ((True, 60, 801, 1), (True, 60, 801, 0), (True, 60, 128, 1), (True, 60, 64, 1), (True, 2, 801, 1), (False, 801, 60, 1), (False, 100, 200, 1)):
    heads, d = 8, 64; inner = heads * d; rows = n_seq * L
    qkv = torch.randn(rows, 3 * inner, device="cuda").half(); out = torch.empty(rows, inner, device="cuda", dtype=torch.float16); ld = 3 * inner
    if over_time: ss, rs, os_, or_ = ld, n_seq * ld, inner, n_seq * inner
    else: ss, rs, os_, or_ = L * ld, ld, L * inner, inner
    table = torch.zeros(L, d // 2, 2, device="cuda"); ctx.check(lib.alsep_nn_rotary_table(h, _lib.ptr(table), L, d), "table")
    fn = lambda: ctx.check(lib.alsep_nn_attention_f16(h, _lib.ptr(qkv), _lib.ptr(out), n_seq, L, heads, d, ss, rs, os_, or_, 0.125, _lib.ptr(table) if rot_on else None,
                                             None, 0, 0), "attn")
    fn(); torch.cuda.synchronize(); ref = out.clone(); bad = 0; worst = 0.0
    # float64 reference of a few (seq, head)
    for _ in range(6):
        out.zero_(); fn(); torch.cuda.synchronize()
        if not torch.equal(ref, out):
            bad += 1; worst = max(worst, float((ref.float() - out.float()).abs().max()))
    nz = int((ref == 0).all(dim=1).sum())
    print(f"QB={os.environ.get('ALSEP_ATTN_QB', 'auto')} over_time {over_time} n_seq {n_seq} L {L} rot {rot_on}: {'deterministic' if bad == 0 else 'DIFFERS %d/6 max %.3g' % (bad, worst)}; all-zero rows {nz}", flush=True)
