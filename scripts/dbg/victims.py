import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from audiolab_amd import _lib
from audiolab_amd._lib import Context
from audiolab_amd.mdx import StftPlan
from audiolab_amd.synth import synth_mix
ctx0 = _lib.Context("cuda:0")
lib = ctx0.lib
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [Context(ctx0.device, stream=s.cuda_stream) for s in streams]
c = ctxs[0]
xl = torch.randn(256 * 1024, 128, device="cuda").half(); wl = (torch.randn(128, 3, 3, 128, device="cuda") / 34).half(); yl = torch.empty(256 * 1024, 128, device="cuda")
L = 261120
x = torch.from_numpy(synth_mix(L)).cuda()
plans = {n: StftPlan(c, n, 1024, n // 2 if n != 8192 else 4096, L // 1024 + 1) for n in (8192, 6144, 4096, 2048)}
xin = torch.randn(65536, 128, device="cuda"); g = torch.ones(128, device="cuda"); b = torch.zeros(128, device="cuda")
ws = torch.empty(int(lib.alsep_nn_instnorm_workspace_bytes(65536, 128)), dtype=torch.uint8, device="cuda")
wf = torch.randn(3, 3, 128, 128, device="cuda") / 34; sc = torch.ones(128, device="cuda"); sh = torch.zeros(128, device="cuda")
A = torch.randn(4096, 1024, device="cuda"); B = torch.randn(1024, 1024, device="cuda")
torch.cuda.synchronize()

def victim(kind):
    if kind.startswith("stft"):
        n = int(kind[4:])
        return plans[n].stft_strided(x, L, 2 * L, 1, torch.float32, _lib.LAYOUT_REF).clone()
    if kind == "instnorm":
        y = torch.empty(65536, 128, device="cuda")
        c.check(lib.alsep_nn_instnorm(c.handle, _lib.ptr(xin), _lib.ptr(y), _lib.ptr(g), _lib.ptr(b), 65536, 128, 1e-5, 3, _lib.ptr(ws)), "i")
        return y
    if kind == "conv_f32":
        y = torch.empty(65536, 128, device="cuda")
        c.check(lib.alsep_nn_conv2d(c.handle, _lib.ptr(xin), _lib.ptr(wf), _lib.ptr(sc), _lib.ptr(sh), _lib.ptr(y), 1, 256, 256, 128, 128, 3, 3, 1, 1, 1, 1, 1, 1, 0, 128, 0), "c")
        return y
    if kind == "torch_mm":
        return torch.mm(A, B)
    if kind == "istft8192":
        spec = plans[8192].stft_strided(x, L, 2 * L, 1, torch.float32, _lib.LAYOUT_REF)
        out = torch.empty(1, 2, L, device="cuda")
        plans[8192].istft_strided(spec, _lib.LAYOUT_REF, out, L, 2 * L, 0, L, L)
        return out

import ctypes as C
can = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcanary.so"))
can.canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
bad = torch.zeros(1, dtype=torch.int64, device="cuda")
MODE = {"m": "conv_hh"}
def load(n):
    cl = ctxs[1]
    if MODE["m"].startswith("canary"):
        can.canary_launch(C.c_void_p(streams[1].cuda_stream), 4096, int(MODE["m"][6:]), 60, C.c_void_p(bad.data_ptr()))
        return
    for _ in range(n):
        cl.check(lib.alsep_nn_conv2d_f16(cl.handle, _lib.ptr(xl), _lib.ptr(wl), _lib.ptr(yl), None, 128, 1, 256, 1024, 128, 128, 3, 3, 1, 1, 1, 1, 128, 0, None, 0), "l")

MODE["m"] = "conv_hh"
for kind in ("stft8192", "stft4096"):
    with torch.cuda.stream(streams[0]):
        ref = victim(kind)
    torch.cuda.synchronize()
    for rep in range(5):
        with torch.cuda.stream(streams[1]):
            load(5)
        with torch.cuda.stream(streams[0]):
            got = victim(kind)
        torch.cuda.synchronize()
        d = (got - ref).abs()[0]            # [4, F, T]
        badmask = d > 0
        nb = int(badmask.sum())
        if nb:
            idx = badmask.nonzero()
            ts = idx[:, 2].unique().tolist(); ks = idx[:, 1]; cs = idx[:, 0].unique().tolist()
            print(kind, "rep", rep, "bad elements", nb, "of", d.numel(), "| frames:", ts[:20], "n_frames", len(ts), "| bins min/max", int(ks.min()), int(ks.max()), "| planes", cs, flush=True)
        else:
            print(kind, "rep", rep, "clean", flush=True)
