"""(neutral.hip aggressor codes: "neutral:<lds bytes>:<0 = 112 VGPRs | 1 = 256 VGPRs | 2 = no MFMA, VALU fma instead>"; victim "nv3" = a dense butterfly network in plain C++ that the compiler turns into packed f32 arithmetic, checked against an int32
model; victim "nv2" = the
instruction classes of the FFT kernels -- packed f32 arithmetic with op_sel / neg, scalar fma, the LDS access shapes -- self-checked.)

Discriminating experiments for the cross-stream corruption (VERDICT r3 "what's weak" 3).

Part A -- who is needed for the effect: aggressors {none, libalsep's f16 GEMM, libalsep's f16 convolution, a NEUTRAL f16 MFMA kernel at
256 VGPRs x 64 KiB LDS (two workgroups per CU), the same at 112 VGPRs, the same at 36 KiB} on one stream x victims {libalsep STFT
n_fft 8192 / 2048 / 4096, a NEUTRAL table-read + LDS round-trip kernel at 32 / 48 / 64 KiB} on another.  A victim launch is "bad" when its
result differs from its solo result (STFT) or when the kernel's own self-check counts a mismatch (neutral victim).

Part B -- what a corrupted frame looks like: corrupted STFT frames (n_fft 8192) are saved with their reference to gpurun_out/xstream_frames.npz
for the offline hypothesis test of scripts/dbg/analyse_frames.py (wrong twiddles? wrong input? wrong exchange?)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from audiolab_amd import _lib
if os.environ.get("DBG_LIB"):                                   # a differently compiled libalsep (e.g. fft.hip without packed-f32 SLP)
    _lib._LIB = _lib.bind(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["DBG_LIB"]))
from audiolab_amd._lib import Context
from audiolab_amd.mdx import StftPlan
from audiolab_amd.synth import synth_mix

HERE = os.path.dirname(os.path.abspath(__file__))
neu = C.CDLL(os.path.join(HERE, "libneutral.so"))
neu.nv_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
neu.nv_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
neu.na_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
neu.nv2_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
neu.nv3_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
neu.nv4_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
neu.nv5_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]

ctx0 = _lib.Context("cuda:0")
lib = ctx0.lib
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [Context(ctx0.device, stream=s.cuda_stream) for s in streams]
c = ctxs[0]
L = 261120
x = torch.from_numpy(synth_mix(L)).cuda()
plans = {n: StftPlan(c, n, 1024, n // 2 if n != 8192 else 4096, L // 1024 + 1) for n in (8192, 4096, 2048)}
ga = torch.randn(48060, 384, device="cuda").half(); gw = torch.randn(1536, 384, device="cuda").half()
gc = torch.empty(48060, 1536, device="cuda", dtype=torch.float16)
xl = torch.randn(256 * 1024, 128, device="cuda").half(); wl = (torch.randn(128, 3, 3, 128, device="cuda") / 34).half()
yl = torch.empty(256 * 1024, 128, device="cuda")
nA = torch.randn(48128, 384, device="cuda").half(); nB = torch.randn(1536, 384, device="cuda").half()
nC = torch.empty(48128, 1536, device="cuda")
NTAB = 8192
tab = torch.empty(NTAB, 2, dtype=torch.int32, device="cuda")
assert neu.nv_fill(C.c_void_p(0), C.c_void_p(tab.data_ptr()), NTAB) == 0
bad = torch.zeros(4, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()


def aggress(kind):
    cl = ctxs[1]
    st = C.c_void_p(streams[1].cuda_stream)
    for _ in range(8):
        if kind == "lib_gemm_hh":
            cl.check(lib.alsep_nn_gemm_f16(cl.handle, _lib.ptr(ga), 384, 0, _lib.ptr(gw), 384, 0, _lib.ptr(gc), 1, 1536, 0, None, 0, None, 0, 0, 1,
                                           48060, 1536, 384, 1.0, 0, None), "g")
        elif kind == "lib_conv_hh":
            cl.check(lib.alsep_nn_conv2d_f16(cl.handle, _lib.ptr(xl), _lib.ptr(wl), _lib.ptr(yl), None, 128, 1, 256, 1024, 128, 128, 3, 3, 1, 1, 1, 1,
                                             128, 0, None, 0), "l")
        elif kind.startswith("neutral"):
            _, lds, v256 = kind.split(":")
            rc = neu.na_launch(st, C.c_void_p(nA.data_ptr()), C.c_void_p(nB.data_ptr()), C.c_void_p(nC.data_ptr()), 48128, 1536, 384, int(lds), int(v256))
            assert rc == 0, rc


def victim(kind):
    """-> (result tensor or None, self-check counts or None)"""
    if kind.startswith("stft"):
        return plans[int(kind[4:])].stft_strided(x, L, 2 * L, 1, torch.float32, _lib.LAYOUT_REF), None
    if kind in ("nv3", "nv4", "nv5"):
        bad.zero_()
        rc = {"nv3": neu.nv3_launch, "nv4": neu.nv4_launch, "nv5": neu.nv5_launch}[kind](C.c_void_p(streams[0].cuda_stream), 2048, 64, C.c_void_p(bad.data_ptr()))
        assert rc == 0, rc
        return None, bad
    if kind == "nv2":
        bad.zero_()
        rc = neu.nv2_launch(C.c_void_p(streams[0].cuda_stream), 1024, 24, C.c_void_p(bad.data_ptr()))
        assert rc == 0, rc
        return None, bad
    _, lds = kind.split(":")
    bad.zero_()
    rc = neu.nv_launch(C.c_void_p(streams[0].cuda_stream), 1024, C.c_void_p(tab.data_ptr()), NTAB, int(lds), 24, C.c_void_p(bad.data_ptr()))
    assert rc == 0, rc
    return None, bad


AGGRESSORS = ["none", "lib_gemm_hh", "lib_conv_hh", "neutral:65536:1", "neutral:65536:0", "neutral:36864:1", "neutral:65536:2"]
VICTIMS = ["stft8192", "stft2048", "stft4096", "nvictim:32768", "nvictim:65536", "nv2", "nv3", "nv4", "nv5"]
if os.environ.get("XS_VICTIMS"):
    VICTIMS = os.environ["XS_VICTIMS"].split(",")
if os.environ.get("XS_AGGRESSORS"):
    AGGRESSORS = os.environ["XS_AGGRESSORS"].split(",")
REPS = int(os.environ.get("XS_REPS", "10"))
print("== part A: corrupted victim launches out of", REPS, "(victim on stream 0, aggressor on stream 1)", flush=True)
refs = {}
for v in VICTIMS:
    if v.startswith("stft"):
        with torch.cuda.stream(streams[0]):
            refs[v] = victim(v)[0].clone()
torch.cuda.synchronize()
saved = []
for a in AGGRESSORS:
    row = []
    for v in VICTIMS:
        nbad = 0
        detail = ""
        for rep in range(REPS):
            if a != "none":
                with torch.cuda.stream(streams[1]):
                    aggress(a)
            with torch.cuda.stream(streams[0]):
                got, counts = victim(v)
            torch.cuda.synchronize()
            if got is not None:
                d = (got - refs[v]).abs()[0]
                if bool((d > 0).any()):
                    nbad += 1
                    if v == "stft8192" and len(saved) < 6:
                        frames = (d > 0).any(0).any(0).nonzero().flatten().tolist()
                        for t in frames[:3]:
                            saved.append((a, t, got[0, :, :, t].cpu().numpy(), refs[v][0, :, :, t].cpu().numpy()))
            else:
                cnt = counts.cpu().tolist()
                if cnt[0] or cnt[1] or (v == "nv2" and cnt[2]):
                    nbad += 1
                    detail = (f" (wrong components {cnt[0]}, workgroups {cnt[1]})" if v in ("nv3", "nv4", "nv5") else
                              f" (packed-f32 {cnt[0]}, scalar fma {cnt[1]}, LDS {cnt[2]}, workgroups {cnt[3]})" if v == "nv2" else
                              f" (table-in-register mismatches {cnt[0]}, after-LDS mismatches {cnt[1]}, workgroups {cnt[2]})")
        row.append(f"{v}: {nbad}{detail}")
    print(f"aggressor {a:18s} | " + " | ".join(row), flush=True)

out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "xstream_frames.npz")
os.makedirs(os.path.dirname(out), exist_ok=True)
if saved:
    np.savez_compressed(out, aggressor=np.array([s[0] for s in saved]), frame=np.array([s[1] for s in saved]),
                        got=np.stack([s[2] for s in saved]), ref=np.stack([s[3] for s in saved]), L=L)
    print("== part B: saved", len(saved), "corrupted frames to", out, flush=True)
else:
    print("== part B: no corrupted STFT frame seen", flush=True)
