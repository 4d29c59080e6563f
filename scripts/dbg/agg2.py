"""Reproduction of the cross-stream interference that keeps half-precision runners on one lane (roformer.RoformerRunner): the f16 GEMM on one
HIP stream, the STFT (n_fft 8192) on another; the STFT is a pure function of constant inputs, yet 2-7 of 16 launches come back with whole
frames wrong.  scripts/dbg/victims.py lists which kernels are affected (the FFT kernels; not InstanceNorm, the fp32 convolution, rocBLAS),
scripts/dbg/canary.py shows that LDS beside the GEMM stays intact.  DBG_LIB=<other libalsep.so> runs a differently compiled library
(one workgroup per CU for the GEMM: no corruption)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from audiolab_amd import _lib
if os.environ.get("DBG_LIB"):
    _lib._LIB = _lib.bind(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["DBG_LIB"]))
from audiolab_amd._lib import Context
from audiolab_amd.mdx import StftPlan
from audiolab_amd.synth import synth_mix
ctx0 = _lib.Context("cuda:0")
lib = ctx0.lib
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [Context(ctx0.device, stream=s.cuda_stream) for s in streams]
c = ctxs[0]
L = 261120
x = torch.from_numpy(synth_mix(L)).cuda()
plan = StftPlan(c, 8192, 1024, 4096, 256)
ga = torch.randn(48060, 384, device="cuda").half(); gw = torch.randn(1536, 384, device="cuda").half(); gc = torch.empty(48060, 1536, device="cuda", dtype=torch.float16)
torch.cuda.synchronize()
with torch.cuda.stream(streams[0]):
    ref = plan.stft_strided(x, L, 2 * L, 1, torch.float32, _lib.LAYOUT_REF).clone()
torch.cuda.synchronize()
worst, nbad = 0.0, 0
for rep in range(16):
    with torch.cuda.stream(streams[1]):
        cl = ctxs[1]
        for _ in range(10):
            cl.check(lib.alsep_nn_gemm_f16(cl.handle, _lib.ptr(ga), 384, 0, _lib.ptr(gw), 384, 0, _lib.ptr(gc), 1, 1536, 0, None, 0, None, 0, 0, 1, 48060, 1536, 384, 1.0, 0, None), "g")
    with torch.cuda.stream(streams[0]):
        got = plan.stft_strided(x, L, 2 * L, 1, torch.float32, _lib.LAYOUT_REF)
    torch.cuda.synchronize()
    d = float((got - ref).abs().max()); worst = max(worst, d); nbad += d > 0
print("lib", os.environ.get("DBG_LIB", "product"), "| gemm_hh aggressor: STFT corrupted in", nbad, "of 16 reps, worst", worst, flush=True)
