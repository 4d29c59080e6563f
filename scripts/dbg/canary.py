import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from audiolab_amd import _lib
from audiolab_amd._lib import Context
ctx0 = _lib.Context("cuda:0")
lib = ctx0.lib
can = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libcanary.so"))
can.canary_launch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
streams = [torch.cuda.Stream() for _ in range(2)]
ctxs = [Context(ctx0.device, stream=s.cuda_stream) for s in streams]
xl = torch.randn(256 * 1024, 128, device="cuda").half(); wl = (torch.randn(128, 3, 3, 128, device="cuda") / 34).half(); yl = torch.empty(256 * 1024, 128, device="cuda")
ga = torch.randn(48060, 384, device="cuda").half(); gw = torch.randn(1536, 384, device="cuda").half(); gc = torch.empty(48060, 1536, device="cuda", dtype=torch.float16)
bad = torch.zeros(1, dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
def load(kind, n):
    c = ctxs[1]
    for _ in range(n):
        if kind == "conv_hh":
            c.check(lib.alsep_nn_conv2d_f16(c.handle, _lib.ptr(xl), _lib.ptr(wl), _lib.ptr(yl), None, 128, 1, 256, 1024, 128, 128, 3, 3, 1, 1, 1, 1, 128, 0, None, 0), "l")
        elif kind == "gemm_hh":
            c.check(lib.alsep_nn_gemm_f16(c.handle, _lib.ptr(ga), 384, 0, _lib.ptr(gw), 384, 0, _lib.ptr(gc), 1, 1536, 0, None, 0, None, 0, 0, 1, 48060, 1536, 384, 1.0, 0, None), "g")
for kind in ("none", "conv_hh", "gemm_hh"):
    for lds in (65536, 32768, 16384):
        bad.zero_()
        torch.cuda.synchronize()
        for rep in range(4):
            with torch.cuda.stream(streams[1]):
                load(kind, 4)
            with torch.cuda.stream(streams[0]):
                rc = can.canary_launch(C.c_void_p(streams[0].cuda_stream), 1024, lds, 40, C.c_void_p(bad.data_ptr()))
                assert rc == 0, rc
            torch.cuda.synchronize()
        print("co-runner", kind, "canary LDS", lds, "corrupted words seen:", int(bad.item()), flush=True)
