"""a chunked-runner model in half precision, 120 s: libalsep built with the SLP vectoriser (DBG_LIB=libalsep_slp.so) against the product build"""
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
if os.environ.get("DBG_LIB"):
    _lib._LIB = _lib.bind(os.path.join(os.path.dirname(os.path.abspath(__file__)), os.environ["DBG_LIB"]))
from audiolab_amd.engine import Separator
name = sys.argv[1] if len(sys.argv) > 1 else "MDX23C-8KFFT-InstVoc_HQ.ckpt"
ctx = _lib.Context("cuda:0")
eng = Separator(ctx=ctx, use_autocast=True, allow_synthetic=True)
eng.load_model(name)
runner = eng.model_instance.roformer
mix = torch.randn(2, 120 * 44100, device="cuda") * 0.1
for _ in range(2):
    runner.demix(mix)
torch.cuda.synchronize()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); runner.demix(mix); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print(f"{os.environ.get('DBG_LIB', 'product build (no SLP)'):28s} {name}: {min(ts) * 1e3:.1f} ms per 120 s track", flush=True)
