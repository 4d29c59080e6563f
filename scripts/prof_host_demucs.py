"""host-side cost of one htdemucs_6s segment forward (launches are asynchronous: what cProfile sees is Python + driver time)"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
from audiolab_amd.engine import Separator
ctx = _lib.Context("cuda:0")
eng = Separator(ctx=ctx, use_autocast=False, allow_synthetic=True)
eng.load_model("htdemucs_6s.yaml")
net = eng.model_instance.demucs.net
x = torch.randn(2, net.cfg.segment_samples, device="cuda") * 0.2
for _ in range(3): net.forward(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): net.forward(x)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host enqueue per forward %.2f ms, + drain %.2f ms" % ((t1 - t0) * 100, (t2 - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(10): net.forward(x)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
