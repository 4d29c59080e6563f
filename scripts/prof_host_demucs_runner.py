"""htdemucs_6s, 10 min: host time of DemucsRunner.separate (launches are asynchronous) against the time until the GPU is done, per lane count"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
from audiolab_amd.engine import Separator
from audiolab_amd.htdemucs import DemucsRunner
ctx = _lib.Context("cuda:0")
eng = Separator(ctx=ctx, use_autocast=False, allow_synthetic=True)
eng.load_model("htdemucs_6s.yaml")
net = eng.model_instance.demucs.net
mix = torch.randn(2, 600 * 44100, device="cuda") * 0.1
for lanes in (1, 2, 4, 8):
    r = DemucsRunner(net, shifts=2, overlap=0.25, seed=0, lanes=lanes)
    r.separate(mix); torch.cuda.synchronize()
    t0 = time.perf_counter(); r.separate(mix); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"lanes {lanes}: host {1e3 * (t1 - t0):.0f} ms, GPU done after {1e3 * (t2 - t0):.0f} ms", flush=True)
r = DemucsRunner(net, shifts=2, overlap=0.25, seed=0, lanes=4)
r.separate(mix); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); r.separate(mix); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
