"""host-side cost of the chunked Roformer runner (half precision, graph replay): where the host spends a track, and whether it ever waits
for the GPU -- launches are asynchronous, so a call that shows up with milliseconds per call is one that blocks"""
import cProfile, os, pstats, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
from audiolab_amd.engine import Separator
name = sys.argv[1] if len(sys.argv) > 1 else "vocals_mel_band_roformer.ckpt"
ctx = _lib.Context("cuda:0")
eng = Separator(ctx=ctx, use_autocast=True, allow_synthetic=True)
eng.load_model(name)
runner = eng.model_instance.roformer
mix = torch.randn(2, 120 * 44100, device="cuda") * 0.1
for _ in range(2):
    runner.demix(mix)
torch.cuda.synchronize()
t0 = time.perf_counter()
out = runner.demix(mix)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host time of one track %.1f ms, then %.1f ms until the GPU is done" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
out = runner.demix(mix)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
