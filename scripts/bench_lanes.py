import sys, time, torch, numpy as np, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
if os.environ.get("DBG_LIB"):
    _lib._LIB = _lib.bind(os.environ["DBG_LIB"])
from audiolab_amd.engine import Separator
from audiolab_amd.synth import synth_mix
ctx = _lib.Context("cuda:0")
half = "--half" in sys.argv
eng = Separator(ctx=ctx, use_autocast=half, allow_synthetic=True)
mix = torch.from_numpy(synth_mix(44100 * 120)).cuda()
for name in [a for a in sys.argv[1:] if a != "--half"]:
    eng.load_model(name)
    eng.separate_array(mix); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.separate_array(mix); torch.cuda.synchronize()
    print(name, "half" if half else "fp32", "lanes", os.environ.get("ALSEP_RUNNER_LANES"), round((time.perf_counter() - t0) * 1e3, 1), "ms for 120 s")
