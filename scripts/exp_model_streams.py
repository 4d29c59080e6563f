"""experiment: the four MDX models of the bench on four HIP streams (one alsep_ctx each) instead of one after the other"""
import os, sys, time, types, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from audiolab_amd import _lib
from audiolab_amd.mdx import Predictor
from audiolab_amd.synth import synth_mix, synthetic_state_dict
from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
dev = torch.device("cuda", 0)
cfg = TDFNetConfig()
mix = torch.from_numpy(synth_mix(300 * 44100)).to(dev)
sds = [synthetic_state_dict(cfg, seed=s) for s in range(4)]
pargs = types.SimpleNamespace(margin=44100, chunks=0, denoise=False, dim_f=cfg.dim_f, dim_t=8, n_fft=cfg.n_fft)
for mode in ("one", "four", "one", "four"):
    if mode == "one":
        streams = [None] * 4
        ctxs = [_lib.Context(dev)] * 4
    else:
        streams = [torch.cuda.Stream(device=dev) for _ in range(4)]
        ctxs = [_lib.Context(dev, stream=s.cuda_stream) for s in streams]
    nets = [TDFNet(cfg, sd, ctx=c, dtype=torch.bfloat16, max_batch=52) for sd, c in zip(sds, ctxs)]
    preds = [Predictor(pargs, n, ctx=c, max_batch=0) for n, c in zip(nets, ctxs)]

    def step():
        outs = []
        main = torch.cuda.current_stream(dev)
        for p, s in zip(preds, streams):
            if s is None:
                outs.append(p.demix(mix))
            else:
                s.wait_stream(main)
                with torch.cuda.stream(s):
                    outs.append(p.demix(mix))
        for s in streams:
            if s is not None:
                main.wait_stream(s)
        return outs
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    print(mode, "streams:", round((time.perf_counter() - t0) / 3 * 1e3, 2), "ms per step")
    del nets, preds
    torch.cuda.empty_cache()
