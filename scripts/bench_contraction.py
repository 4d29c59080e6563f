"""float32 model families with their contractions as exact f32 MFMA (four runner lanes) vs split-half products on the f16 pipe (csrc/nn_f32s.h;
one lane: 16-bit MFMA kernels must not share the GPU with other lanes' FFT launches), one GPU:
htdemucs_6s (10 min), Mel-Band Roformer, BS Roformer and MDX23C in their float32 modes (120 s); seconds per track, same stems check."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from audiolab_amd import _lib
from audiolab_amd.engine import Separator
from audiolab_amd.synth import synth_mix
ctx = _lib.Context("cuda:0")
cases = [("htdemucs_6s.yaml", 600), ("vocals_mel_band_roformer.ckpt", 120), ("model_bs_roformer_ep_368_sdr_12.9628.ckpt", 120), ("MDX23C-8KFFT-InstVoc_HQ.ckpt", 120)]
for name, secs in cases:
    mix = torch.from_numpy(synth_mix(secs * 44100)).cuda()
    outs = {}
    for mode in ("exact", "split"):
        eng = Separator(ctx=ctx, use_autocast=False, allow_synthetic=True, nn_contraction=mode)
        eng.load_model(name)
        eng.separate_array(mix[:, : 30 * 44100])
        torch.cuda.synchronize()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            out = eng.separate_array(mix)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        outs[mode] = out
        print(f"{name:44s} {secs:4d} s float32 {mode:5s}: {min(ts):.3f} s per track ({secs / min(ts):.0f} x realtime)", flush=True)
        del eng
        torch.cuda.empty_cache()
    k = next(iter(outs["exact"]))
    d = float((outs["exact"][k] - outs["split"][k]).abs().max()); p = float(outs["exact"][k].abs().max())
    print(f"    split vs exact, stem {k}: max|delta| {d:.3e} (peak {p:.3f})", flush=True)
