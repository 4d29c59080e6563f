"""MDX23C / TFC-TDF v3 (audiolab_amd/mdx23c.py -> csrc/nn.hip, vrnet.hip conv, fft.hip) against the torch-CPU fp32 oracle
(oracle/mdx23c_oracle.py; PARITY UNPINNED) on the emulated kernels and on the GPU."""
import dataclasses

import numpy as np
import pytest
import torch

from oracle import mdx23c_oracle as mo
from oracle import roformer_oracle as ro
from tests.conftest import host, on


def small_cfg(**kw):
    base = dict(instruments=("vocals", "other"), n_fft=256, hop=64, dim_f=96, num_subbands=2, num_scales=2, num_blocks_per_scale=2,
                num_channels=16, growth=8, bottleneck_factor=4, chunk_size=64 * 15, num_overlap=2, sample_rate=8000)
    base.update(kw)
    return mo.MDX23CConfig(**base)


def build(dev, ocfg, seed=1):
    from audiolab_amd.mdx23c import MDX23C, MDX23CConfig
    sd = mo.synthetic_state_dict(ocfg, seed)
    return MDX23C(MDX23CConfig(**dataclasses.asdict(ocfg)), sd, ctx=dev), sd


def test_forward_one_chunk_vs_oracle(dev):
    ocfg = small_cfg()
    net, sd = build(dev, ocfg)
    x = torch.randn(2, ocfg.chunk_size, generator=torch.Generator().manual_seed(3)) * 0.3
    want = mo.forward(ocfg, sd, x[None])[0].numpy()
    got = host(net.forward(on(dev, x)))
    assert got.shape == want.shape == (2, 2, ocfg.chunk_size)
    err = float(np.max(np.abs(got - want)))
    print(f"mdx23c forward: max|delta| = {err:.3e}, peak = {np.max(np.abs(want)):.3f}")
    assert np.max(np.abs(want)) > 1e-3 and err < 1e-4 * max(1.0, float(np.max(np.abs(want))))


def test_six_way_drum_split_through_the_runner(dev):
    """the drum-kit splitter's shape (six instruments) through the chunked runner and the engine's roster entry"""
    from audiolab_amd.roformer import RoformerRunner
    if dev.device.type == "cpu":
        pytest.skip("GPU only (the emulated suite covers the network above)")
    names = ("kick", "snare", "toms", "hh", "ride", "crash")
    ocfg = small_cfg(instruments=names)
    net, sd = build(dev, ocfg, seed=4)
    mix = torch.randn(2, 3000, generator=torch.Generator().manual_seed(6)) * 0.3
    rcfg = ro.RoformerConfig(chunk_size=ocfg.chunk_size, num_overlap=ocfg.num_overlap, num_stems=6)
    want = ro.demix_track(rcfg, None, mix, fwd=lambda x: mo.forward(ocfg, sd, x)).numpy()
    out = RoformerRunner(net, tuple(n.capitalize() for n in names)).separate(on(dev, mix))
    got = np.stack([host(out[k.capitalize()]) for k in names])
    assert float(np.max(np.abs(got - want))) < 1e-4 * max(1.0, float(np.max(np.abs(want))))


@pytest.mark.gpu
def test_full_size_chunk_vs_oracle(gpu_ctx):
    """MDX23C-8KFFT-InstVoc_HQ's shape (n_fft 8192, dim_f 4096, 4 sub-bands, 5 scales, 128..768 channels) on one 5.9 s chunk"""
    import time
    from audiolab_amd.mdx23c import MDX23C, MDX23CConfig
    from audiolab_amd.synth import synth_mix
    ocfg = mo.MDX23CConfig()
    sd = mo.synthetic_state_dict(ocfg, 0)
    net = MDX23C(MDX23CConfig(), sd, ctx=gpu_ctx)
    x = torch.from_numpy(synth_mix(ocfg.chunk_size))
    want = mo.forward(ocfg, sd, x[None])[0].numpy()
    gpu_ctx.synchronize()
    t0 = time.perf_counter()
    got = net.forward(x.cuda())
    gpu_ctx.synchronize()
    dt = time.perf_counter() - t0
    err = float(np.max(np.abs(got.cpu().numpy() - want)))
    print(f"mdx23c full-size chunk: max|delta| = {err:.3e}, peak = {np.max(np.abs(want)):.3f}, {dt * 1e3:.0f} ms (first call)")
    assert np.max(np.abs(want)) > 1e-3 and err < 1e-4 * max(1.0, float(np.max(np.abs(want))))


def test_engine_roster_entries(dev):
    """the reference's two MDX23C model files resolve to this network in the default roster (ensemble slot 4: Vocals / Instrumental;
    drum-kit splitter: the six labels stem_separator.py:563-574 matches), and the orchestrator's drum stage consumes the six outputs"""
    from audiolab_amd.engine import MODEL_ROSTER, Separator
    from audiolab_amd.mdx23c import MDX23CConfig
    from audiolab_amd.separator.stem_separator import EnsembleDemucsMDXMusicSeparationModel
    assert MODEL_ROSTER["MDX23C-8KFFT-InstVoc_HQ.ckpt"][2]["labels"] == ("Vocals", "Instrumental")
    if dev.device.type == "cpu":
        pytest.skip("GPU only")
    name = "MDX23C-DrumSep-aufr33-jarredou.ckpt"
    small = MDX23CConfig(**dataclasses.asdict(small_cfg(instruments=("kick", "snare", "toms", "hh", "ride", "crash"))))
    eng = Separator(ctx=dev, use_autocast=False, allow_synthetic=True, roster={name: ("mdx23c", small, MODEL_ROSTER[name][2])})
    eng.load_model(name)
    drums = torch.randn(2, 2500, generator=torch.Generator().manual_seed(8)) * 0.3
    out = eng.separate_array(drums)
    assert list(out) == ["Kick", "Snare", "Toms", "HH", "Ride", "Crash"]
    model = EnsembleDemucsMDXMusicSeparationModel({}, separator=eng)
    results = {"song": {"sr": 44100, "instrumental": on(dev, drums), "drums": on(dev, drums), "output_folder": "/mem"}}
    model._advanced_drum_separation_all(results)
    r = results["song"]
    for key, label in (("drums_kick", "Kick"), ("drums_snare", "Snare"), ("drums_toms", "Toms"), ("drums_hh", "HH"), ("drums_ride", "Ride"),
                       ("drums_crash", "Crash")):
        assert torch.equal(r[key], out[label])
    assert r["drums_other"].shape == drums.shape
