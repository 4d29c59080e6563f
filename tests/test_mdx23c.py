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


@pytest.mark.parametrize("case", [
    # (H, W, Cin, Cout, KH, KW, stride, pad, residual)
    (9, 13, 64, 72, 3, 3, 1, 1, True),          # ragged pixel / channel tiles, every border tap
    (8, 12, 128, 64, 1, 1, 1, 0, False),        # the shortcut branch
    (10, 6, 64, 128, 2, 2, 2, 0, False),        # the down-scaling convolution
    (130, 3, 64, 4, 3, 3, 1, 1, True),          # more than one pixel tile, the narrowest channel slice
    (8, 32, 768, 136, 3, 3, 1, 1, True),        # the bottleneck's shape class: few tiles, 108 K slices -> split along K, ragged last range
    (4, 8, 1280, 64, 3, 3, 1, 1, False),        # 180 K slices on one tile
])
def test_half_conv_vs_torch(dev, case):
    """alsep_nn_conv2d_f16 against torch.conv2d on the same rounded operands: only the float32 summation order differs"""
    import torch.nn.functional as F
    from audiolab_amd import _lib
    H, W, Cin, Cout, KH, KW, st, pad, with_res = case
    g = torch.Generator().manual_seed(H * 100 + Cout)
    x = (torch.randn(H, W, Cin, generator=g) * 0.7).half()
    w = (torch.randn(Cout, KH, KW, Cin, generator=g) / (KH * KW * Cin) ** 0.5).half()
    Ho, Wo = (H + 2 * pad - KH) // st + 1, (W + 2 * pad - KW) // st + 1
    res = torch.randn(Ho * Wo, Cout, generator=g) if with_res else None
    want = F.conv2d(x.float().permute(2, 0, 1)[None], w.float().permute(0, 3, 1, 2), stride=st, padding=pad)[0].permute(1, 2, 0).reshape(Ho * Wo, Cout)
    if with_res:
        want = want + res
    xd, wd = on(dev, x), on(dev, w)
    y = dev.empty((Ho * Wo, Cout + 8))
    y.fill_(-7.0)
    rd = on(dev, res) if with_res else None
    need = int(dev.lib.alsep_nn_conv2d_f16_workspace_bytes(1, H, W, Cin, Cout, KH, KW, st, st, pad, pad))
    assert (need > 0) == (Cin >= 768)                                                    # only the long-K, few-tile layers are split
    ws = dev.empty((max(need, 16),), torch.uint8)
    dev.check(dev.lib.alsep_nn_conv2d_f16(dev.handle, _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(y), _lib.ptr(rd) if with_res else None, Cout, 1, H, W, Cin,
                                          Cout, KH, KW, st, st, pad, pad, Cout + 8, 4, _lib.ptr(ws) if need else None, need), "alsep_nn_conv2d_f16")
    got = host(y)
    assert np.all(got[:, :4] == -7.0) and np.all(got[:, 4 + Cout:] == -7.0)             # only its channel slice is written
    assert np.max(np.abs(got[:, 4: 4 + Cout] - want.numpy())) < 2e-5 * max(1.0, float(want.abs().max()))


def test_instnorm_half_and_wide_statistics(dev):
    """the float4 statistics kernel (C % 4 == 0) and the half-precision output against torch.instance_norm, ragged slab sizes"""
    import torch.nn.functional as F
    from audiolab_amd import _lib
    for P, Cn in ((1000, 64), (4099, 192), (77, 768), (300, 12)):
        g = torch.Generator().manual_seed(P)
        x = torch.randn(P, Cn, generator=g) * 2 + torch.linspace(-3, 3, Cn)
        gamma, beta = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1
        want = F.gelu(F.instance_norm(x.t()[None, :, :, None], weight=gamma, bias=beta, eps=1e-5))[0, :, :, 0].t()
        ws = dev.empty((int(dev.lib.alsep_nn_instnorm_workspace_bytes(P, Cn)),), torch.uint8)
        xd, gd, bd = on(dev, x), on(dev, gamma), on(dev, beta)
        y32 = dev.empty((P, Cn))
        dev.check(dev.lib.alsep_nn_instnorm(dev.handle, _lib.ptr(xd), _lib.ptr(y32), _lib.ptr(gd), _lib.ptr(bd), P, Cn, 1e-5, 3, _lib.ptr(ws)), "instnorm")
        assert np.max(np.abs(host(y32) - want.numpy())) < 2e-5
        y16 = dev.empty((P, Cn), torch.float16)
        dev.check(dev.lib.alsep_nn_instnorm_f16(dev.handle, _lib.ptr(xd), _lib.ptr(y16), _lib.ptr(gd), _lib.ptr(bd), P, Cn, 1e-5, 3, _lib.ptr(ws)),
                  "instnorm_f16")
        # the float32 result rounded once: equal to rounding the float32 kernel's own output
        assert np.array_equal(host(y16), host(y32).astype(np.float16))


def test_instnorm_half_transposed(dev):
    """alsep_nn_instnorm_f16_t: [T][F][C] float32 -> [T][C][F] half = the half InstanceNorm's rows transposed per frame (ragged tiles)"""
    from audiolab_amd import _lib
    for T, Fw, Cn in ((3, 40, 36), (2, 32, 64), (5, 7, 100)):
        g = torch.Generator().manual_seed(T * 100 + Fw)
        x = torch.randn(T * Fw, Cn, generator=g) * 1.5 + 0.3
        gamma, beta = torch.rand(Cn, generator=g) + 0.5, torch.randn(Cn, generator=g) * 0.1
        ws = dev.empty((int(dev.lib.alsep_nn_instnorm_workspace_bytes(T * Fw, Cn)),), torch.uint8)
        xd, gd, bd = on(dev, x), on(dev, gamma), on(dev, beta)
        y32 = dev.empty((T * Fw, Cn))
        dev.check(dev.lib.alsep_nn_instnorm(dev.handle, _lib.ptr(xd), _lib.ptr(y32), _lib.ptr(gd), _lib.ptr(bd), T * Fw, Cn, 1e-5, 3, _lib.ptr(ws)), "instnorm")
        yt = dev.empty((T, Cn, Fw), torch.float16)
        dev.check(dev.lib.alsep_nn_instnorm_f16_t(dev.handle, _lib.ptr(xd), _lib.ptr(yt), _lib.ptr(gd), _lib.ptr(bd), T, Fw, Cn, 1e-5, 3, _lib.ptr(ws)),
                  "instnorm_f16_t")
        want = host(y32).reshape(T, Fw, Cn).transpose(0, 2, 1).astype(np.float16)
        assert np.array_equal(host(yt), want)


def test_half_mode_vs_storage_oracle(dev):
    """precision="f16" (half-precision convolutions on IEEE-half activations, everything else float32) against the storage-mode
    oracle: the same roundings, float32 arithmetic -- and, as the yardstick, against the float32 oracle (what the mode costs)"""
    from audiolab_amd.mdx23c import MDX23C, MDX23CConfig
    from audiolab_amd._lib import AlsepError
    ocfg = small_cfg(num_channels=64, growth=64, dim_f=64, num_subbands=2, num_scales=1, num_blocks_per_scale=1, chunk_size=64 * 7)
    sd = mo.synthetic_state_dict(ocfg, 2)
    net = MDX23C(MDX23CConfig(**dataclasses.asdict(ocfg)), sd, ctx=dev, precision="f16")
    x = torch.randn(2, ocfg.chunk_size, generator=torch.Generator().manual_seed(5)) * 0.3
    want_h = mo.forward(ocfg, sd, x[None], half=True)[0].numpy()
    want_f = mo.forward(ocfg, sd, x[None])[0].numpy()
    got = host(net.forward(on(dev, x)))
    peak = float(np.max(np.abs(want_f)))
    e_h, cost = float(np.max(np.abs(got - want_h))), float(np.max(np.abs(want_h - want_f)))
    print(f"mdx23c half mode: vs storage oracle {e_h:.3e}, the mode's own cost {cost:.3e}, peak {peak:.3f}")
    assert peak > 1e-3 and cost > 0
    assert e_h < 0.9 * cost + 1e-5 * max(1.0, peak)                     # rounding flips of re-ordered sums stay a fraction of the mode's cost
    with pytest.raises(AlsepError, match="multiples of 64"):
        MDX23C(MDX23CConfig(**dataclasses.asdict(small_cfg())), mo.synthetic_state_dict(small_cfg(), 1), ctx=dev, precision="f16")


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


_FULL_FP32: dict = {}


def _full_size_fp32_oracle():
    from audiolab_amd.synth import synth_mix
    if "w" not in _FULL_FP32:
        ocfg = mo.MDX23CConfig()
        _FULL_FP32["w"] = mo.forward(ocfg, mo.synthetic_state_dict(ocfg, 0), torch.from_numpy(synth_mix(ocfg.chunk_size))[None])[0].numpy()
    return _FULL_FP32["w"]


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
    want = _full_size_fp32_oracle()
    gpu_ctx.synchronize()
    t0 = time.perf_counter()
    got = net.forward(x.cuda())
    gpu_ctx.synchronize()
    dt = time.perf_counter() - t0
    err = float(np.max(np.abs(got.cpu().numpy() - want)))
    print(f"mdx23c full-size chunk: max|delta| = {err:.3e}, peak = {np.max(np.abs(want)):.3f}, {dt * 1e3:.0f} ms (first call)")
    assert np.max(np.abs(want)) > 1e-3 and err < 1e-4 * max(1.0, float(np.max(np.abs(want))))


@pytest.mark.gpu
def test_full_size_chunk_half_precision(gpu_ctx):
    """the same full-size chunk in the half-precision mode against the storage-mode oracle, with the float32 oracle as yardstick"""
    import time
    from audiolab_amd.mdx23c import MDX23C, MDX23CConfig
    from audiolab_amd.synth import synth_mix
    ocfg = mo.MDX23CConfig()
    sd = mo.synthetic_state_dict(ocfg, 0)
    net = MDX23C(MDX23CConfig(), sd, ctx=gpu_ctx, precision="f16")
    x = torch.from_numpy(synth_mix(ocfg.chunk_size))
    want_h = mo.forward(ocfg, sd, x[None], half=True)[0].numpy()
    want_f = _full_size_fp32_oracle()
    got = net.forward(x.cuda())
    gpu_ctx.synchronize()
    t0 = time.perf_counter()
    got = net.forward(x.cuda())
    gpu_ctx.synchronize()
    dt = time.perf_counter() - t0
    got = got.cpu().numpy()
    peak = float(np.max(np.abs(want_f)))
    e_h, cost = float(np.max(np.abs(got - want_h))), float(np.max(np.abs(want_h - want_f)))
    rel = float(np.linalg.norm(got - want_f) / np.linalg.norm(want_f))
    print(f"mdx23c full-size half mode: vs storage oracle {e_h:.3e}, the mode's own cost {cost:.3e} (rel L2 vs fp32 {rel:.3e}), peak {peak:.3f}, "
          f"{dt * 1e3:.0f} ms (second call)")
    assert gpu_ctx.launch_count("nn_conv_hh_kernel") > 0
    assert e_h < 0.9 * cost + 1e-5 * max(1.0, peak)


@pytest.mark.gpu
def test_runner_default_configuration_is_reproducible_full_size(gpu_ctx):
    """the full-size network through the chunked runner in its DEFAULT configuration (half mode: one lane replaying a HIP graph; float32
    mode: four lanes on their own streams) against one lane with plain launches: the same kernels on the same chunks, so the stems agree
    up to the order of the per-lane sums (float32: ~1e-7 of the peak), run after run.  Half-precision networks are held to one lane because
    FFT launches beside the f16 MFMA kernels on another stream come out wrong now and then on this stack (roformer.RoformerRunner)."""
    from audiolab_amd.mdx23c import MDX23C, MDX23CConfig
    from audiolab_amd.roformer import RoformerRunner
    from audiolab_amd.synth import synth_mix
    sd = mo.synthetic_state_dict(mo.MDX23CConfig(), 0)
    mix = torch.from_numpy(synth_mix(700000)).cuda()
    for prec in ("f16", "f32"):
        net = MDX23C(MDX23CConfig(), sd, ctx=gpu_ctx, precision=prec)
        one = RoformerRunner(net, ("Vocals", "Instrumental"), lanes=1, graphs=False).separate(mix)
        dflt = RoformerRunner(net, ("Vocals", "Instrumental"))
        assert dflt.lanes == (1 if prec == "f16" else 4) and dflt.graphs
        peak = float(one["Vocals"].abs().max())
        for rep in range(3):                                                             # from the second pass on every lane replays its graph
            out = dflt.separate(mix)
            for k in one:
                assert float((one[k] - out[k]).abs().max()) < 2e-6 * peak, (prec, k, rep, float((one[k] - out[k]).abs().max()), peak)
        assert peak > 1e-3
    asked = RoformerRunner(MDX23C(MDX23CConfig(), sd, ctx=gpu_ctx, precision="f16"), ("Vocals", "Instrumental"), lanes=4)
    assert asked.lanes == 1                                                              # refused with a warning


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
    auto = Separator(ctx=dev, use_autocast=True, allow_synthetic=True, roster={name: ("mdx23c", small, MODEL_ROSTER[name][2])})
    auto.load_model(name)                                                # 16 / 8 channels: no half mode for this one -- float32 with a warning, not an error
    assert auto.model_instance.net.half is False
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
