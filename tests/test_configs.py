"""GPU (-m gpu): BASELINE.json configs[3] and configs[4] in their NAMED combinations, through the same entry points the
reference's callers use, against the oracle composition.

configs[3] "batch of tracks, MDX+Demucs ensemble (wrappers/separate ensemble mode)": several tracks through
``Separate().process_audio`` (wrappers/separate.py:233-388) -> ``separate_music`` (stem_separator.py:949-1001): two MDX-Net
vocal models at the geometry of the reference's own vocal members (UVR-MDX-NET-Voc_FT / Kim_Vocal_2: n_fft 7680, dim_f 3072,
dim_t 256, L = 11, g = 48; stem_separator.py:384-385), blended + de-bled, then htdemucs_6s at its real size on the mix
(:459-503).  Checked per stem file against the composition of the oracle parts, |delta| < 1e-4 PCM (fp32).

configs[4] "60 min 48 kHz 8-channel long-form, fp16, overlap=0.75": the bench-geometry network in IEEE-half storage with f16
MFMA, Hann overlap-add at overlap 0.75, an 8-channel 48 kHz file through ``Separator.separate`` at its native rate, against
``oracle.mdx_oracle.demix_ola`` with the f16 STORAGE oracle; plus the size-independent property the overlap-add offers at
the full 60-minute length (a shift of the input by one chunk step shifts the interior of the output by the same amount)."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import ensemble_oracle as eo
from oracle import htdemucs_oracle as ho
from oracle import mdx_oracle as mo
from oracle import tdfnet_oracle

pytestmark = pytest.mark.gpu


def _seed(name: str) -> int:
    return int.from_bytes(hashlib.sha256(name.encode()).digest()[:4], "little")


def _rel(a, b) -> float:
    d = (np.asarray(a) - np.asarray(b)).astype(np.float64)
    return float(np.sqrt((d ** 2).sum() / max((np.asarray(b).astype(np.float64) ** 2).sum(), 1e-30)))


def test_config3_track_batch_mdx7680_ensemble_plus_htdemucs(gpu_ctx, tmp_path, monkeypatch):
    from audiolab_amd import wavio
    from audiolab_amd.engine import MODEL_ROSTER, Separator
    from audiolab_amd.handlers import config
    from audiolab_amd.synth import synth_mix, synthetic_state_dict
    from audiolab_amd.util.data_classes import ProjectFiles
    from audiolab_amd.wrappers.separate import Separate
    monkeypatch.setattr(config, "output_path", str(tmp_path / "outputs"))
    Separate._instance = None
    # configs[3] is "MDX + Demucs": the roster is cut to the MDX-Net files and htdemucs, so that the first two ensemble members
    # the orchestrator finds are the reference's MDX-Net vocal models at their real geometry (bench.py --workload tracks does the same)
    roster = {k: v for k, v in MODEL_ROSTER.items() if k.endswith(".onnx") or v[0] == "demucs"}
    assert roster["htdemucs_6s.yaml"][2] == {"shifts": 2, "overlap": 0.25}       # DemucsSeparator's defaults: two shift passes
    members = ("UVR-MDX-NET-Voc_FT.onnx", "Kim_Vocal_2.onnx")
    for m in members:
        cfg = roster[m][2]
        assert (cfg.n_fft, cfg.dim_f, cfg.dim_t, cfg.num_blocks, cfg.g) == (7680, 3072, 256, 11, 48)
    eng = Separator(ctx=gpu_ctx, use_autocast=False, allow_synthetic=True, roster=roster, max_batch=2)
    gen = 1024 * 255 - 7680
    lengths = (150001, gen + 40000)                            # one model window; two model windows (ragged tail)
    mixes, srcs = [], []
    for k, n in enumerate(lengths):
        mix = synth_mix(n, seed=300 + k)
        src = tmp_path / f"track{k}.wav"
        wavio.write_wav(str(src), mix, 44100)
        mixes.append(mix)
        srcs.append(src)
    wrapper = Separate()
    monkeypatch.setattr(Separate, "engine_options", {"separator": eng, "ensemble_strength": 2, "precision": "fp32"})
    gpu_ctx.launch_counts_reset()
    out = wrapper.process_audio([ProjectFiles(str(s)) for s in srcs], vocals_only=False, separate_bg_vocals=False)
    assert len(out) == 2
    assert gpu_ctx.launch_count("stft_r16_kernel") >= 4        # the three-pass 7680 STFT ran for both members on both tracks

    def oracle_mdx(name, mix):
        cfg = roster[name][2]
        sd = synthetic_state_dict(cfg, seed=_seed(name))       # weights are data; the forward below is the oracle's
        g = mo.MDXGeometry(cfg.dim_f, cfg.dim_t, cfg.n_fft, cfg.hop)

        def run(spek):
            with torch.no_grad():
                return tdfnet_oracle.forward(sd, torch.from_numpy(np.ascontiguousarray(spek, dtype=np.float32)), cfg.num_blocks,
                                             cfg.l, cfg.bn).numpy()
        return mo.demix(mix, g, run, chunks=0, margin=44100, dtype=np.float32)[0]

    ocfg = ho.HTDemucsConfig()
    dsd = ho.synthetic_state_dict(ocfg, _seed("htdemucs_6s.yaml"))
    worst = {}
    for proj, mix, n in zip(out, mixes, lengths):
        stems = {os.path.basename(p).split("__")[1][:-4]: wavio.read_wav(p)[0] for p in proj.last_outputs}
        assert set(stems) == {"(Vocals)", "(Instrumental)", "(Drums)", "(Bass)", "(Guitar)", "(Piano)", "(Other)"}
        v = [oracle_mdx(m, mix) for m in members]
        i = [mix - x for x in v]
        vocals = eo.blend_tracks(v, [6.9, 6.9])                                  # stem_separator.py:412
        inst = eo.blend_tracks(i, [14.9, 14.9])                                  # :413
        inst, _ = eo.debleed(mix, vocals, inst, 44100, 0.2)                      # :415-456 (residual_blend capped at 0.2, :389-390)
        six = ho.separate(ocfg, dsd, torch.from_numpy(mix), shifts=2, overlap=0.25, seed=0).numpy()
        want = {"(Vocals)": vocals, "(Instrumental)": inst}
        for idx, name in enumerate(ocfg.sources):
            if name != "vocals":                                                 # :491-500: the vocals output is ignored
                want[f"({name.capitalize()})"] = six[idx]
        for k in want:
            assert stems[k].shape == (2, n)
            err = float(np.max(np.abs(stems[k] - want[k])))
            worst[k] = max(worst.get(k, 0.0), err)
            assert np.max(np.abs(want[k])) > 1e-3
    print("configs[3] max|delta| per stem vs the oracle composition:", {k: f"{e:.2e}" for k, e in worst.items()})
    assert max(worst.values()) < 1e-4, worst


def _longform_engine(ctx, **kw):
    from audiolab_amd.engine import Separator
    from audiolab_amd.tdfnet import TDFNetConfig
    return Separator(ctx=ctx, dtype=torch.float16, allow_synthetic=True, max_batch=8, chunker="ola", overlap=0.75, sample_rate=48000,
                     roster={"longform_vocals.onnx": ("Vocals", "Instrumental", TDFNetConfig())}, **kw)


def test_config4_f16_ola075_48k_8ch_vs_storage_oracle(gpu_ctx, tmp_path):
    """f16 x overlap-add 0.75 x 48 kHz x 8 channels in ONE run: an 8-channel 48 kHz WAV through Separator.separate (native rate: no
    resampling when the engine's sample_rate is the file's), four chunks per stereo pair at the bench geometry."""
    from audiolab_amd import wavio
    from audiolab_amd.synth import synth_mix, synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNetConfig
    cfg = TDFNetConfig()
    g = mo.MDXGeometry(cfg.dim_f, cfg.dim_t, cfg.n_fft, cfg.hop)
    n = 100000
    assert len(mo.ola_plan(n, g, 0.75)["starts"]) >= 3
    mix8 = np.concatenate([synth_mix(n, sr=48000, seed=50 + c) for c in range(4)])
    src = tmp_path / "longform.wav"
    wavio.write_wav(str(src), mix8, 48000)
    eng = _longform_engine(gpu_ctx, output_dir=str(tmp_path / "out"))
    eng.load_model("longform_vocals.onnx")
    gpu_ctx.launch_counts_reset()
    names = eng.separate(str(src))
    assert gpu_ctx.launch_count("conv3x3_bf16_m0_kernel") > 0 and gpu_ctx.launch_count("conv3x3_bf16_mq_kernel") > 0   # f16 build of the production kernels
    got = {}
    for name in names:
        audio, sr = wavio.read_wav(os.path.join(str(tmp_path / "out"), name))
        assert sr == 48000 and audio.shape == (8, n)
        got["Vocals" if "(Vocals)" in name else "Instrumental"] = audio
    assert set(got) == {"Vocals", "Instrumental"}
    sd = synthetic_state_dict(cfg, seed=_seed("longform_vocals.onnx"))

    def run_with(storage):
        def run(spek):
            with torch.no_grad():
                return tdfnet_oracle.forward(sd, torch.from_numpy(np.ascontiguousarray(spek, dtype=np.float32)), cfg.num_blocks, cfg.l,
                                             cfg.bn, storage=storage).numpy()
        return run
    # oracle on the first and the last stereo pair (storage mode and fp32); the middle pairs are checked against the 2-channel path below
    for c0 in (0, 6):
        pair = mix8[c0:c0 + 2]
        # the engine's sequence in this mode (normalise to 0.9, overlap-add, spectral inversion for the secondary stem): oracle/mdx_oracle.separate_ola
        want_st, _ = mo.separate_ola(pair, g, run_with(torch.float16), overlap=0.75, compensate=1.0)
        want_32, sec_32 = mo.separate_ola(pair, g, run_with(None), overlap=0.75, compensate=1.0)
        r_st, r_32, r_oo = _rel(got["Vocals"][c0:c0 + 2], want_st), _rel(got["Vocals"][c0:c0 + 2], want_32), _rel(want_st, want_32)
        print(f"configs[4] channels {c0}-{c0 + 1}: vs f16-storage oracle rel L2 = {r_st:.3e}, vs fp32 oracle {r_32:.3e} "
              f"(SDR {-20 * np.log10(r_32):.1f} dB), storage oracle vs fp32 oracle {r_oo:.3e}, max|delta| vs fp32 oracle = "
              f"{np.max(np.abs(got['Vocals'][c0:c0 + 2] - want_32)):.3e} (peak {np.max(np.abs(want_32)):.3f})")
        # the yardstick of tests/test_gpu_parity.py::test_full_size_mdx_f16_vs_oracle: distance to the storage oracle below the cost of
        # the storage type itself, total error within 1.25 x of it, and an absolute bound on the f16 error against fp32
        assert r_st < 0.8 * r_oo and r_32 < 1.25 * r_oo and r_32 < 5e-2
        # secondary stem: the inversion is linear, so against the oracle's it carries the primary's error plus the f16 spectrogram of the
        # match-mix pass; and it must be the inversion of the GPU's own primary to fp32 accuracy apart from that pass
        r_sec = _rel(got["Instrumental"][c0:c0 + 2], sec_32)
        assert r_sec < 1.25 * r_oo * np.linalg.norm(want_32) / np.linalg.norm(sec_32) + 2e-3, r_sec
    # every pair of the 8-channel run equals the same pair run alone through the 2-channel path (bit for bit: same kernels, same order)
    for c0 in (2, 4):
        alone = eng.separate_array(mix8[c0:c0 + 2])["Vocals"].cpu().numpy()
        assert np.array_equal(alone, got["Vocals"][c0:c0 + 2])


def test_config4_full_length_shift_property(gpu_ctx):
    """configs[4] at its stated length -- 60 min at 48 kHz, the bench-geometry network in f16, overlap 0.75 -- on one stereo pair:
    delaying the input by one chunk step (65 280 samples) moves every interior chunk one slot down the batch, so the interior of the
    output is the same signal delayed by the step.  (Needs ~12 GB of HBM; runs in ~10 s.)"""
    eng = _longform_engine(gpu_ctx, normalization_threshold=0.0)       # a peak-dependent gain would differ between the two runs' edge regions
    eng.max_batch = 32
    eng.load_model("longform_vocals.onnx")
    n = 3600 * 48000
    step = int(0.25 * 1024 * 255)
    gen_t = torch.Generator(device="cuda").manual_seed(7)
    t = torch.arange(n, device="cuda", dtype=torch.float32) / 48000.0
    x = 0.08 * torch.randn((2, n), device="cuda", generator=gen_t)
    for k, f in enumerate((110.0, 440.0, 3520.0)):
        x[0] += 0.1 * torch.sin(2 * np.pi * f * t + 0.3 * k)
        x[1] += 0.1 * torch.sin(2 * np.pi * f * t + 0.7 + 0.3 * k)
    del t
    ya = eng.separate_array(x)["Vocals"]
    assert ya.shape == (2, n) and bool(torch.isfinite(ya).all())
    xs = torch.zeros_like(x)
    xs[:, step:] = x[:, :-step]
    del x
    yb = eng.separate_array(xs)["Vocals"]
    lo, hi = 8 * step, n - 8 * step
    diff = float((yb[:, lo:hi] - ya[:, lo - step:hi - step]).abs().max())
    peak = float(ya.abs().max())
    print(f"configs[4] full length: shift-by-step interior max|delta| = {diff:.3e} (peak {peak:.3f})")
    assert peak > 1e-3 and diff < 1e-6


def test_config2_htdemucs_6s_ten_minutes_properties(gpu_ctx):
    """BASELINE configs[2] at its stated length -- htdemucs_6s (41 M parameters, six sources), a 10-minute track, overlap 0.25, the
    engine's default two shift passes -- on one GPU, through Separator.separate_array.  No CPU oracle can follow at this length (the
    7.8 s segment and the 25 s runner have theirs in test_htdemucs.py), so the checks are the properties the path offers at any size:
    (1) DemucsSeparator normalises the track by the mean / std of its mono mix and undoes it on the stems, so separate(x / 2) is
    separate(x) / 2 -- a halving is exact in floating point, the network sees bit-identical segments; (2) the run is deterministic
    (per-lane weighted sums are added in a fixed order); (3) six finite stems of the track's length, none of them silent."""
    import time
    from audiolab_amd.engine import MODEL_ROSTER, Separator
    eng = Separator(ctx=gpu_ctx, use_autocast=False, allow_synthetic=True, roster={"htdemucs_6s.yaml": MODEL_ROSTER["htdemucs_6s.yaml"]})
    eng.load_model("htdemucs_6s.yaml")
    assert MODEL_ROSTER["htdemucs_6s.yaml"][2] == {"shifts": 2, "overlap": 0.25}
    n = 600 * 44100
    gen = torch.Generator(device="cuda").manual_seed(11)
    t = torch.arange(n, device="cuda", dtype=torch.float32) / 44100.0
    x = 0.05 * torch.randn((2, n), device="cuda", generator=gen)
    for k, f in enumerate((82.4, 440.0, 2793.0)):
        x[0] += 0.12 * torch.sin(2 * np.pi * f * t + 0.4 * k) * (0.6 + 0.4 * torch.sin(2 * np.pi * 0.05 * (k + 1) * t))
        x[1] += 0.12 * torch.sin(2 * np.pi * f * t + 0.9 + 0.4 * k) * (0.6 + 0.4 * torch.sin(2 * np.pi * 0.07 * (k + 1) * t))
    del t
    gpu_ctx.synchronize()
    t0 = time.perf_counter()
    a = eng.separate_array(x)
    gpu_ctx.synchronize()
    dt = time.perf_counter() - t0
    assert list(a) == ["Drums", "Bass", "Other", "Vocals", "Guitar", "Piano"]
    for k, v in a.items():
        assert v.shape == (2, n) and bool(torch.isfinite(v).all()) and float(v.abs().max()) > 1e-4, k
    b = eng.separate_array(x)
    assert all(torch.equal(a[k], b[k]) for k in a)                                   # (2)
    del b
    h = eng.separate_array(x * 0.5)
    worst = max(float((h[k] - 0.5 * a[k]).abs().max()) / float(a[k].abs().max()) for k in a)
    print(f"configs[2] htdemucs_6s, 10 min: {dt:.2f} s first run ({600 / dt:.0f} x realtime for the six stems); "
          f"separate(x / 2) vs separate(x) / 2: max rel {worst:.2e}")
    assert worst < 1e-6                                                              # (1)
