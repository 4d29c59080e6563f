"""VR multi-band front / back end (audiolab_amd/vr_frontend.py) against the committed reference vectors and the numpy oracle.

tests/golden/vr_frontend.npz was produced by the reference's own spec_utils functions (oracle/make_golden_vr_frontend.py).  The same test
bodies run on the CPU emulation of the kernels and on cuda:0 (``dev`` fixture)."""
import os

import numpy as np
import pytest
import torch

from oracle import vr_oracle as vo
from oracle.toy import synth_mix

from tests.conftest import host, on

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vr_frontend.npz")


def _toy_pred(X):
    mag = np.abs(X)
    return (mag * (0.4 + 0.5 * np.linspace(0, 1, X.shape[1])[None, :, None])).astype(np.float32)


def test_oracle_matches_reference_vectors():
    """the oracle alone against the reference's outputs (CPU only)"""
    g = np.load(GOLD)
    n = int(g["n"][0])
    wave = synth_mix(n, seed=int(g["seed"][0]))
    mp = vo.MODEL_PARAMS["4band_v2"]
    X, he, hh = vo.front_end(wave, mp)
    assert tuple(g["X_shape"]) == X.shape and hh == int(g["hh"][0])
    assert np.max(np.abs(X.reshape(-1)[g["X_idx"]] - g["X_val"])) < 1e-6 * np.max(np.abs(g["X_val"]))
    assert abs(np.sqrt((np.abs(X) ** 2).sum()) - g["X_l2"][0]) < 1e-5 * g["X_l2"][0]
    inst, voc = vo.back_end(X, _toy_pred(X), np.exp(1.0j * np.angle(X)), he, hh, mp)
    assert len(inst) == int(g["inst_len"][0])
    assert np.max(np.abs(inst[::7] - g["inst_wave"])) < 1e-5
    assert np.max(np.abs(voc[::7] - g["voc_wave"])) < 1e-5


def test_resampler_restatements_vs_scipy():
    """oracle/vr_oracle.py resample_poly / resample_fft against the scipy routines librosa calls for "polyphase" and "scipy" (scipy is a
    dependency of the reference and importable here): the pin of the band chain's resamplers"""
    import scipy.signal as ss
    rng = np.random.default_rng(0)
    for n in (1000, 4411, 44100 + 7):
        x = rng.standard_normal((2, n)).astype(np.float32)
        for up, down in ((1, 3), (1, 2), (2, 1), (3, 1), (160, 147)):
            want = ss.resample_poly(x, up, down, axis=-1)
            got = vo.resample_poly(x, up, down)
            assert got.shape == want.shape and got.dtype == want.dtype and np.max(np.abs(got - want)) < 2e-6
        for num in (2 * n, 3 * n, n // 2, n // 3 + 1, n + 1):
            want = ss.resample(x, num, axis=-1)                                          # float32 pocketfft; the restatement runs in float64
            assert np.max(np.abs(vo.resample_fft(x, num) - want)) < 5e-6
    with pytest.raises(ValueError, match="kaiser_fast"):
        vo.resample(x, 48000, 44100, "kaiser_fast")
    assert vo.resample(x, 7350, 7350, "sinc_fastest") is x


@pytest.mark.parametrize("n", [999, 480 * 40 + 3])
def test_device_resamplers_vs_oracle(dev, n):
    """alsep_resample_poly / alsep_resample_fft through ensemble.resample_poly / resample_fft: the chain's ratios, a ragged general
    ratio, an odd row count, down- and upsampling with even and odd lengths (scipy's Nyquist rule)"""
    from audiolab_amd import ensemble
    rng = np.random.default_rng(n)
    x = rng.standard_normal((3, n)).astype(np.float32)
    xd = on(dev, x)
    for sr_in, sr_out in ((44100, 14700), (14700, 7350), (7350, 14700), (48000, 44100)):
        got = host(ensemble.resample_poly(dev, xd, sr_in, sr_out))
        want = vo.resample(x, sr_in, sr_out, "polyphase")
        assert got.shape == want.shape and np.max(np.abs(got - want)) < 3e-6
    for sr_in, sr_out in ((7350, 14700), (14700, 44100), (44100, 14700), (14700, 7350), (44100, 48000)):
        for rows in (3, 2):
            got = host(ensemble.resample_fft(dev, xd[:rows], sr_in, sr_out))
            want = vo.resample(x[:rows], sr_in, sr_out, "scipy")
            assert got.shape == want.shape and np.max(np.abs(got - want)) < 3e-6
    even = on(dev, x[:1, : n - n % 2])                                                   # even length both ways
    for sr_in, sr_out in ((2, 1), (1, 2)):
        want = vo.resample(host(even), sr_in, sr_out, "scipy")
        assert np.max(np.abs(host(ensemble.resample_fft(dev, even, sr_in, sr_out)) - want)) < 3e-6
    assert ensemble.resample_poly(dev, xd, 7350, 7350) is xd


def test_device_resamplers_long_tracks(dev):
    """lengths past the grid caps of the launches (2048 x 256 outputs for the polyphase kernel, 65536 x 256 elements for the Fourier
    resampler's pack / spectrum / unpack kernels): a stereo minute down the VR chain's first band ratio, and -- on the GPU, where the
    2^25-point transforms take milliseconds -- 6.4 minutes up from 14.7 kHz (outputs beyond the cap were never written)"""
    from audiolab_amd import ensemble
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 60 * 44100)).astype(np.float32)
    got = host(ensemble.resample_poly(dev, on(dev, x), 44100, 14700))
    want = vo.resample(x, 44100, 14700, "polyphase")
    assert got.shape == want.shape and got.shape[1] * 2 > 2048 * 256
    assert np.max(np.abs(got - want)) < 3e-6
    assert np.max(np.abs(got[:, -1000:] - want[:, -1000:])) < 3e-6 and np.any(got[:, -1000:] != 0)
    if dev.device.type != "cuda":
        return
    n = 65536 * 256 // 3 + 12345                                                        # output 3 n > 65536 x 256
    x = rng.standard_normal((2, n)).astype(np.float32)
    got = host(ensemble.resample_fft(dev, on(dev, x), 14700, 44100))
    want = vo.resample(x, 14700, 44100, "scipy")
    assert got.shape == want.shape and got.shape[1] > 65536 * 256
    assert np.max(np.abs(got - want)) < 5e-6


def test_front_and_back_end_vs_reference_vectors(dev):
    from audiolab_amd.vr_frontend import VRFrontEnd
    g = np.load(GOLD)
    n = int(g["n"][0])
    wave = synth_mix(n, seed=int(g["seed"][0]))
    f = VRFrontEnd("4band_v2", dev)
    X_d, he_d = f.analyse(on(dev, wave))
    X, he = host(X_d), host(he_d)
    assert X.shape == tuple(g["X_shape"]) and he.shape[1] == int(g["hh"][0])
    peak = np.max(np.abs(g["X_val"]))
    assert np.max(np.abs(X.reshape(-1)[g["X_idx"]] - g["X_val"])) < 2e-5 * peak          # fp32 FFT + fp32 resampler taps vs float64 numpy
    assert abs(np.sqrt((np.abs(X) ** 2).sum()) - g["X_l2"][0]) < 1e-5 * g["X_l2"][0]
    pred = on(dev, _toy_pred(X))
    y, v = f.split(pred, X_d)
    inst, voc = host(f.synthesise(y, he_d)), host(f.synthesise(v, he_d))                 # [2, n']
    assert inst.shape == (2, int(g["inst_len"][0]))
    scale = np.max(np.abs(g["inst_wave"]))
    assert np.max(np.abs(inst.T[::7] - g["inst_wave"])) < 2e-5 * max(scale, 1.0)
    assert np.max(np.abs(voc.T[::7] - g["voc_wave"])) < 2e-5 * max(scale, 1.0)


@pytest.mark.parametrize("n", [44100 + 7, 480 * 40])
def test_stages_vs_oracle(dev, n):
    """each stage separately on another length (one with every band's frame count cut by the shortest band)"""
    from audiolab_amd.vr_frontend import VRFrontEnd
    wave = synth_mix(n, seed=n)
    mp = vo.MODEL_PARAMS["4band_v3"]
    f = VRFrontEnd("4band_v3", dev)
    X_o, he_o, hh = vo.front_end(wave, mp)
    X_d, he_d = f.analyse(on(dev, wave))
    assert np.max(np.abs(host(X_d) - X_o)) < 2e-5 * np.max(np.abs(X_o))
    assert np.max(np.abs(host(he_d) - he_o)) < 2e-5 * np.max(np.abs(X_o))
    rng = np.random.default_rng(3)
    pred = (np.abs(X_o) * rng.random(X_o.shape)).astype(np.float32)
    X_in = on(dev, np.ascontiguousarray(X_o))
    y_d, v_d = f.split(on(dev, pred), X_in)
    y_o = pred * np.exp(1.0j * np.angle(X_o))
    assert np.max(np.abs(host(y_d) - y_o)) < 1e-6 * np.max(np.abs(X_o))
    assert np.max(np.abs(host(v_d) - (X_o - y_o))) < 1e-6 * np.max(np.abs(X_o))
    he_in = on(dev, np.ascontiguousarray(he_o))
    for spec in (y_o.astype(np.complex64), (X_o - y_o).astype(np.complex64)):
        w_o = vo.cmb_spectrogram_to_wave(spec, mp, hh, vo.mirroring(spec, he_o, mp)).T
        w_d = host(f.synthesise(on(dev, np.ascontiguousarray(spec)), he_in))
        assert w_d.shape == w_o.shape
        assert np.max(np.abs(w_d - w_o)) < 2e-5 * max(1.0, np.max(np.abs(w_o)))
    w_o = vo.cmb_spectrogram_to_wave(X_o, mp).T                                           # no mirroring (high_end_process "none")
    assert np.max(np.abs(host(f.synthesise(X_in)) - w_o)) < 2e-5 * max(1.0, np.max(np.abs(w_o)))


@pytest.mark.parametrize("params", ["4band_v2_sn", "4band_v3_sn"])
def test_stereo_n_sets_vs_oracle(dev, params):
    """the "_sn" parameter sets (top band converted with "stereo_n", modelparams/4band_v2_sn.json:48; the BG-vocal model's set): front
    end and back end against the oracle, the conversion's exact inverse (analyse -> synthesise without mirroring returns what the plain
    set returns), and that the conversion really happened (the top band's rows differ from the plain set's, the lower bands' do not).
    The rule is upstream's, restated from memory -- no vector of the reference pins it (oracle/vr_oracle.py MODEL_PARAMS note)."""
    from audiolab_amd.vr_frontend import VRFrontEnd
    n = 480 * 30 + 5
    wave = synth_mix(n, seed=17)
    wave[1] = 0.6 * wave[1] + 0.2 * np.roll(wave[0], 37)                                 # channels that differ: the conversion is visible
    mp, plain = vo.MODEL_PARAMS[params], params[:-3]
    f, f_plain = VRFrontEnd(params, dev), VRFrontEnd(plain, dev)
    X_o, he_o, hh = vo.front_end(wave, mp)
    X_d, he_d = f.analyse(on(dev, wave))
    peak = np.max(np.abs(X_o))
    assert np.max(np.abs(host(X_d) - X_o)) < 2e-5 * peak and np.max(np.abs(host(he_d) - he_o)) < 2e-5 * peak
    X_p = host(f_plain.analyse(on(dev, wave))[0])
    lo = sum(mp["band"][d]["crop_stop"] - mp["band"][d]["crop_start"] for d in (1, 2, 3))
    assert np.max(np.abs(host(X_d)[:, :lo] - X_p[:, :lo])) == 0.0
    assert np.max(np.abs(host(X_d)[:, lo:] - X_p[:, lo:])) > 1e-2 * peak
    want = (X_p[:, lo:] + 0.25 * X_p[::-1, lo:]) / 0.9375
    assert np.max(np.abs(host(X_d)[:, lo:] - want)) < 1e-6 * peak
    for spec in (X_o.astype(np.complex64), (0.5 * X_o).astype(np.complex64)):
        w_o = vo.cmb_spectrogram_to_wave(spec, mp, hh, vo.mirroring(spec, he_o, mp)).T
        w_d = host(f.synthesise(on(dev, np.ascontiguousarray(spec)), on(dev, np.ascontiguousarray(he_o))))
        assert w_d.shape == w_o.shape and np.max(np.abs(w_d - w_o)) < 2e-5 * max(1.0, np.max(np.abs(w_o)))
    back, back_plain = host(f.synthesise(X_d)), host(f_plain.synthesise(f_plain.analyse(on(dev, wave))[0]))
    assert np.max(np.abs(back - back_plain)) < 2e-5 * max(1.0, np.max(np.abs(back_plain)))


def test_unknown_channel_conversion_is_refused(dev, monkeypatch):
    from audiolab_amd import vr_frontend
    from audiolab_amd._lib import AlsepError
    bad = {**vr_frontend._BANDS_4, 4: dict(vr_frontend._BANDS_4[4], convert_channels="mid_side_c")}
    monkeypatch.setitem(vr_frontend.MODEL_PARAMS, "bad", dict(vr_frontend.MODEL_PARAMS["4band_v3"], band=bad))
    with pytest.raises(AlsepError, match="mid_side_c"):
        vr_frontend.VRFrontEnd("bad", dev)


class _TiltNet:
    """stands in for a VRNet in the runner: [B, bins, frames, 2] -> the input times a frequency tilt"""
    offset, output_bin = 8, 673

    def __init__(self, ctx):
        self.ctx = ctx

    def forward_nhwc(self, x, aggressiveness=None):
        return x * torch.linspace(0.3, 0.9, x.shape[1], device=x.device)[None, :, None, None]


def test_separator_end_to_end(dev):
    """wave -> front end -> VR network (windowed runner) -> back end: VRSeparator against the oracle's back end fed with the same
    prediction (the network and its runner have their own reference vectors: tests/test_emul_vrnet.py)"""
    from audiolab_amd import vrnet
    from audiolab_amd.vr_frontend import VRSeparator
    n = 480 * 20 + 11
    wave = synth_mix(n, seed=99)
    if dev.device.type == "cpu":                         # the 673-bin network takes minutes on the emulation: a stand-in mask there
        net = _TiltNet(dev)
    else:
        net = vrnet.VRNet(1344, vrnet.random_state_dict(vrnet.WIDTHS["nets"], seed=5), variant="nets", ctx=dev)
        net.offset = 8
    sep = VRSeparator(net, "4band_v2", agg=10, window_size=32, max_batch=2)
    inst_d, voc_d = sep.separate(on(dev, wave))
    mp = vo.MODEL_PARAMS["4band_v2"]
    X_d, he_d = sep.front.analyse(on(dev, wave))
    pred, _, _ = vrnet.vr_inference(net, X_d, {"value": 0.1, "split_bin": mp["band"][1]["crop_stop"]}, 32, False, 2)
    X, he = host(X_d), host(he_d)
    inst_o, voc_o = vo.back_end(X, host(pred), np.exp(1.0j * np.angle(X)), he, he.shape[1], mp)
    assert inst_d.shape == (2, 480 * 20) and float(np.max(np.abs(host(inst_d)))) > 1e-3
    s = max(1.0, float(np.max(np.abs(wave))))
    assert np.max(np.abs(host(inst_d) - inst_o.T)) < 2e-5 * s
    assert np.max(np.abs(host(voc_d) - voc_o.T)) < 2e-5 * s


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["17_HP-Wind_Inst-UVR.pth", "UVR-DeNoise-Lite.pth", "UVR-BVE-4B_SN-44100-1.pth"])
def test_engine_runs_vr_models_full_size(gpu_ctx, tmp_path, name):
    """the roster's VR entries at their real size (673 bins, window 512; seeded random-init weights): labels, lengths, and the
    size-independent property of the split -- the two output spectrograms sum to the mix's, and the back end without mirroring is
    linear, so the two stems sum to the band-split / recombined mix"""
    from audiolab_amd.engine import Separator
    from audiolab_amd.vr_frontend import VRFrontEnd
    sep = Separator(model_file_dir=str(tmp_path), ctx=gpu_ctx, allow_synthetic=True)
    sep.load_model(name)
    labels = sep.roster[name][2]["labels"]
    n = 44100 * 8 + 321
    wave = synth_mix(n, seed=17)
    out = sep.separate_array(wave)
    assert set(out) == set(labels)
    a, b = host(out[labels[0]]), host(out[labels[1]])
    assert a.shape == b.shape == (2, n) and np.isfinite(a).all() and np.isfinite(b).all()
    assert not np.any(a[:, 480 * (n // 480):]) and float(np.max(np.abs(a))) > 1e-3
    f = VRFrontEnd(sep.roster[name][1]["params"], gpu_ctx)
    X, _ = f.analyse(on(gpu_ctx, wave))
    whole = host(f.synthesise(X))
    assert np.max(np.abs(a[:, :whole.shape[1]] + b[:, :whole.shape[1]] - whole)) < 1e-4 * max(1.0, float(np.max(np.abs(whole))))
    # the recombined mix is the input minus what lies above the top band's crop (17.6 kHz; the oracle measures 25 % of this signal's
    # RMS there, 1.3 % with the high end put back) -- not a parity statement: a sanity bound on the whole chain
    mid = slice(4800, whole.shape[1] - 4800)
    assert np.sqrt(np.mean((whole[:, mid] - wave[:, :whole.shape[1]][:, mid]) ** 2)) < 0.35 * np.sqrt(np.mean(wave ** 2))
