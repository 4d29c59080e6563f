"""Ensemble ops (audiolab_amd/ensemble.py -> alsep_axpby / peak_abs / scale_by_device / dot3 / xcorr_window / shift_subtract)
against the reference's outputs; every case runs on the emulated kernels (-m "not gpu") AND on the GPU (-m gpu) (tests/golden/ensemble.npz from stem_separator.py:173-262)."""
import os

import numpy as np
import pytest
import torch

from oracle import ensemble_oracle as eo
from oracle.toy import resid_case, synth_mix
from tests.conftest import host, on


def test_blend_tracks_vs_reference(dev, golden_dir):
    from audiolab_amd import ensemble
    z = np.load(os.path.join(golden_dir, "ensemble.npz"))
    tr = [on(dev, z[f"blend_t{i}"]) for i in range(3)]
    out = host(ensemble.blend_tracks(dev, tr, [float(w) for w in z["blend_w"]]))
    assert out.shape == z["blend_out"].shape
    assert np.max(np.abs(out - z["blend_out"])) < 1e-6
    assert abs(np.max(np.abs(out)) - 1.0) < 1e-6
    zero = ensemble.blend_tracks(dev, [on(dev, torch.zeros(2, 100)), on(dev, torch.zeros(2, 90))], [1.0, 2.0])
    assert float(zero.abs().max()) == 0.0                     # all-zero blend stays zero (peak > 0 guard)


@pytest.mark.parametrize("tag", ["p", "m", "z", "big"])
def test_residual_subtract_vs_reference(dev, golden_dir, tag):
    from audiolab_amd import ensemble
    z = np.load(os.path.join(golden_dir, "ensemble.npz"))
    lag, gain = z[f"resid_{tag}_cfg"]
    base, comp = resid_case(int(lag), float(gain))
    out, params = ensemble.residual_subtract(dev, on(dev, base), on(dev, comp), 44100, return_params=True)
    out = host(out)
    ref = z[f"resid_{tag}_out"]
    cmp = out if tag == "p" else out[:, ::5]
    assert np.max(np.abs(cmp - ref)) < 2e-6
    _, want_params = eo.residual_subtract(base, comp, 44100, return_params=True)
    for (l0, a0), (l1, a1) in zip(params, want_params):
        assert l0 == l1 and abs(a0 - a1) < 1e-6


def test_debleed_matches_oracle(dev):
    from audiolab_amd import ensemble
    n = 50000
    voc = synth_mix(n, seed=11) * np.float32(0.4)
    other = synth_mix(n, seed=12) * np.float32(0.3)
    mix = (voc + other).astype(np.float32)
    for inst in ((other + 0.3 * voc).astype(np.float32),      # bleeding instrumental -> refinement accepted
                 other.copy(),                                 # clean instrumental -> rejected
                 np.zeros_like(other)):                        # silent -> residual fallback
        want, acc_w = eo.debleed(mix, voc, inst, 44100, 0.2)
        got, acc_g = ensemble.debleed(dev, on(dev, mix), on(dev, voc), on(dev, inst), 44100, 0.2)
        assert acc_g == acc_w
        assert np.max(np.abs(host(got) - want)) < 2e-6


@pytest.mark.parametrize("sr_in,sr_out,n", [(48000, 44100, 4801), (44100, 48000, 3000), (22050, 44100, 1000)])
def test_resample_vs_oracle(dev, sr_in, sr_out, n):
    """alsep_resample (the 48 kHz -> 44.1 kHz input path of stem_separator.py:865) against its numpy restatement, and the property any
    band-limited resampler has: a sine well below both Nyquist rates comes out as the same sine at the new rate."""
    from audiolab_amd import ensemble
    from oracle import mdx_oracle as mo
    rng = np.random.default_rng(5)
    t = np.arange(n) / sr_in
    x = np.stack([0.5 * np.sin(2 * np.pi * 1000.0 * t), 0.1 * rng.standard_normal(n)]).astype(np.float32)
    got = host(ensemble.resample(dev, on(dev, x), sr_in, sr_out))
    want = mo.resample(x, sr_in, sr_out)
    assert got.shape == want.shape == (2, -(-n * sr_out // sr_in))
    assert np.max(np.abs(got - want)) < 2e-6
    m = np.arange(got.shape[1])
    inner = slice(200, got.shape[1] - 200)
    assert np.max(np.abs(got[0, inner] - 0.5 * np.sin(2 * np.pi * 1000.0 * m / sr_out)[inner])) < 2e-3
