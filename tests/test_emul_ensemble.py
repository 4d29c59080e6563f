"""CPU (-m "not gpu"): ensemble ops (audiolab_amd/ensemble.py) over the emulated kernels against the
reference's outputs (tests/golden/ensemble.npz from stem_separator.py:173-262)."""
import os

import numpy as np
import pytest
import torch

from oracle import ensemble_oracle as eo
from oracle.toy import resid_case, synth_mix


def test_blend_tracks_vs_reference(emul, golden_dir):
    from audiolab_amd import ensemble
    z = np.load(os.path.join(golden_dir, "ensemble.npz"))
    tr = [torch.from_numpy(z[f"blend_t{i}"]) for i in range(3)]
    out = ensemble.blend_tracks(emul, tr, [float(w) for w in z["blend_w"]]).numpy()
    assert out.shape == z["blend_out"].shape
    assert np.max(np.abs(out - z["blend_out"])) < 1e-6
    assert abs(np.max(np.abs(out)) - 1.0) < 1e-6
    zero = ensemble.blend_tracks(emul, [torch.zeros(2, 100), torch.zeros(2, 90)], [1.0, 2.0])
    assert float(zero.abs().max()) == 0.0                     # all-zero blend stays zero (peak > 0 guard)


@pytest.mark.parametrize("tag", ["p", "m", "z", "big"])
def test_residual_subtract_vs_reference(emul, golden_dir, tag):
    from audiolab_amd import ensemble
    z = np.load(os.path.join(golden_dir, "ensemble.npz"))
    lag, gain = z[f"resid_{tag}_cfg"]
    base, comp = resid_case(int(lag), float(gain))
    out, params = ensemble.residual_subtract(emul, torch.from_numpy(base), torch.from_numpy(comp), 44100, return_params=True)
    out = out.numpy()
    ref = z[f"resid_{tag}_out"]
    cmp = out if tag == "p" else out[:, ::5]
    assert np.max(np.abs(cmp - ref)) < 2e-6
    _, want_params = eo.residual_subtract(base, comp, 44100, return_params=True)
    for (l0, a0), (l1, a1) in zip(params, want_params):
        assert l0 == l1 and abs(a0 - a1) < 1e-6


def test_debleed_matches_oracle(emul):
    from audiolab_amd import ensemble
    n = 50000
    voc = synth_mix(n, seed=11) * np.float32(0.4)
    other = synth_mix(n, seed=12) * np.float32(0.3)
    mix = (voc + other).astype(np.float32)
    for inst in ((other + 0.3 * voc).astype(np.float32),      # bleeding instrumental -> refinement accepted
                 other.copy(),                                 # clean instrumental -> rejected
                 np.zeros_like(other)):                        # silent -> residual fallback
        want, acc_w = eo.debleed(mix, voc, inst, 44100, 0.2)
        got, acc_g = ensemble.debleed(emul, torch.from_numpy(mix), torch.from_numpy(voc), torch.from_numpy(inst), 44100, 0.2)
        assert acc_g == acc_w
        assert np.max(np.abs(got.numpy() - want)) < 2e-6
