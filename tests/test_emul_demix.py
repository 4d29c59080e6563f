"""CPU (-m "not gpu"): Predictor.demix (audiolab_amd/mdx.py) over the emulated kernels against the
reference's own demix outputs (tests/golden/demix.npz, produced by mdxnet.py:109-197)."""
import os
import types

import numpy as np
import pytest
import torch

from oracle.toy import synth_mix, toy_net, toy_net_affine

NETS = {"lin": toy_net, "aff": toy_net_affine}


class SpecFn:
    """ORT-shaped stand-in: run(None, {"input": spek}) -> [pred], reference layout."""

    def __init__(self, fn):
        self.fn = fn

    def run(self, _n, feed):
        return [torch.from_numpy(self.fn(feed["input"].numpy()))]


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_demix_small_vs_reference(emul, golden_dir, tag):
    from audiolab_amd.mdx import Predictor
    z = np.load(os.path.join(golden_dir, "demix.npz"))
    n_fft, hop, dta, dim_f = (int(v) for v in z["small_geom"])
    n, chunks, margin, denoise = (int(v) for v in z[f"small_{tag}_cfg"])
    args = types.SimpleNamespace(margin=margin, chunks=chunks, denoise=bool(denoise), dim_f=dim_f, dim_t=dta, n_fft=n_fft)
    pred = Predictor(args, SpecFn(NETS[str(z[f"small_{tag}_net"])]), ctx=emul, hop=hop, max_batch=5)
    mix = torch.from_numpy(synth_mix(n, seed=300 + n + chunks))
    out = pred.demix(mix).numpy()
    ref = z[f"small_{tag}_out"]
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref)) < 1e-5
