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


def test_ola_runner_vs_oracle(emul):
    """Hann overlap-add chunker (unpinned upstream algorithm) against oracle/mdx_oracle.demix_ola, with a
    small random-init TFC-TDF network on both sides."""
    from audiolab_amd.mdx import OlaRunner
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    from oracle import mdx_oracle as mo
    from oracle import tdfnet_oracle
    cfg = TDFNetConfig(dim_f=64, dim_t=32, n_fft=256, hop=64, num_blocks=3, g=16)
    sd = synthetic_state_dict(cfg, seed=5)
    net = TDFNet(cfg, sd, ctx=emul, max_batch=3)
    g = mo.MDXGeometry(cfg.dim_f, cfg.dim_t, cfg.n_fft, cfg.hop)

    def run(spek):
        return tdfnet_oracle.forward(sd, torch.from_numpy(np.ascontiguousarray(spek, dtype=np.float32)), cfg.num_blocks, cfg.l, cfg.bn).numpy()
    for n, overlap, denoise in ((9000, 0.25, False), (5000, 0.75, True)):
        mix = synth_mix(n, seed=40 + n)
        want = mo.demix_ola(mix, g, run, overlap=overlap, denoise=denoise, zero_low_bins=3, compensate=1.035)
        got = OlaRunner(net, overlap=overlap, denoise=denoise, compensate=1.035, max_batch=4).demix(torch.from_numpy(mix)).numpy()
        assert got.shape == want.shape == (2, n)
        assert np.max(np.abs(got - want)) < 1e-4
