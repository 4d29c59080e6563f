"""CPU: the oracle (oracle/*.py) against the golden vectors produced by running the
reference (oracle/make_golden.py; /root/reference modules/rvc/infer/modules/uvr5/mdxnet.py
and modules/separator/stem_separator.py).  This is what pins the oracle."""
import os

import numpy as np
import pytest

from oracle import ensemble_oracle as eo
from oracle import mdx_oracle as mo
from oracle.toy import resid_case, synth_mix, toy_net, toy_net_affine

NETS = {"lin": toy_net, "aff": toy_net_affine}


def geom_of(arr):
    n_fft, hop, dta, dim_f = (int(v) for v in arr)
    return mo.MDXGeometry(dim_f=dim_f, dim_t=2 ** dta, n_fft=n_fft, hop=hop)


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


@pytest.mark.parametrize("name", ["p2", "p3", "p15", "full"])
def test_stft_istft_small(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "mdx_small.npz"))
    g = geom_of(z[f"{name}_geom"])
    spec = mo.stft(z[f"{name}_x"], g)
    assert spec.shape == z[f"{name}_spec"].shape
    assert relerr(spec, z[f"{name}_spec"]) < 2e-6          # reference is fp32 torch.stft
    y = mo.istft(z[f"{name}_spec"], g)
    assert y.shape == z[f"{name}_y"].shape
    assert np.max(np.abs(y - z[f"{name}_y"])) < 5e-6
    y2 = mo.istft(z[f"{name}_s2"], g)                      # arbitrary spectrogram
    assert relerr(y2, z[f"{name}_y2"]) < 5e-6


@pytest.mark.parametrize("name", ["n6144", "n7680", "n4096"])
def test_stft_istft_real_geometry(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "mdx_real.npz"))
    g = geom_of(z[f"{name}_geom"])
    x = np.random.default_rng(int(z[f"{name}_seed"])).standard_normal((2, 2, g.chunk_size)).astype(np.float32)
    spec = mo.stft(x, g)
    scale = np.max(np.abs(spec))
    assert np.max(np.abs(spec[:, :, :8, :4] - z[f"{name}_spec_lo"])) < 3e-6 * scale
    assert np.max(np.abs(spec[:, :, -8:, -4:] - z[f"{name}_spec_hi"])) < 3e-6 * scale
    assert np.max(np.abs(spec.reshape(-1)[z[f"{name}_spec_idx"]] - z[f"{name}_spec_val"])) < 3e-6 * scale
    assert abs(np.sqrt((spec ** 2).sum()) - float(z[f"{name}_spec_l2"])) < 1e-5 * float(z[f"{name}_spec_l2"])
    y = mo.istft(spec.astype(np.float32), g)
    assert np.max(np.abs(y.reshape(-1)[z[f"{name}_y_idx"]] - z[f"{name}_y_val"])) < 1e-5
    assert np.max(np.abs(y[:, :, :64] - z[f"{name}_y_head"])) < 1e-5
    assert np.max(np.abs(y[:, :, -64:] - z[f"{name}_y_tail"])) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e", "f"])
def test_demix_small(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "demix.npz"))
    g = geom_of(z["small_geom"])
    n, chunks, margin, denoise = (int(v) for v in z[f"small_{tag}_cfg"])
    mix = synth_mix(n, seed=300 + n + chunks)
    out = mo.demix(mix, g, NETS[str(z[f"small_{tag}_net"])], chunks=chunks, margin=margin, denoise=bool(denoise))
    ref = z[f"small_{tag}_out"]
    assert out.shape == ref.shape == (1, 2, n)
    assert np.max(np.abs(out - ref)) < 5e-6


@pytest.mark.parametrize("tag", ["r0", "r15"])
def test_demix_real_30s(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "demix.npz"))
    g = geom_of(z["real_geom"])
    n, chunks, margin, denoise = (int(v) for v in z[f"real_{tag}_cfg"])
    mix = synth_mix(n)
    out = mo.demix(mix, g, NETS[str(z[f"real_{tag}_net"])], chunks=chunks, margin=margin,
                   denoise=bool(denoise), dtype=np.float32)
    assert out.shape == (1, 2, n)
    tol = 2e-5
    assert np.max(np.abs(out.reshape(-1)[z[f"real_{tag}_idx"]] - z[f"real_{tag}_val"])) < tol
    assert np.max(np.abs(out[0, :, ::2003] - z[f"real_{tag}_strided"])) < tol
    gen = g.gen_size
    assert np.max(np.abs(out[0, :, gen - 64: gen + 64] - z[f"real_{tag}_seam"])) < tol
    assert np.max(np.abs(out[0, :, 15 * 44100 - 64: 15 * 44100 + 64] - z[f"real_{tag}_seg"])) < tol


def test_blend_tracks(golden_dir):
    z = np.load(os.path.join(golden_dir, "ensemble.npz"))
    out = eo.blend_tracks([z["blend_t0"], z["blend_t1"], z["blend_t2"]], list(z["blend_w"]))
    assert out.shape == z["blend_out"].shape
    assert np.max(np.abs(out - z["blend_out"])) < 1e-6
    assert abs(np.max(np.abs(out)) - 1.0) < 1e-6


@pytest.mark.parametrize("tag", ["p", "m", "z", "big"])
def test_residual_subtract(golden_dir, tag):
    z = np.load(os.path.join(golden_dir, "ensemble.npz"))
    lag, gain = z[f"resid_{tag}_cfg"]
    base, comp = resid_case(int(lag), float(gain))
    out, params = eo.residual_subtract(base, comp, 44100, return_params=True)
    ref = z[f"resid_{tag}_out"]
    cmp = out if tag == "p" else out[:, ::5]
    assert np.max(np.abs(cmp - ref)) < 2e-6
    if abs(lag) <= 529:                                    # +-12 ms @44.1k is the search range
        assert all(p[0] == int(lag) for p in params)


def test_envelope_ripple_7680():
    """n_fft/hop = 7.5 -> the interior envelope is not constant."""
    g = mo.MDXGeometry(dim_f=3072, dim_t=256, n_fft=7680)
    env = mo.window_envelope(g)[g.n_fft // 2 + g.trim: -(g.n_fft // 2 + g.trim)]
    assert (env.max() - env.min()) / env.mean() > 1e-4
    g2 = mo.MDXGeometry(dim_f=3072, dim_t=256, n_fft=6144)
    env2 = mo.window_envelope(g2)[g2.n_fft // 2 + g2.trim: -(g2.n_fft // 2 + g2.trim)]
    assert np.allclose(env2, 2.25, rtol=0, atol=1e-9)


def test_ola_identity_roundtrip():
    """OLA chunker (unpinned) with the identity network and full band reproduces the mix."""
    g = mo.MDXGeometry(dim_f=129, dim_t=32, n_fft=256, hop=64)
    mix = synth_mix(7000, seed=5)
    out = mo.demix_ola(mix, g, lambda s: s, overlap=0.25, zero_low_bins=0)
    assert out.shape == mix.shape
    assert np.max(np.abs(out - mix)) < 1e-6
