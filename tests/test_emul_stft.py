"""CPU (-m "not gpu"): the unchanged STFT/iSTFT kernel sources, run through the host emulation
(tests/cpu_emul), against the golden vectors of the reference and the oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import mdx_oracle as mo


def geom_of(arr):
    n_fft, hop, dta, dim_f = (int(v) for v in arr)
    return n_fft, hop, dta, dim_f


@pytest.mark.parametrize("name", ["p2", "p3", "p15", "full"])
def test_stft_istft_small_vs_golden(emul, golden_dir, name):
    from audiolab_amd.mdx import ConvTDFNetTrim
    z = np.load(os.path.join(golden_dir, "mdx_small.npz"))
    n_fft, hop, dta, dim_f = geom_of(z[f"{name}_geom"])
    net = ConvTDFNetTrim("cpu", "Conv-TDF", "vocals", 11, dim_f, dta, n_fft, hop=hop, ctx=emul)
    x = torch.from_numpy(z[f"{name}_x"])
    spec = net.stft(x).numpy()
    ref = z[f"{name}_spec"]
    assert spec.shape == ref.shape
    assert np.max(np.abs(spec - ref)) < 3e-6 * np.max(np.abs(ref))
    y = net.istft(torch.from_numpy(ref)).numpy()
    assert y.shape == z[f"{name}_y"].shape
    assert np.max(np.abs(y - z[f"{name}_y"])) < 1e-5
    y2 = net.istft(torch.from_numpy(z[f"{name}_s2"])).numpy()
    assert np.max(np.abs(y2 - z[f"{name}_y2"])) < 1e-5 * max(1.0, np.max(np.abs(z[f"{name}_y2"])))


def test_layout_convert_and_nhwc_roundtrip(emul):
    from audiolab_amd import _lib
    from audiolab_amd.mdx import StftPlan
    plan = StftPlan(emul, 384, 64, 160, 16)
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((2, 2, plan.chunk_size)).astype(np.float32))
    ref = plan.stft_strided(x, plan.chunk_size, 2 * plan.chunk_size, 2, torch.float32, _lib.LAYOUT_REF)
    nhwc = plan.stft_strided(x, plan.chunk_size, 2 * plan.chunk_size, 2, torch.float32, _lib.LAYOUT_NHWC)
    assert torch.equal(plan.convert(ref, _lib.LAYOUT_REF), nhwc)
    assert torch.equal(plan.convert(nhwc, _lib.LAYOUT_NHWC), ref)
    assert torch.equal(nhwc.permute(0, 3, 2, 1).contiguous(), ref)
    bf = plan.stft_strided(x, plan.chunk_size, 2 * plan.chunk_size, 2, torch.bfloat16, _lib.LAYOUT_NHWC)
    assert torch.equal(bf, nhwc.to(torch.bfloat16))
    out = emul.empty((2, 2, plan.chunk_size))
    plan.istft_strided(nhwc, _lib.LAYOUT_NHWC, out, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size,
                       3 * plan.chunk_size)
    g = mo.MDXGeometry(160, 16, 384, 64)
    want = mo.istft(ref.numpy(), g)
    assert np.max(np.abs(out.numpy() - want)) < 1e-5


def test_unsupported_geometry_raises(emul):
    from audiolab_amd._lib import AlsepError
    from audiolab_amd.mdx import StftPlan
    with pytest.raises(AlsepError):
        StftPlan(emul, 1000, 64, 96, 16)          # no FFT kernel for n_fft=1000
    with pytest.raises(AlsepError):
        StftPlan(emul, 256, 64, 96, 2)            # chunk <= n_fft/2: reflect padding undefined


def test_istft_register_ring_hop1024(emul):
    """hop = 1024 geometries take the register-ring iSTFT kernel: check it (plain and stitched stores)
    against the oracle, including an arbitrary (non-consistent) spectrogram."""
    from audiolab_amd import _lib
    from audiolab_amd.mdx import StftPlan
    plan = StftPlan(emul, 2048, 1024, 640, 20)
    g = mo.MDXGeometry(640, 20, 2048, 1024)
    rng = np.random.default_rng(11)
    spec = torch.from_numpy(rng.standard_normal((3, 4, 640, 20)).astype(np.float32))
    want = mo.istft(spec.numpy(), g)
    nhwc = plan.convert(spec, _lib.LAYOUT_REF)
    out = emul.empty((3, 2, plan.chunk_size))
    plan.istft_strided(nhwc, _lib.LAYOUT_NHWC, out, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size, 5 * plan.chunk_size)
    assert np.max(np.abs(out.numpy() - want)) < 2e-5 * max(1.0, np.max(np.abs(want)))
    # stitched: keep [trim, chunk-trim), windows abut every gen samples, output cut at `limit`
    trim, gen = plan.trim, plan.gen_size
    limit = 2 * gen + 777
    st = emul.zeros((2, 3 * gen))
    plan.istft_strided(nhwc, _lib.LAYOUT_NHWC, st, 3 * gen, gen, trim, plan.chunk_size - trim, limit)
    ref = want[:, :, trim:-trim].transpose(1, 0, 2).reshape(2, -1)
    assert np.max(np.abs(st.numpy()[:, :limit] - ref[:, :limit])) < 2e-5 * max(1.0, np.max(np.abs(want)))
    assert float(st[:, limit:].abs().max()) == 0.0


@pytest.mark.parametrize("n_fft,dim_f,dim_t", [(6144, 3072, 8), (6144, 3073, 7), (6144, 1000, 8), (4096, 2049, 6), (4096, 2048, 9),
                                               (7680, 3072, 10), (7680, 3841, 10)])
def test_three_pass_kernels_production_sizes(emul, n_fft, dim_f, dim_t):
    """n_fft 4096 / 6144 (and, for the forward transform, 7680 = 16*16*30 with padded rows) with hop 1024 take the three-pass kernels of fft_r16.h (radix 16,16,R2 forward with the
    in-thread two-for-one split; radix R2,16,16 inverse with the in-thread Hermitian extension and the register
    overlap-add).  Small dim_t keeps both the reflect-padded edge frames and interior frames in play; dim_f covers
    the full band (Nyquist bin), the production band (n_fft/2) and a narrow band (zero-filled bins)."""
    from audiolab_amd import _lib
    from audiolab_amd.mdx import StftPlan
    plan = StftPlan(emul, n_fft, 1024, dim_f, dim_t)
    g = mo.MDXGeometry(dim_f, dim_t, n_fft, 1024)
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 2, plan.chunk_size)).astype(np.float32)
    want = mo.stft(x, g)
    ref = plan.stft_strided(torch.from_numpy(x), plan.chunk_size, 2 * plan.chunk_size, 2, torch.float32, _lib.LAYOUT_REF)
    nhwc = plan.stft_strided(torch.from_numpy(x), plan.chunk_size, 2 * plan.chunk_size, 2, torch.float32, _lib.LAYOUT_NHWC)
    assert np.max(np.abs(ref.numpy() - want)) < 3e-6 * np.max(np.abs(want))
    assert torch.equal(nhwc.permute(0, 3, 2, 1).contiguous(), ref)
    bf = plan.stft_strided(torch.from_numpy(x), plan.chunk_size, 2 * plan.chunk_size, 2, torch.bfloat16, _lib.LAYOUT_NHWC)
    assert torch.equal(bf, nhwc.to(torch.bfloat16))
    # inverse on an arbitrary (non-consistent) spectrogram, plain and stitched stores
    spec = rng.standard_normal((2, 4, dim_f, dim_t)).astype(np.float32)
    want_y = mo.istft(spec, g)
    tol = 2e-5 * max(1.0, float(np.max(np.abs(want_y))))
    for layout in (_lib.LAYOUT_REF, _lib.LAYOUT_NHWC):
        sp = torch.from_numpy(spec)
        if layout == _lib.LAYOUT_NHWC:
            sp = plan.convert(sp, _lib.LAYOUT_REF)
        out = emul.empty((2, 2, plan.chunk_size))
        plan.istft_strided(sp, layout, out, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size, 3 * plan.chunk_size)
        assert np.max(np.abs(out.numpy() - want_y)) < tol
    trim, gen = plan.trim, plan.gen_size
    if gen > 0:
        limit = gen + 777
        st = emul.zeros((2, 2 * gen))
        plan.istft_strided(plan.convert(torch.from_numpy(spec), _lib.LAYOUT_REF), _lib.LAYOUT_NHWC, st, 2 * gen, gen, trim,
                           plan.chunk_size - trim, limit)
        want_s = want_y[:, :, trim:-trim].transpose(1, 0, 2).reshape(2, -1)
        assert np.max(np.abs(st.numpy()[:, :limit] - want_s[:, :limit])) < tol
        assert float(st[:, limit:].abs().max()) == 0.0


@pytest.mark.parametrize("n_fft,dim_f,dim_t", [(5120, 2560, 8), (16384, 2048, 20)])
def test_roster_geometries_5120_16384_vs_oracle(emul, n_fft, dim_f, dim_t):
    """n_fft of UVR-MDX-NET_Crowd_HQ_1 (5120 = 5 * 2**10) and of kuielab_a_bass.onnx (16384, the alt-bass model of
    stem_separator.py:512) at hop 1024: STFT and iSTFT (register-ring kernel, one frame in LDS) against the oracle."""
    from audiolab_amd import _lib
    from audiolab_amd.mdx import StftPlan
    plan = StftPlan(emul, n_fft, 1024, dim_f, dim_t)
    g = mo.MDXGeometry(dim_f, dim_t, n_fft, 1024)
    rng = np.random.default_rng(23)
    x = rng.standard_normal((1, 2, plan.chunk_size)).astype(np.float32)
    want = mo.stft(x, g)
    got = plan.stft_strided(torch.from_numpy(x), plan.chunk_size, 2 * plan.chunk_size, 1, torch.float32, _lib.LAYOUT_REF).numpy()
    assert np.max(np.abs(got - want)) < 3e-6 * np.max(np.abs(want))
    spec = rng.standard_normal((1, 4, dim_f, dim_t)).astype(np.float32)
    want_y = mo.istft(spec, g)
    out = emul.empty((1, 2, plan.chunk_size))
    plan.istft_strided(torch.from_numpy(spec), _lib.LAYOUT_REF, out, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size, plan.chunk_size)
    assert np.max(np.abs(out.numpy() - want_y)) < 2e-5 * max(1.0, float(np.max(np.abs(want_y))))
