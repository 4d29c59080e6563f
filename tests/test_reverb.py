"""Reverb impulse-response extraction (audiolab_amd/reverb.py -> csrc/reverb.hip) on the emulated kernels (-m "not gpu") and on the GPU
(-m gpu), same bodies: against tests/golden/reverb.npz -- outputs of the reference's own ``handlers/reverb.py`` functions, produced by
oracle/make_golden_reverb.py -- and against the float64 mode of the oracle restatement.

Tolerances.  The reference's environment (numpy 2.x) runs these FFTs in single precision; the HIP path computes in double.  So the
golden impulse responses are matched to 1e-6 of their peak (measured distance between the reference's float32 evaluation and an exact
one: ~1e-8 of the peak at these sizes), the float64 oracle to 1e-10, the integer correlation peak exactly."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import reverb_oracle as ro
from oracle.reverb_cases import CASES, make_case
from tests.conftest import host, on

SCALARS = ("sample_rate", "pre_delay", "decay_time", "early_reflection_ratio", "late_reverb_ratio", "diffusion", "spectral_centroid")


def chan_major(x: np.ndarray) -> np.ndarray:
    """the reference's [N, C] / [N] arrays -> this build's [C, N]"""
    return np.ascontiguousarray(x.T if x.ndim == 2 else x[None])


@pytest.mark.parametrize("n,inverse", [(1, 0), (2, 0), (8, 1), (64, 0), (4096, 1), (3, 0), (100, 1), (6001, 0), (12289, 1), (88200, 0)])
def test_dft_f64_any_length_vs_numpy(dev, n, inverse):
    """alsep_dft_f64: radix 8 / 4 / 2 Stockham passes for powers of two, Bluestein otherwise, both directions"""
    import ctypes as C
    from audiolab_amd import _lib
    if dev.device.type == "cpu" and n > 20000:
        pytest.skip("emulated suite keeps the small transforms")
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    xin = on(dev, np.ascontiguousarray(np.stack([x.real, x.imag], -1)))
    out = torch.empty_like(xin)
    need = int(dev.lib.alsep_dft_f64_workspace_bytes(n))
    ws = torch.empty((need,), dtype=torch.uint8, device=dev.device)
    dev.check(dev.lib.alsep_dft_f64(dev.handle, _lib.ptr(xin), _lib.ptr(out), n, inverse, _lib.ptr(ws), need), "alsep_dft_f64")
    got = host(out)
    got = got[:, 0] + 1j * got[:, 1]
    want = np.fft.ifft(x) * n if inverse else np.fft.fft(x)
    assert np.max(np.abs(got - want)) < 1e-11 * max(1.0, np.max(np.abs(want)))
    _ = C


@pytest.mark.parametrize("name", list(CASES))
def test_xcorr_peak_and_values_vs_reference(dev, golden_dir, name):
    """fft_xcorr (reverb.py:55-66) + argmax: the index the reference found, and the correlation at 64 probes it stored"""
    from audiolab_amd import reverb
    z = np.load(os.path.join(golden_dir, "reverb.npz"))
    dry, wet, sr = make_case(name)
    if dev.device.type == "cpu" and len(wet) > 60000:
        pytest.skip("emulated suite keeps the short cases")
    d, w = on(dev, chan_major(dry)), on(dev, chan_major(wet))
    need = int(dev.lib.alsep_reverb_workspace_bytes(w.shape[1], d.shape[1]))
    ws = torch.empty((need,), dtype=torch.uint8, device=dev.device)
    n_corr = w.shape[1] + d.shape[1] - 1
    probes = np.linspace(0, n_corr - 1, 64).astype(np.int64)
    idx, vals = reverb.xcorr_argmax(dev, w, d, ws, probe_idx=probes)
    assert idx == int(z[f"{name}_corr_argmax"][0])
    want = z[f"{name}_corr_probe"]
    assert np.max(np.abs(vals - want)) < 2e-6 * np.max(np.abs(want)) + 1e-7     # the reference's values are float32


@pytest.mark.parametrize("name", list(CASES))
def test_extract_reverb_vs_reference_and_oracle(dev, golden_dir, name, tmp_path):
    from audiolab_amd import reverb
    z = np.load(os.path.join(golden_dir, "reverb.npz"))
    dry, wet, sr = make_case(name)
    if dev.device.type == "cpu" and len(wet) > 60000:
        pytest.skip("emulated suite keeps the short cases")
    out = str(tmp_path / "impulse_response.ir")
    assert reverb.extract_reverb(on(dev, chan_major(dry)), on(dev, chan_major(wet)), out, sr=sr, ctx=dev) == out
    with open(out) as f:
        text = f.read()
    p = json.loads(text)
    assert list(p) == list(SCALARS) + ["impulse_response"]                    # keys and their order as reverb.py:159-168
    assert text.startswith('{\n  "sample_rate": ')                            # indent=2 (:40)
    ir = np.asarray(p["impulse_response"])
    # (1) the reference's own outputs
    g = z[f"{name}_scalars"]
    assert len(ir) == int(z[f"{name}_ir_len"][0])
    assert p["sample_rate"] == int(g[0]) and p["pre_delay"] == g[1]
    peak = float(np.max(np.abs(z[f"{name}_ir_head"])))
    assert np.max(np.abs(ir[:4096] - z[f"{name}_ir_head"])) < 1e-6 * peak
    assert np.max(np.abs(ir[z[f"{name}_ir_idx"]] - z[f"{name}_ir_val"])) < 1e-6 * peak
    assert abs(np.sqrt(np.sum(ir ** 2)) - float(z[f"{name}_ir_l2"][0])) < 1e-5 * float(z[f"{name}_ir_l2"][0])
    for k, want in zip(SCALARS[3:], g[3:]):                                   # early_reflection_ratio ... spectral_centroid
        assert abs(p[k] - want) <= 2e-5 * abs(want), (k, p[k], want)
    # decay_time: the curve the fit runs on (float32, as the reference computes it) to an ulp of its largest values, the host fit itself
    # exactly on the reference's curve, and end to end wherever the fit is determined at all (case "ragged" is not: its rate b ~ 1e-4
    # sits in a flat direction of the least-squares problem, where one flipped float32 rounding of the curve moves 3 / b by 10 %)
    env_ref = ro.envelope_db(wet)
    env = host(reverb.envelope_db(dev, on(dev, chan_major(wet))))
    assert env.dtype == np.float32 and np.max(np.abs(env - env_ref)) <= 4e-5
    assert reverb.fit_decay(env_ref, sr, 5000) == g[2]
    if g[2] < 60.0:
        assert abs(p["decay_time"] - g[2]) <= 1e-4 * g[2], (p["decay_time"], g[2])
    # (2) the exact evaluation (float64 oracle)
    exact = ro.extract_reverb(dry, wet, sr, precision=np.float64)
    e_ir = np.asarray(exact["impulse_response"])
    assert np.max(np.abs(ir - e_ir)) < 1e-10 * max(1.0, float(np.max(np.abs(e_ir))))
    for k in (SCALARS[1],) + SCALARS[3:]:
        assert abs(p[k] - exact[k]) <= 1e-6 * abs(exact[k]) + 1e-12, (k, p[k], exact[k])


def test_sample_rate_mismatch_and_paths(dev, tmp_path):
    """file inputs (the reference's calling convention) and its ValueError on differing sample rates (:121-122)"""
    from audiolab_amd import reverb, wavio
    dry, wet, sr = make_case("odd_stereo")
    wavio.write_wav(str(tmp_path / "dry.wav"), chan_major(dry), sr)
    wavio.write_wav(str(tmp_path / "wet.wav"), chan_major(wet), sr)
    wavio.write_wav(str(tmp_path / "wet2.wav"), chan_major(wet), sr * 2)
    a = reverb.extract_reverb_params(str(tmp_path / "dry.wav"), str(tmp_path / "wet.wav"), ctx=dev)
    b = reverb.extract_reverb_params(on(dev, chan_major(dry)), on(dev, chan_major(wet)), sr=sr, ctx=dev)
    assert a == b
    with pytest.raises(ValueError):
        reverb.extract_reverb_params(str(tmp_path / "dry.wav"), str(tmp_path / "wet2.wav"), ctx=dev)


def test_transform_chain_stores_the_impulse_response(dev, tmp_path):
    """stem_separator.py:822-829: de-reverb on the vocals with store_reverb_ir -> <stems>/impulse_response.ir between the model's
    "No Reverb" (chosen) and "Reverb" (other) outputs; nothing is written for other stems or with the option off"""
    from audiolab_amd.engine import Separator
    from audiolab_amd.separator.stem_separator import EnsembleDemucsMDXMusicSeparationModel as Model
    from audiolab_amd.tdfnet import TDFNetConfig
    from oracle.toy import synth_mix
    cfg = TDFNetConfig(dim_f=64, dim_t=32, n_fft=256, hop=64, num_blocks=3, g=16)
    roster = {Model.REVERB_MODEL: ("No Reverb", "Reverb", cfg)}
    eng = Separator(ctx=dev, use_autocast=False, allow_synthetic=True, roster=roster, max_batch=2)
    vocals = on(dev, synth_mix(6000, seed=3))
    folder = tmp_path / "stems"
    folder.mkdir()
    model = Model({"reverb_removal": "Main Vocals", "store_reverb_ir": True}, separator=eng)
    out = model._apply_transform_chain(vocals, "song", "vocals", output_folder=str(folder), sr=44100)
    path = folder / "impulse_response.ir"
    assert path.exists()
    p = json.loads(path.read_text())
    assert list(p) == list(SCALARS) + ["impulse_response"] and p["sample_rate"] == 44100 and len(p["impulse_response"]) == 6000
    # the file is what extract_reverb yields for (chosen = the stage's output, other = input - output)
    eng.load_model(Model.REVERB_MODEL)
    stems = eng.separate_array(vocals)
    assert torch.equal(out, stems["No Reverb"])
    want = ro.extract_reverb(host(stems["No Reverb"]).T, host(stems["Reverb"]).T, 44100, precision=np.float64)
    assert np.max(np.abs(np.asarray(p["impulse_response"]) - np.asarray(want["impulse_response"]))) < 1e-9
    path.unlink()
    Model({"reverb_removal": "All", "store_reverb_ir": True}, separator=eng)._apply_transform_chain(vocals, "song", "instrumental",
                                                                                                    output_folder=str(folder), sr=44100)
    Model({"reverb_removal": "Main Vocals", "store_reverb_ir": False}, separator=eng)._apply_transform_chain(vocals, "song", "vocals",
                                                                                                             output_folder=str(folder), sr=44100)
    assert not path.exists()


@pytest.mark.gpu
def test_full_length_track_recovers_a_known_response(gpu_ctx):
    """BASELINE track size (5 min at 44.1 kHz, 13.23 M samples -- not a power of two: the Wiener step runs Bluestein on 2^25 points,
    the correlation a 2^25-point transform): with wet = a sparse circular filter applied to a white dry signal the deconvolution must
    return that filter (|H|^2 >> eps everywhere), and the correlation peak sits at the filter's dominant tap."""
    import time
    from audiolab_amd import reverb
    n, sr = 13230000, 44100
    g = torch.Generator(device="cuda").manual_seed(11)
    dry = torch.randn((2, n), device="cuda", generator=g) * 0.1
    taps = {0: 0.25, 441: 1.0, 2205: -0.5, 30000: 0.3, 88000: 0.125}        # all inside the 2 s the reference keeps
    wet = torch.zeros_like(dry)
    for lag, gain in taps.items():
        wet += gain * torch.roll(dry, lag, dims=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    need = int(gpu_ctx.lib.alsep_reverb_workspace_bytes(n, n))
    ws = torch.empty((need,), dtype=torch.uint8, device="cuda")
    peak = reverb.xcorr_argmax(gpu_ctx, wet, dry, ws)
    ir = reverb.wiener_ir(gpu_ctx, wet, dry, 1e-6, 2 * sr, ws)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert peak == 441
    ir = ir.cpu().numpy()
    assert ir.shape == (2 * sr,)
    want = np.zeros(2 * sr)
    for lag, gain in taps.items():
        want[lag] = gain
    err = float(np.max(np.abs(ir - want)))
    print(f"reverb at 13.23 M samples: correlation + deconvolution in {dt * 1e3:.0f} ms, max|ir - filter| = {err:.2e}")
    assert err < 1e-6
