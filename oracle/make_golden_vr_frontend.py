#!/usr/bin/env python3
"""Golden vectors for the VR multi-band front / back end (TEST INFRASTRUCTURE).

Imports the REFERENCE module modules/rvc/infer/lib/uvr5_pack/lib_v5/spec_utils.py in this container with ``librosa`` replaced by a stub
(librosa, resampy, samplerate are not in the image): ``stft`` / ``istft`` are the primitives of oracle/vr_oracle.py; ``resample`` does
what librosa.resample does for the kinds that run -- equal rates: the input; "polyphase": scipy.signal.resample_poly; "scipy":
scipy.signal.resample -- with SCIPY ITSELF (a dependency of the reference, importable here), and raises for any other kind;
runs the reference's own ``wave_to_spectrogram``, ``combine_spectrograms``, ``mirroring`` and ``cmb_spectrogram_to_wave`` on a seeded
signal with the reference's 4band_v2 parameter file, and records the results: tests/golden/vr_frontend.npz.  What these vectors pin is
everything the reference does AROUND those primitives (cropping, stacking, pre-filter gains, filter ramps, mirroring, band recombination
and its resampling chain).

    python oracle/make_golden_vr_frontend.py       (only where /root/reference exists)
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import vr_oracle as vo  # noqa: E402
from oracle.toy import synth_mix  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "vr_frontend.npz")


def load_ref():
    lib = types.ModuleType("librosa")
    lib.stft = lambda y, n_fft=2048, hop_length=None, **kw: vo.stft(np.asarray(y), n_fft, hop_length)
    lib.istft = lambda stft_matrix, hop_length=None, **kw: vo.istft(np.asarray(stft_matrix), hop_length)

    def resample(y, orig_sr, target_sr, res_type="soxr_hq", **kw):               # librosa/core/audio.py resample, the reachable branches
        import math
        import scipy.signal
        y = np.asarray(y)
        if orig_sr == target_sr:
            return y
        ratio = float(target_sr) / orig_sr
        n_samples = int(np.ceil(y.shape[-1] * ratio))
        if res_type in ("scipy", "fft"):
            y_hat = scipy.signal.resample(y, n_samples, axis=-1)
        elif res_type == "polyphase":
            g = math.gcd(int(orig_sr), int(target_sr))
            y_hat = scipy.signal.resample_poly(y, int(target_sr) // g, int(orig_sr) // g, axis=-1)
        else:
            raise RuntimeError(f"librosa stub: resampler kind {res_type!r} needs a package that is not in the image")
        return np.asarray(y_hat, dtype=y.dtype)
    lib.resample = resample
    sys.modules["librosa"] = lib
    sys.modules.setdefault("soundfile", types.ModuleType("soundfile"))
    pkg_dir = os.path.join(REF, "modules/rvc/infer/lib/uvr5_pack/lib_v5")
    pkg = types.ModuleType("ref_lib_v5")
    pkg.__path__ = [pkg_dir]
    sys.modules["ref_lib_v5"] = pkg
    su = importlib.import_module("ref_lib_v5.spec_utils")
    mpi = importlib.import_module("ref_lib_v5.model_param_init")
    return su, mpi


def main():
    su, mpi = load_ref()
    mp = mpi.ModelParameters(os.path.join(REF, "modules/rvc/infer/lib/uvr5_pack/lib_v5/modelparams/4band_v2.json"))
    P = mp.param
    mine = vo.MODEL_PARAMS["4band_v2"]
    for key in ("bins", "pre_filter_start", "pre_filter_stop", "sr"):
        assert P[key] == mine[key]
    for d in P["band"]:
        for k, v in mine["band"][d].items():
            assert P["band"][d][k] == v, (d, k)
    n = 44100 * 2 + 123
    wave = synth_mix(n, seed=4242)
    bands_n = len(P["band"])
    X_wave, X_spec_s = {}, {}
    for d in range(bands_n, 0, -1):                             # vr.py:55-96 with the reference's own functions
        bp = P["band"][d]
        X_wave[d] = wave if d == bands_n else su.librosa.resample(X_wave[d + 1], orig_sr=P["band"][d + 1]["sr"], target_sr=bp["sr"],
                                                                  res_type=bp["res_type"])
        X_spec_s[d] = su.wave_to_spectrogram(X_wave[d], bp["hl"], bp["n_fft"], P["mid_side"], P["mid_side_b2"], P["reverse"])
        if d == bands_n:
            hh = (bp["n_fft"] // 2 - bp["crop_stop"]) + (P["pre_filter_stop"] - P["pre_filter_start"])
            high_end = X_spec_s[d][:, bp["n_fft"] // 2 - hh: bp["n_fft"] // 2, :]
    X = su.combine_spectrograms(X_spec_s, mp)
    mine_X, mine_he, mine_hh = vo.front_end(wave, mine)
    assert mine_hh == hh and np.max(np.abs(mine_X - X)) < 1e-6 * np.max(np.abs(X)) and np.array_equal(mine_he, high_end)
    # a deterministic "network": keep 70 % of the magnitude, tilted along frequency
    mag, phase = np.abs(X), np.exp(1.0j * np.angle(X))
    pred = (mag * (0.4 + 0.5 * np.linspace(0, 1, X.shape[1])[None, :, None])).astype(np.float32)
    y_spec = pred * phase
    v_spec = X - y_spec
    outs = {}

    class ZeroedAlloc:
        """spec_utils.cmb_spectrogram_to_wave allocates every band's spectrogram with ``np.ndarray(shape, dtype=complex)`` (:360-362):
        UNINITIALISED memory.  Its filters zero every bin it does not fill except the Nyquist bin of the top band, which therefore holds
        whatever the allocator returns (measured here: a 2e-4 component at 0.998 x Nyquist once freed blocks get reused).  For deterministic
        vectors the golden run hands that one call zero-filled memory; everything else is numpy itself."""

        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def ndarray(shape, dtype=float):
            return np.zeros(shape, dtype=dtype)
    su.np = ZeroedAlloc()
    for tag, spec in (("inst", y_spec), ("voc", v_spec)):
        he = su.mirroring("mirroring", spec, high_end, mp)
        assert np.allclose(he, vo.mirroring(spec, high_end, mine))
        w = np.asarray(su.cmb_spectrogram_to_wave(spec, mp, hh, he), dtype=np.float32)        # [n, 2]
        w_mine = vo.cmb_spectrogram_to_wave(spec, mine, hh, he)
        err = float(np.max(np.abs(w - w_mine)))
        assert err < 1e-5, (tag, err)
        outs[f"{tag}_wave"] = w
        outs[f"{tag}_he_abs_sum"] = np.array([np.abs(he).sum()], dtype=np.float64)
    idx = np.random.default_rng(7).integers(0, X.size, 4000)
    np.savez_compressed(OUT, n=np.array([n]), seed=np.array([4242]), X_shape=np.array(X.shape), X_idx=idx,
                        X_val=X.reshape(-1)[idx], X_l2=np.array([np.sqrt((np.abs(X) ** 2).sum())]), hh=np.array([hh]),
                        inst_wave=outs["inst_wave"][::7], voc_wave=outs["voc_wave"][::7], inst_len=np.array([len(outs["inst_wave"])]),
                        inst_he_abs_sum=outs["inst_he_abs_sum"], voc_he_abs_sum=outs["voc_he_abs_sum"])
    print("wrote", OUT, os.path.getsize(OUT), "bytes; X", X.shape, "wave", outs["inst_wave"].shape)


if __name__ == "__main__":
    main()
