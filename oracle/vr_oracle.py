"""numpy restatement of the VR models' multi-band front / back end (TEST INFRASTRUCTURE -- see ``oracle/__init__.py``).

Follows the in-tree reference /root/reference/modules/rvc/infer/lib/uvr5_pack/lib_v5/spec_utils.py (``wave_to_spectrogram`` :30-56,
``combine_spectrograms`` :95-125, ``cmb_spectrogram_to_wave`` :353-429, ``fft_lp_filter`` / ``fft_hp_filter`` :432-451, ``mirroring``
:453-490) and the driver modules/rvc/infer/modules/uvr5/vr.py:43-196 (``AudioPre._path_audio_``).

PINNED (tests/golden/vr_frontend.npz, oracle/make_golden_vr_frontend.py): the band cropping / stacking, the pre-filter gains, the low-pass /
high-pass ramps, mirroring and the band recombination are checked against the reference's own functions, imported here with ``librosa``
replaced by a stub: ``stft`` / ``istft`` are the primitives below, ``resample`` dispatches to the scipy routines librosa itself calls.
  * ``stft`` / ``istft`` -- UNPINNED (librosa is not in the image): librosa >= 0.10 semantics (requirements.txt:47 asks >= 0.11):
    periodic Hann window, ``center=True`` with ``pad_mode="constant"``, inverse normalised by the summed squared window, length
    ``hop * (frames - 1)``;
  * ``resample`` -- the reference's text names four kinds.  The two that run in the 4-band sets are restated here and PINNED against
    scipy itself (tests/test_vr_frontend.py::test_resamplers_vs_scipy): "polyphase" (the parameter files' ``res_type`` for the chain
    going down) = scipy.signal.resample_poly, "scipy" (the chain going up, spec_utils.py:427) = scipy.signal.resample.  The other two
    never resample in these sets: "sinc_fastest" (:404-414) sits between bands 1 and 2, both 7350 Hz, and the top band's "kaiser_fast"
    is librosa.load's kind for a file that is not at 44.1 kHz (vr.py:62-70; the engine brings its input to 44.1 kHz beforehand).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np


# model parameter sets of the reference (lib_v5/modelparams/4band_v2.json, 4band_v3.json): hyper-parameters of the published models
_BANDS_4 = {
    1: dict(sr=7350, hl=80, n_fft=640, crop_start=0, crop_stop=85, lpf_start=25, lpf_stop=53, res_type="polyphase"),
    2: dict(sr=7350, hl=80, n_fft=320, crop_start=4, crop_stop=87, hpf_start=25, hpf_stop=12, lpf_start=31, lpf_stop=62, res_type="polyphase"),
    3: dict(sr=14700, hl=160, n_fft=512, crop_start=17, crop_stop=216, hpf_start=48, hpf_stop=24, lpf_start=139, lpf_stop=210,
            res_type="polyphase"),
    4: dict(sr=44100, hl=480, n_fft=960, crop_start=78, crop_stop=383, hpf_start=130, hpf_stop=86, res_type="kaiser_fast"),
}
# "_sn": modelparams/4band_v2_sn.json differs from 4band_v2.json by ``"convert_channels": "stereo_n"`` on band 4 (:48); 4band_v3_sn (the
# set of UVR-BVE-4B_SN-44100-1.pth, stem_separator.py:752) is not in the reference tree and is taken as 4band_v3 + the same line.  The
# conversion itself is not in the tree's spec_utils either: it is the rule of audio-separator's VR spec_utils (``convert_channels``,
# ``spectrogram_to_wave``), restated from memory -- upstream, uncited; PARITY UNPINNED for the "_sn" sets (no vector of the reference).
_BANDS_4_SN = {**_BANDS_4, 4: dict(_BANDS_4[4], convert_channels="stereo_n")}
MODEL_PARAMS = {
    "4band_v2": dict(bins=672, unstable_bins=8, reduction_bins=637, band=_BANDS_4, sr=44100, pre_filter_start=668, pre_filter_stop=672),
    "4band_v3": dict(bins=672, unstable_bins=8, reduction_bins=530, band=_BANDS_4, sr=44100, pre_filter_start=668, pre_filter_stop=672),
    "4band_v2_sn": dict(bins=672, unstable_bins=8, reduction_bins=637, band=_BANDS_4_SN, sr=44100, pre_filter_start=668, pre_filter_stop=672),
    "4band_v3_sn": dict(bins=672, unstable_bins=8, reduction_bins=530, band=_BANDS_4_SN, sr=44100, pre_filter_start=668, pre_filter_stop=672),
}


def stft(x: np.ndarray, n_fft: int, hop: int) -> np.ndarray:
    """[n] float32 -> complex64 [n_fft/2+1, 1 + n // hop] (librosa.stft, center, zero padding, periodic Hann)"""
    x = np.asarray(x, dtype=np.float32)
    w = (0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)).astype(np.float32)
    xp = np.pad(x, (n_fft // 2, n_fft // 2))
    n_frames = 1 + len(x) // hop
    frames = np.stack([xp[t * hop: t * hop + n_fft] * w for t in range(n_frames)], axis=1)
    return np.fft.rfft(frames.astype(np.float64), axis=0).astype(np.complex64)


def istft(z: np.ndarray, hop: int) -> np.ndarray:
    """complex [n_fft/2+1, T] -> float32 [hop * (T - 1)] (librosa.istft, center, window-sum-square normalisation)"""
    n_fft = 2 * (z.shape[0] - 1)
    T = z.shape[1]
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_fft) / n_fft)
    y = np.zeros(n_fft + hop * (T - 1))
    env = np.zeros_like(y)
    frames = np.fft.irfft(z.astype(np.complex128), n=n_fft, axis=0)
    for t in range(T):
        y[t * hop: t * hop + n_fft] += frames[:, t] * w
        env[t * hop: t * hop + n_fft] += w * w
    nz = env > np.finfo(np.float32).tiny
    y[nz] /= env[nz]
    return y[n_fft // 2: n_fft // 2 + hop * (T - 1)].astype(np.float32)


def _kaiser(n: int, beta: float) -> np.ndarray:
    """symmetric Kaiser window (scipy.signal.get_window(("kaiser", beta), n, fftbins=False)): I0(beta sqrt(1 - r^2)) / I0(beta)"""
    r = (np.arange(n) - (n - 1) / 2.0) / ((n - 1) / 2.0)
    return np.i0(beta * np.sqrt(np.clip(1.0 - r * r, 0.0, None))) / np.i0(beta)


def resample_poly(x: np.ndarray, up: int, down: int) -> np.ndarray:
    """scipy.signal.resample_poly(x, up, down) (scipy 1.15 ``_signaltools.resample_poly``, default window ("kaiser", 5.0), padtype
    "constant") restated with numpy: the filter is firwin(2 half_len + 1, 1 / max_rate) -- cutoff sinc(cutoff m) under the window,
    normalised to unit DC gain -- cast to the data's type and times ``up``; upfirdn = zero-stuff by ``up``, convolve with the filter
    padded in front by n_pre_pad zeros, keep every ``down``-th sample, entries [n_pre_remove, n_pre_remove + n_out).  Last axis."""
    g = math.gcd(up, down)
    up, down = up // g, down // g
    x = np.asarray(x)
    if up == down == 1:
        return x.copy()
    n_in = x.shape[-1]
    n_out = -(-n_in * up // down)
    max_rate = max(up, down)
    half_len = 10 * max_rate
    m = np.arange(-half_len, half_len + 1, dtype=np.float64)
    cutoff = 1.0 / max_rate
    h = cutoff * np.sinc(cutoff * m) * _kaiser(2 * half_len + 1, 5.0)
    h = (h / h.sum()).astype(x.dtype) * x.dtype.type(up)
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    hp = np.concatenate([np.zeros(n_pre_pad, dtype=h.dtype), h])
    out = np.empty(x.shape[:-1] + (n_out,), dtype=x.dtype)
    for idx in np.ndindex(x.shape[:-1]):
        xu = np.zeros(n_in * up, dtype=x.dtype)
        xu[::up] = x[idx]
        full = np.convolve(xu, hp)[::down]
        seg = full[n_pre_remove: n_pre_remove + n_out]
        out[idx] = np.pad(seg, (0, n_out - len(seg)))
    return out


def resample_fft(x: np.ndarray, num: int) -> np.ndarray:
    """scipy.signal.resample(x, num) (Fourier method, real input, no window; scipy 1.15 ``_signaltools.resample``) with numpy: rfft, the
    N // 2 + 1 lowest bins (N = min(num, Nx)) into an otherwise zero spectrum, the Nyquist bin of an even N doubled (downsampling) or
    halved (upsampling), irfft to ``num`` samples, times num / Nx.  Last axis."""
    x = np.asarray(x)
    nx = x.shape[-1]
    X = np.fft.rfft(x.astype(np.float64), axis=-1)
    Y = np.zeros(x.shape[:-1] + (num // 2 + 1,), dtype=complex)
    N = min(num, nx)
    Y[..., : N // 2 + 1] = X[..., : N // 2 + 1]
    if N % 2 == 0:
        if num < nx:
            Y[..., N // 2] *= 2.0
        elif nx < num:
            Y[..., N // 2] *= 0.5
    return (np.fft.irfft(Y, num, axis=-1) * (float(num) / float(nx))).astype(x.dtype)


def resample(x: np.ndarray, sr_in: int, sr_out: int, res_type: str = "polyphase") -> np.ndarray:
    """librosa.resample(y, orig_sr, target_sr, res_type) for the kinds the VR band chain reaches (librosa/core/audio.py ``resample``;
    librosa is not in /root/reference -- its dispatch restated): equal rates return the input; "polyphase" is scipy.signal.resample_poly
    (y, target // gcd, orig // gcd), "scipy" / "fft" scipy.signal.resample(y, ceil(n target / orig)); output in the input's dtype."""
    if sr_in == sr_out:
        return x
    if res_type == "polyphase":
        return resample_poly(x, int(sr_out), int(sr_in))
    if res_type in ("scipy", "fft"):
        return resample_fft(x, -(-x.shape[-1] * int(sr_out) // int(sr_in)))
    raise ValueError(f"resampler kind {res_type!r} is not restated (samplerate's / resampy's filter tables are not in the image)")


def wave_to_spectrogram(wave: np.ndarray, hop: int, n_fft: int, convert_channels=None) -> np.ndarray:
    """spec_utils.py:30-56 (no mid-side / reverse: both False in the 4-band parameter sets): [2, n] -> complex [2, bins, frames];
    ``convert_channels="stereo_n"``: upstream's L' = (L + R / 4) / 0.9375, R' = (R + L / 4) / 0.9375 (see MODEL_PARAMS)"""
    spec = np.stack([stft(wave[0], n_fft, hop), stft(wave[1], n_fft, hop)])
    if convert_channels == "stereo_n":
        spec = np.stack([(spec[0] + spec[1] * 0.25) / 0.9375, (spec[1] + spec[0] * 0.25) / 0.9375])
    elif convert_channels is not None:
        raise ValueError(f"convert_channels {convert_channels!r}")
    return spec


def fft_lp_filter(spec, bin_start, bin_stop):
    g = 1.0
    for b in range(bin_start, bin_stop):
        g -= 1 / (bin_stop - bin_start)
        spec[:, b, :] = g * spec[:, b, :]
    spec[:, bin_stop:, :] *= 0
    return spec


def fft_hp_filter(spec, bin_start, bin_stop):
    g = 1.0
    for b in range(bin_start, bin_stop, -1):
        g -= 1 / (bin_start - bin_stop)
        spec[:, b, :] = g * spec[:, b, :]
    spec[:, 0: bin_stop + 1, :] *= 0
    return spec


def combine_spectrograms(specs: Dict[int, np.ndarray], mp: dict) -> np.ndarray:
    """spec_utils.py:95-125"""
    l = min(specs[i].shape[2] for i in specs)
    spec_c = np.zeros((2, mp["bins"] + 1, l), dtype=np.complex64)
    offset = 0
    bands_n = len(mp["band"])
    for d in range(1, bands_n + 1):
        bp = mp["band"][d]
        h = bp["crop_stop"] - bp["crop_start"]
        spec_c[:, offset: offset + h, :l] = specs[d][:, bp["crop_start"]: bp["crop_stop"], :l]
        offset += h
    if mp["pre_filter_start"] > 0:
        gp = 1
        for b in range(mp["pre_filter_start"] + 1, mp["pre_filter_stop"]):
            g = math.pow(10, -(b - mp["pre_filter_start"]) * (3.5 - gp) / 20.0)
            gp = g
            spec_c[:, b, :] *= g
    return spec_c


def mirroring(spec_m: np.ndarray, input_high_end: np.ndarray, mp: dict) -> np.ndarray:
    """spec_utils.py:453-470 ("mirroring")"""
    h = input_high_end.shape[1]
    lo = mp["pre_filter_start"] - 10 - h
    mirror = np.flip(np.abs(spec_m[:, lo: lo + h, :]), 1)
    mirror = mirror * np.exp(1.0j * np.angle(input_high_end))
    return np.where(np.abs(input_high_end) <= np.abs(mirror), input_high_end, mirror)


def cmb_spectrogram_to_wave(spec_m: np.ndarray, mp: dict, extra_bins_h=None, extra_bins=None) -> np.ndarray:
    """spec_utils.py:353-429 -> [n, 2].  (The reference allocates each band's spectrogram with np.ndarray, i.e. uninitialised; every bin
    it does not fill is zeroed by the filters except the Nyquist bin of the top band -- taken as zero here.)"""
    bands_n = len(mp["band"])
    offset = 0
    wave = None
    for d in range(1, bands_n + 1):
        bp = mp["band"][d]
        spec_s = np.zeros((2, bp["n_fft"] // 2 + 1, spec_m.shape[2]), dtype=complex)
        h = bp["crop_stop"] - bp["crop_start"]
        spec_s[:, bp["crop_start"]: bp["crop_stop"], :] = spec_m[:, offset: offset + h, :]
        offset += h
        def to_wave(s, bp=bp):
            w = np.stack([istft(s[0], bp["hl"]), istft(s[1], bp["hl"])])
            if bp.get("convert_channels") == "stereo_n":                                  # upstream spectrogram_to_wave: undone on the wave
                w = np.stack([w[0] - w[1] * 0.25, w[1] - w[0] * 0.25])
            return w
        if d == bands_n:
            if extra_bins_h:
                max_bin = bp["n_fft"] // 2
                spec_s[:, max_bin - extra_bins_h: max_bin, :] = extra_bins[:, :extra_bins_h, :]
            if bp.get("hpf_start", 0) > 0:
                spec_s = fft_hp_filter(spec_s, bp["hpf_start"], bp["hpf_stop"] - 1)
            wave = to_wave(spec_s) if bands_n == 1 else np.add(wave, to_wave(spec_s))
        else:
            sr = mp["band"][d + 1]["sr"]
            if d == 1:
                spec_s = fft_lp_filter(spec_s, bp["lpf_start"], bp["lpf_stop"])
                wave = resample(to_wave(spec_s), bp["sr"], sr, "sinc_fastest")                  # :404-414 (equal rates in the 4-band sets)
            else:
                spec_s = fft_hp_filter(spec_s, bp["hpf_start"], bp["hpf_stop"] - 1)
                spec_s = fft_lp_filter(spec_s, bp["lpf_start"], bp["lpf_stop"])
                wave = resample(np.add(wave, to_wave(spec_s)), bp["sr"], sr, "scipy")            # :427
    return wave.T


def front_end(wave: np.ndarray, mp: dict) -> Tuple[np.ndarray, np.ndarray, int]:
    """vr.py:55-96: [2, n] at mp["sr"] -> (X_spec_m, input_high_end, input_high_end_h)"""
    bands_n = len(mp["band"])
    X_wave, X_spec_s = {}, {}
    for d in range(bands_n, 0, -1):
        bp = mp["band"][d]
        X_wave[d] = wave if d == bands_n else resample(X_wave[d + 1], mp["band"][d + 1]["sr"], bp["sr"], bp["res_type"])   # vr.py:74-79
        X_spec_s[d] = wave_to_spectrogram(X_wave[d], bp["hl"], bp["n_fft"], bp.get("convert_channels"))
        if d == bands_n:
            hh = (bp["n_fft"] // 2 - bp["crop_stop"]) + (mp["pre_filter_stop"] - mp["pre_filter_start"])
            high_end = X_spec_s[d][:, bp["n_fft"] // 2 - hh: bp["n_fft"] // 2, :]
    return combine_spectrograms(X_spec_s, mp), high_end, hh


def back_end(X_spec_m: np.ndarray, pred: np.ndarray, X_phase: np.ndarray, high_end: np.ndarray, hh: int, mp: dict):
    """vr.py:103-113, 161-168: pred (magnitudes) -> (instrument wave [n, 2], vocal wave [n, 2])"""
    y_spec_m = pred * X_phase
    v_spec_m = X_spec_m - y_spec_m
    out = []
    for spec in (y_spec_m, v_spec_m):
        he = mirroring(spec, high_end, mp)
        out.append(cmb_spectrogram_to_wave(spec, mp, hh, he))
    return out[0], out[1]
