"""CPU restatement (numpy + scipy) of the reference's reverb impulse-response extraction, ``handlers/reverb.py:112-172`` with its
helpers ``to_mono`` (:52-53), ``fft_xcorr`` (:55-66), ``estimate_rt60`` (:69-91) and ``wiener_deconvolution`` (:94-106).
TEST INFRASTRUCTURE -- see ``oracle/__init__.py``: only tests / smoke / the bench's cpu_baseline may import this.

PINNED by tests/golden/reverb.npz (oracle/make_golden_reverb.py runs the reference's own functions here).

Precision.  The reference pins numpy 2.x (setup.sh:93 ``numpy==2.0.2``), whose ``np.fft`` keeps float32 inputs in SINGLE precision: with
``precision=np.float32`` every step below runs in the dtype the reference's environment runs it in (the audio arrays are float32) and
reproduces the golden vectors bit for bit on the same numpy.  ``precision=np.float64`` promotes the two mono signals first -- what numpy
1.x did implicitly, and the arithmetic the HIP path uses (csrc/reverb.hip computes in double: a 13-million-point single-precision
deconvolution has no bits to spare) -- and is the yardstick for how far the reference's own rounding sits from the exact result.

Reference behaviours kept on purpose:
  * ``fft_xcorr`` returns the CIRCULAR correlation (lag l at index l, negative lags wrapped to the end), but ``extract_reverb`` reads it
    as np.correlate's 'full' layout (``argmax - (len(dry) - 1)``, :131): a positive lag far below len(dry) - 1 gives a negative shift,
    clamped to 0 (:132) -- the pre-delay is 0 for every realistic input;
  * an odd-length wet signal yields n - 1 impulse-response samples (``np.fft.irfft`` default length, :104);
  * with numpy >= 2 the reference's json.dump raises on the np.float32 ratios (:39-41) and the call site logs the error
    (stem_separator.py:828-829); this restatement returns plain Python floats, the values the reference computed.
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def to_mono(signal: np.ndarray) -> np.ndarray:
    """:52-53"""
    return np.mean(signal, axis=1) if signal.ndim == 2 else signal


def fft_xcorr(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """:55-66"""
    n = len(a) + len(b) - 1
    n_fft = 1 << (n - 1).bit_length()
    fa = np.fft.rfft(a, n=n_fft)
    fb = np.fft.rfft(b, n=n_fft)
    return np.fft.irfft(fa * np.conjugate(fb), n=n_fft)[:n]


def envelope_db(signal: np.ndarray) -> np.ndarray:
    """the curve the decay is fitted to (:74-81): float32 arithmetic on float32 audio"""
    eps = 1e-10
    env = (np.sqrt(np.sum(signal ** 2, axis=1)) if signal.ndim == 2 else np.abs(signal)) + eps
    return 20.0 * np.log10(env)


def fit_decay(env_db: np.ndarray, sr: int, maxfev: int = 5000) -> float:
    """:82-91 -- scipy's Levenberg-Marquardt from its default start (1, 1, 1)"""
    from scipy.optimize import curve_fit
    time = np.linspace(0, len(env_db) / sr, len(env_db))

    def exp_decay(x, a, b, c):
        return a * np.exp(-b * x) + c
    popt, _ = curve_fit(exp_decay, time, env_db, maxfev=maxfev)
    decay_time = 3.0 / popt[1] if popt[1] != 0 else 0.5
    return float(max(decay_time, 0.01))


def wiener_deconvolution(signal: np.ndarray, kernel: np.ndarray, epsilon: float = 1e-6) -> np.ndarray:
    """:94-106"""
    h = np.fft.rfft(kernel, len(signal))
    y = np.fft.rfft(signal)
    return np.fft.irfft((np.conjugate(h) * y) / (np.abs(h) ** 2 + epsilon))


def ir_statistics(ir: np.ndarray, sr: int) -> Dict[str, float]:
    """:146-157"""
    early = int(0.05 * sr)
    early_energy = np.sum(np.square(ir[:early]))
    total_energy = np.sum(np.square(ir)) + 1e-10
    mag = np.abs(np.fft.rfft(ir))
    freqs = np.fft.rfftfreq(len(ir), d=1.0 / sr)
    return {"early_reflection_ratio": float(early_energy / total_energy),
            "late_reverb_ratio": float((total_energy - early_energy) / total_energy),
            "diffusion": float(np.var(np.abs(ir))),
            "spectral_centroid": float(np.sum(freqs * mag) / (np.sum(mag) + 1e-10))}


def extract_reverb(dry: np.ndarray, wet: np.ndarray, sr: int, wiener_epsilon: float = 1e-6, curve_fit_maxfev: int = 5000,
                   precision=np.float32) -> Dict:
    """:112-172 on arrays as ``read_audio`` returns them (float32 ``[N, C]`` or ``[N]``) -> the dictionary the reference saves."""
    dry, wet = np.asarray(dry, dtype=np.float32), np.asarray(wet, dtype=np.float32)
    dry_mono, wet_mono = to_mono(dry).astype(precision), to_mono(wet).astype(precision)
    corr = fft_xcorr(wet_mono, dry_mono)
    best_shift = max(int(np.argmax(corr)) - (len(dry_mono) - 1), 0)
    decay_time = fit_decay(envelope_db(wet), sr, curve_fit_maxfev)
    ir = wiener_deconvolution(wet_mono, dry_mono, wiener_epsilon)[: int(sr * 2)]
    out = {"sample_rate": sr, "pre_delay": float(best_shift / sr), "decay_time": decay_time}
    out.update(ir_statistics(ir, sr))
    out["impulse_response"] = ir.tolist()
    return out
