"""CPU restatement of the ensemble post-ops of the Separate engine (numpy).

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Not imported by the product.

Follows (file:line relative to /root/reference/modules/separator/stem_separator.py):
  * blend_tracks        :241-262  (_blend_tracks)
  * residual_subtract   :173-239  (_residual_subtract)
  * debleed             :415-456  (post-blend de-bleed + silent-instrumental fallback)
Pinned by tests/golden/ensemble_*.npz (oracle/make_golden.py imports the reference).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np


def blend_tracks(tracks: Sequence[np.ndarray], weights: Sequence[float]) -> np.ndarray:
    """:241-262.  Weighted mean (zero-padded to the longest), then peak-normalise to 1.0."""
    max_len = max(t.shape[-1] for t in tracks)
    acc = np.zeros((tracks[0].shape[0], max_len), dtype=np.float32)
    total_w = max(sum(weights), 1e-6)
    for i, t in enumerate(tracks):
        w = weights[i] if i < len(weights) else 1.0
        acc[:, : t.shape[-1]] += t * float(w)
    acc = acc / total_w
    peak = np.max(np.abs(acc))
    if peak > 0:
        acc /= peak
    return acc


def best_lag(ref: np.ndarray, sig: np.ndarray, max_shift: int, probe: int = 44100) -> int:
    """:216-224.  argmax over lags in [-max_shift, +max_shift] of sum_n ref[n+lag]*sig[n]
    on the first ``probe`` samples (== np.correlate(ref, sig, 'full') centre window)."""
    n = min(len(ref), len(sig), probe)
    r = ref[:n].astype(np.float32)
    s = sig[:n].astype(np.float32)
    vals = np.empty(2 * max_shift + 1, dtype=np.float64)
    for j, lag in enumerate(range(-max_shift, max_shift + 1)):
        if lag >= 0:
            vals[j] = np.dot(r[lag:], s[: n - lag]) if lag < n else 0.0
        else:
            vals[j] = np.dot(r[: n + lag], s[-lag:]) if -lag < n else 0.0
    return int(np.argmax(vals)) - max_shift


def shift_signal(x: np.ndarray, lag: int) -> np.ndarray:
    """:196-208."""
    if lag == 0:
        return x
    if lag > 0:
        return np.concatenate([np.zeros(lag, dtype=x.dtype), x[:-lag]])
    lag = -lag
    return np.concatenate([x[lag:], np.zeros(lag, dtype=x.dtype)])


def residual_subtract(base: np.ndarray, component: np.ndarray, sr: int,
                      max_shift_ms: float = 12.0,
                      return_params: bool = False):
    """:173-239.  Per channel: align (xcorr argmax, +-max_shift), LS gain clipped to
    [0,1.25], subtract.  Shapes [C,N]."""
    if base.ndim == 1:
        base = np.stack([base, base], axis=0)
    if component.ndim == 1:
        component = np.stack([component, component], axis=0)
    max_shift = max(int((max_shift_ms / 1000.0) * float(sr)), 0)
    n = min(base.shape[-1], component.shape[-1])
    residual = np.copy(base)
    params: List[Tuple[int, float]] = []
    for ch in range(base.shape[0]):
        ref = base[ch, :n]
        sig = component[ch, :n]
        lag = best_lag(ref, sig, max_shift) if (max_shift > 0 and n > 0) else 0
        sig_al = shift_signal(sig, lag)
        denom = float(np.dot(sig_al, sig_al)) + 1e-8
        alpha = float(np.clip(float(np.dot(ref, sig_al)) / denom, 0.0, 1.25))
        residual[ch, :n] = ref - alpha * sig_al
        params.append((lag, alpha))
    if not np.isfinite(residual).all():
        residual = np.nan_to_num(residual, nan=0.0, posinf=0.0, neginf=0.0)
    return (residual, params) if return_params else residual


def cosine_abs(a: np.ndarray, b: np.ndarray) -> float:
    """:430-434."""
    a = a.reshape(-1)
    b = b.reshape(-1)
    denom = (np.linalg.norm(a) * np.linalg.norm(b)) + 1e-8
    return float(abs(np.dot(a, b)) / denom)


def debleed(mix: np.ndarray, vocals: np.ndarray, instrumental: np.ndarray, sr: int,
            residual_blend: float = 0.4) -> Tuple[np.ndarray, bool]:
    """:415-456.  Returns (instrumental', accepted)."""
    accepted = False
    resid = residual_subtract(mix, vocals, sr)
    m = min(resid.shape[-1], instrumental.shape[-1])
    resid = resid[:, :m]
    inst = instrumental[:, :m]
    sim_inst = cosine_abs(inst, vocals[:, :m])
    sim_resid = cosine_abs(resid, vocals[:, :m])
    out = instrumental
    if sim_resid + 1e-6 < sim_inst - 0.01:
        b = min(max(float(residual_blend), 0.0), 1.0)
        ref = (1.0 - b) * inst + b * resid
        peak = float(np.max(np.abs(ref)))
        if peak > 0.99:
            ref = ref * (0.99 / peak)
        out = ref
        accepted = True
    i_peak = float(np.max(np.abs(out))) if out.size else 0.0
    if i_peak < 1e-6:                                   # :448-456
        resid = residual_subtract(mix, vocals, sr)
        peak = float(np.max(np.abs(resid)))
        if peak > 1.0:
            resid = resid / peak
        out = resid
    return out, accepted
