"""Deterministic stand-in "networks" and synthetic signals shared by the golden generator
and the tests.  TEST INFRASTRUCTURE -- see ``oracle/__init__.py``."""
from __future__ import annotations

import numpy as np


def toy_net(spek: np.ndarray) -> np.ndarray:
    """Linear spectrogram->spectrogram map mixing frames and L/R channels.

    Linear on purpose: then ``0.5*f(x) - 0.5*f(-x) == f(x)`` (the denoise path of
    mdxnet.py:168-173) is a checkable identity, and channel/frame mix-ups show."""
    s = np.asarray(spek)
    return (0.6 * s + 0.3 * np.roll(s, 1, axis=3) + 0.1 * s[:, [2, 3, 0, 1]]).astype(s.dtype)


def toy_net_affine(spek: np.ndarray) -> np.ndarray:
    """Non-odd map (has an even part) so the denoise average changes the result."""
    s = np.asarray(spek)
    return (0.7 * s + 0.05 * np.abs(np.roll(s, 2, axis=2))).astype(s.dtype)


def synth_mix(n_samples: int, channels: int = 2, sr: int = 44100, seed: int = 20251017) -> np.ndarray:
    """SURVEY.md 8(d) synthetic input: 0.08*N(0,1) + sines 110/440/3520 Hz @0.1 with
    per-channel phase, 0.25 Hz tremolo, clipped to [-1,1], float32.  [C,N]."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples, dtype=np.float64) / sr
    out = np.empty((channels, n_samples), dtype=np.float32)
    for c in range(channels):
        x = 0.08 * rng.standard_normal(n_samples)
        for k, f in enumerate((110.0, 440.0, 3520.0)):
            x += 0.1 * np.sin(2 * np.pi * f * t + 0.7 * c + 0.3 * k)
        x *= 0.75 + 0.25 * np.sin(2 * np.pi * 0.25 * t + 0.5 * c)
        out[c] = np.clip(x, -1.0, 1.0).astype(np.float32)
    return out


def resid_case(lag: int, gain: float, n: int = 48000):
    """(base, component) pair for the residual-subtract fixtures: base = gain*shift(comp,lag) + noise."""
    comp = synth_mix(n, seed=77 + abs(lag)) * np.float32(0.5)
    other = np.random.default_rng(900 + abs(lag)).standard_normal((2, n)).astype(np.float32) * np.float32(0.02)
    shifted = np.stack([np.roll(c, lag) for c in comp])
    base = (np.float32(gain) * shifted + other).astype(np.float32)
    return base, comp.astype(np.float32)
