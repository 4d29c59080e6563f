"""Deterministic (dry, wet) pairs for the reverb impulse-response fixtures (tests/golden/reverb.npz), shared by the golden generator
and the tests.  TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Arrays are what the reference's ``read_audio`` returns
(handlers/reverb.py:22-36): float32, ``[N, C]`` for multi-channel files, ``[N]`` for mono ones."""
from __future__ import annotations

import numpy as np

# name -> (sample rate, dry length, wet length, channels (0 = mono 1-D arrays), pre-delay samples, decay seconds, seed)
CASES = {
    "odd_stereo": (8000, 6001, 6001, 2, 0, 0.12, 1),            # odd length: irfft returns n - 1 samples
    "even_mono": (16000, 20000, 20000, 0, 160, 0.25, 2),        # 1-D arrays (mono files), 10 ms pre-delay
    "delay_stereo": (44100, 48000, 48000, 2, 1632, 0.30, 3),    # 37 ms pre-delay; the track is shorter than the 2 s impulse-response cap
    "ragged": (22050, 28000, 30000, 2, 50, 0.20, 4),            # wet longer than dry: the dry signal is zero-padded to the wet length
    "long_stereo": (44100, 120001, 120001, 2, 441, 0.2, 5),     # impulse response cut at 2 s = 88 200 samples
}


def make_case(name: str):
    """-> (dry, wet, sr): dry = decaying noise bursts + tones, wet = 0.6 * (dry * exponential-noise room response, delayed) + 0.02 * dry"""
    sr, n_dry, n_wet, ch, delay, decay, seed = CASES[name]
    rng = np.random.default_rng(seed)
    c = max(ch, 1)
    t = np.arange(n_dry) / sr
    dry = np.zeros((n_dry, c))
    for k in range(c):
        bursts = rng.standard_normal(n_dry) * (np.exp(-((t * 3.0 + 0.37 * k) % 1.0) * 6.0))
        dry[:, k] = 0.25 * bursts + 0.1 * np.sin(2 * np.pi * (220.0 + 110.0 * k) * t) * np.exp(-t * 1.5)
    m = int(decay * sr * 2.5)
    room = rng.standard_normal(m) * np.exp(-np.arange(m) / (decay * sr / 6.9))
    room[0] = 1.0
    room /= np.sqrt(np.sum(room ** 2))
    wet = np.zeros((n_wet, c))
    for k in range(c):
        full = np.convolve(dry[:, k], room)
        seg = np.concatenate([np.zeros(delay), full])[:n_wet]
        wet[:len(seg), k] = 0.6 * seg
        wet[:min(n_wet, n_dry), k] += 0.02 * dry[:min(n_wet, n_dry), k]
    dry32, wet32 = dry.astype(np.float32), wet.astype(np.float32)
    if ch == 0:
        return dry32[:, 0].copy(), wet32[:, 0].copy(), sr
    return dry32, wet32, sr
