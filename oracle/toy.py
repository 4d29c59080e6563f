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


# Deterministic stand-in "models" for the orchestration fixtures: model file name -> [(output label, gain, shift)] in the
# engine's output order.  y = gain * roll(x, shift) per output: enough structure for the lag search / gain fit of the
# residual subtraction (stem_separator.py:173-239) and the label matching of the transform chain (:808-833) to matter.
TOY_MODELS = {
    "dereverb_mel_band_roformer_anvuew_sdr_19.1729.ckpt": [("No Reverb", 0.8, 0), ("Reverb", 0.2, 7)],
    "dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt": [("No dry", 0.15, 11), ("dry", 0.85, 0)],     # wanted label second
    "UVR-MDX-NET_Crowd_HQ_1.onnx": [("No Crowd", 0.9, 0), ("Crowd", 0.1, 3)],
    "UVR-DeNoise.pth": [("Noise", 0.05, 1), ("No Noise", 0.95, 0)],
    "UVR-BVE-4B_SN-44100-1.pth": [("Vocals", 0.25, 5), ("Instrumental", 0.75, 0)],
    "MDX23C-DrumSep-aufr33-jarredou.ckpt": [("Kick", 0.30, 0), ("Snare", 0.22, 2), ("Toms", 0.15, -3), ("HH", 0.10, 5),
                                            ("Ride", 0.08, -1), ("Crash", 0.05, 4)],
    "17_HP-Wind_Inst-UVR.pth": [("No Woodwinds", 0.7, 0), ("Woodwinds", 0.3, 6)],
    "htdemucs_6s.yaml": [("Vocals", 0.30, 0), ("Drums", 0.20, 1), ("Bass", 0.15, -2), ("Guitar", 0.12, 3), ("Piano", 0.10, -4),
                         ("Other", 0.13, 2)],
}


def toy_model_outputs(model_file: str, x: np.ndarray):
    """[(label, y)] for a [C,N] input."""
    return [(label, (np.float32(g) * np.roll(x, s, axis=-1)).astype(np.float32)) for label, g, s in TOY_MODELS[model_file]]
