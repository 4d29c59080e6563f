"""Synthetic inputs and random-init MDX-Net weights (no dataset / checkpoint is reachable:
models are downloaded at run time by the reference, stem_separator.py:123-124).

Not on the separation path: this only *creates data* -- a deterministic test signal
 and a ``state_dict`` with torch's layer names.  BatchNorm running statistics
are calibrated once on a small random proxy spectrogram (torch CPU, train-mode statistics) so
that activations stay O(1) through the 40-layer multiplicative-skip U-Net, as they do in a
trained model; the inference path itself never touches torch ops.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from .tdfnet import BN_EPS, TDFNetConfig


def synth_mix(n_samples: int, channels: int = 2, sr: int = 44100, seed: int = 20251017) -> np.ndarray:
    """0.08*N(0,1) + sines 110/440/3520 Hz @0.1 with per-channel phase, 0.25 Hz tremolo,
    clipped to [-1,1], float32, [C,N]."""
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


def _kaiming_uniform(gen: torch.Generator, shape, fan_in: int) -> torch.Tensor:
    bound = math.sqrt(6.0 / fan_in)                      # gain sqrt(2) (ReLU)
    return (torch.rand(shape, generator=gen) * 2 - 1) * bound


def proxy_spectrogram(cfg: TDFNetConfig, frames: int, seed: int) -> torch.Tensor:
    """[1,4,dim_f,frames] spectrogram of the synthetic test signal (torch CPU stft, data only):
    the statistics the random network is calibrated on must look like its real input --
    sine peaks ~40x above the noise floor -- because the multiplicative skips make the
    network's gain depend strongly on the input's magnitude distribution."""
    n = cfg.hop * (frames - 1)
    x = torch.from_numpy(synth_mix(n + 4 * cfg.hop, seed=seed + 1234)[:, 2 * cfg.hop: 2 * cfg.hop + n].copy())
    z = torch.stft(x, cfg.n_fft, cfg.hop, window=torch.hann_window(cfg.n_fft, periodic=True), center=True,
                   return_complex=True)[:, : cfg.dim_f]                       # [2, dim_f, frames]
    z = torch.view_as_real(z).permute(0, 3, 1, 2).reshape(1, 4, cfg.dim_f, frames)
    return z.contiguous().float()


def synthetic_state_dict(cfg: TDFNetConfig, seed: int = 0, input_rms: float = 4.0,
                         calib_frames: int = 32, calib: str = "signal") -> Dict[str, torch.Tensor]:
    """Random-init TFC-TDF U-Net weights with calibrated BatchNorm statistics (CPU tensors).
    calib="signal": statistics of the synthetic test signal's spectrogram (default);
    calib="noise": white Gaussian spectrogram of RMS ``input_rms`` (small unit-test nets)."""
    gen = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    n, g = cfg.n, cfg.g

    def add_bn(p: str, c: int, gamma: float = 1.0, beta: float = 0.0):
        sd[p + ".weight"] = torch.full((c,), gamma)
        sd[p + ".bias"] = torch.full((c,), beta)
        sd[p + ".running_mean"] = torch.zeros(c)
        sd[p + ".running_var"] = torch.ones(c)

    def add_conv(p: str, cout: int, cin: int, kh: int, kw: int, fan_in: int, transposed: bool = False):
        shape = (cin, cout, kh, kw) if transposed else (cout, cin, kh, kw)
        sd[p + ".weight"] = _kaiming_uniform(gen, shape, fan_in)
        sd[p + ".bias"] = (torch.rand(cout, generator=gen) * 2 - 1) * 0.05

    def add_block(p: str, c: int, f: int):
        for j in range(cfg.l):
            add_conv(f"{p}.tfc.H.{j}.0", c, c, 3, 3, 9 * c)
            add_bn(f"{p}.tfc.H.{j}.1", c)
        dims = [(f, f)] if cfg.bn == 0 else [(f // cfg.bn, f), (f, f // cfg.bn)]
        for j, (fo, fi) in enumerate(dims):
            sd[f"{p}.tdf.{3 * j}.weight"] = _kaiming_uniform(gen, (fo, fi), fi)
            if cfg.bias:
                sd[f"{p}.tdf.{3 * j}.bias"] = (torch.rand(fo, generator=gen) * 2 - 1) * 0.05
            add_bn(f"{p}.tdf.{3 * j + 1}", c)

    add_conv("first_conv.0", g, cfg.dim_c, 1, 1, cfg.dim_c)
    add_bn("first_conv.1", g)
    c, f = g, cfg.dim_f
    for i in range(n):
        add_block(f"encoding_blocks.{i}", c, f)
        add_conv(f"ds.{i}.0", c + g, c, 2, 2, 4 * c)
        add_bn(f"ds.{i}.1", c + g)
        c += g
        f //= 2
    add_block("bottleneck_block", c, f)
    for i in range(n):
        add_conv(f"us.{i}.0", c - g, c, 2, 2, c, transposed=True)
        # gate-like up path: relu(0.25*z + 1) stays near 1, so the multiplicative skip modulates
        # instead of squaring magnitudes (a trained model's skips are tame in the same way)
        add_bn(f"us.{i}.1", c - g, gamma=0.25, beta=1.0)
        c -= g
        f *= 2
        add_block(f"decoding_blocks.{i}", c, f)
    sd["final_conv.0.weight"] = _kaiming_uniform(gen, (cfg.dim_c, g, 1, 1), g) / math.sqrt(2.0)
    sd["final_conv.0.bias"] = torch.zeros(cfg.dim_c)
    _calibrate(sd, cfg, gen, input_rms, calib_frames, calib, seed)
    return sd


@torch.no_grad()
def _calibrate(sd: Dict[str, torch.Tensor], cfg: TDFNetConfig, gen: torch.Generator, input_rms: float, frames: int,
               calib: str = "signal", seed: int = 0) -> None:
    """Set every BatchNorm's running_mean/var to its batch statistics on a random proxy input
    and scale final_conv so that the output RMS equals the input RMS."""
    frames = max(frames, 2 ** cfg.n)
    frames -= frames % (2 ** cfg.n)
    if calib == "signal" and cfg.hop * (frames - 1) > cfg.n_fft // 2:
        x = proxy_spectrogram(cfg, frames, seed)
    else:
        x = torch.randn((1, cfg.dim_c, cfg.dim_f, frames), generator=gen) * input_rms

    def bn_relu(y: torch.Tensor, p: str) -> torch.Tensor:
        dims = [0, 2, 3]
        mean = y.mean(dim=dims)
        var = y.var(dim=dims, unbiased=False).clamp_min(1e-8)
        sd[p + ".running_mean"] = mean.clone()
        sd[p + ".running_var"] = var.clone()
        y = (y - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
        return F.relu(y * sd[p + ".weight"][None, :, None, None] + sd[p + ".bias"][None, :, None, None])

    def block(y: torch.Tensor, p: str) -> torch.Tensor:
        for j in range(cfg.l):
            y = bn_relu(F.conv2d(y, sd[f"{p}.tfc.H.{j}.0.weight"], sd[f"{p}.tfc.H.{j}.0.bias"], padding=1), f"{p}.tfc.H.{j}.1")
        t = y
        for j in range(1 if cfg.bn == 0 else 2):
            t = bn_relu(F.linear(t, sd[f"{p}.tdf.{3 * j}.weight"], sd.get(f"{p}.tdf.{3 * j}.bias")), f"{p}.tdf.{3 * j + 1}")
        return y + t

    y = bn_relu(F.conv2d(x, sd["first_conv.0.weight"], sd["first_conv.0.bias"]), "first_conv.1").transpose(-1, -2)
    skips = []
    for i in range(cfg.n):
        y = block(y, f"encoding_blocks.{i}")
        skips.append(y)
        y = bn_relu(F.conv2d(y, sd[f"ds.{i}.0.weight"], sd[f"ds.{i}.0.bias"], stride=2), f"ds.{i}.1")
    y = block(y, "bottleneck_block")
    for i in range(cfg.n):
        y = bn_relu(F.conv_transpose2d(y, sd[f"us.{i}.0.weight"], sd[f"us.{i}.0.bias"], stride=2), f"us.{i}.1")
        y = y * skips[-i - 1]
        y = block(y, f"decoding_blocks.{i}")
    out = F.conv2d(y.transpose(-1, -2), sd["final_conv.0.weight"], sd["final_conv.0.bias"])
    gain = float(x.pow(2).mean().sqrt() / out.pow(2).mean().sqrt().clamp_min(1e-12))
    sd["final_conv.0.weight"] = sd["final_conv.0.weight"] * gain
