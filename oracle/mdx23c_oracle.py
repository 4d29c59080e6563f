"""Plain torch-CPU fp32 restatement of the MDX23C network (TFC-TDF v3, "TFC_TDF_net") and of its chunked runner.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Not imported by the product.

PARITY UNPINNED.  The reference runs ``MDX23C-8KFFT-InstVoc_HQ.ckpt`` as ensemble slot 4 (/root/reference/modules/separator/
stem_separator.py:383) and ``MDX23C-DrumSep-aufr33-jarredou.ckpt`` as its drum-kit splitter (:541, six outputs matched by ``(kick)`` ...
``(crash)`` :563-574) through ``audio-separator[gpu]>=0.32.0`` (setup.sh:96), whose vendored ``tfc_tdf_v3.py`` is not in /root/reference.
Restated from the published design with that code's parameter names (``encoder_blocks.0.tfc_tdf.blocks.1.tdf.2.weight`` ...):

  * STFT n_fft 8192 / hop 1024 of both channels, complex-as-channels, bins < dim_f, split into ``num_subbands`` frequency sub-bands
    stacked on the channel axis (cac2cws);
  * first 1x1 conv, ``num_scales`` encoder stages (TFC_TDF block, then norm-act-strided conv down-scale), bottleneck TFC_TDF,
    decoder stages (norm-act-transposed conv up-scale, concatenation with the encoder output, TFC_TDF);
  * TFC_TDF block = ``l`` x [shortcut 1x1 conv; norm-act-conv3x3; + TDF (norm-act-Linear(f, f/bn), norm-act-Linear(f/bn, f) over the
    frequency axis); norm-act-conv3x3; + shortcut]; norm = InstanceNorm2d(affine), act = GELU, no biases;
  * output: x * first_conv_out, 1x1-conv head on [mix, x], sub-bands merged back (cws2cac), iSTFT per target instrument.

Runner: the same chunked inference as the Roformer models of the training project (oracle/roformer_oracle.demix_track).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class MDX23CConfig:
    instruments: Tuple[str, ...] = ("vocals", "other")        # MDX23C-8KFFT-InstVoc_HQ
    n_fft: int = 8192
    hop: int = 1024
    dim_f: int = 4096
    num_subbands: int = 4
    num_scales: int = 5
    scale: Tuple[int, int] = (2, 2)
    num_blocks_per_scale: int = 2
    num_channels: int = 128
    growth: int = 128
    bottleneck_factor: int = 4
    chunk_size: int = 261120
    num_overlap: int = 4
    sample_rate: int = 44100

    @property
    def num_stems(self) -> int:
        return len(self.instruments)

    @property
    def dim_c(self) -> int:
        return self.num_subbands * 2 * 2


def _norm_act(w, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.gelu(F.instance_norm(x, weight=w[p + ".weight"], bias=w[p + ".bias"], eps=1e-5))


def _h(x: torch.Tensor, half: bool) -> torch.Tensor:
    """storage rounding of the build's half-precision mode: what a half-precision convolution reads is IEEE half (float32 arithmetic on
    the rounded values: the MFMA accumulates in float32)"""
    return x.half().float() if half else x


def _tfc_tdf(w, p: str, x: torch.Tensor, l: int, half: bool = False) -> torch.Tensor:
    for i in range(l):
        q = f"{p}.blocks.{i}"
        s = F.conv2d(_h(x, half), _h(w[q + ".shortcut.weight"], half))
        x = F.conv2d(_h(_norm_act(w, q + ".tfc1.0", x), half), _h(w[q + ".tfc1.2.weight"], half), padding=1)
        hl = half and x.shape[-1] % 32 == 0                  # the build runs the TDF linears in half where the frequency count allows (mdx23c._block)
        t = F.linear(_h(_norm_act(w, q + ".tdf.0", x), hl), _h(w[q + ".tdf.2.weight"], hl))
        t = F.linear(_h(_norm_act(w, q + ".tdf.3", t), hl), _h(w[q + ".tdf.5.weight"], hl))
        x = x + t
        x = F.conv2d(_h(_norm_act(w, q + ".tfc2.0", x), half), _h(w[q + ".tfc2.2.weight"], half), padding=1)
        x = x + s
    return x


@torch.no_grad()
def forward(cfg: MDX23CConfig, w: Dict[str, torch.Tensor], audio: torch.Tensor, half: bool = False) -> torch.Tensor:
    """audio [B, 2, L] (L = hop * (T - 1)) -> [B, num_stems, 2, L].  ``half``: the storage-mode oracle of the build's half-precision mode
    (audiolab_amd/mdx23c.py ``precision="f16"``): the inputs and weights of the TFC-TDF blocks' convolutions, of the down- / up-scaling
    convolutions and of the TDF linears rounded to IEEE half, everything else float32 -- NOT torch autocast (which would also round
    those layers' outputs); the reference's own autocast run is a third thing this oracle does not claim to reproduce."""
    B, C, L = audio.shape
    win = torch.hann_window(cfg.n_fft)
    z = torch.stft(audio.reshape(B * C, L), cfg.n_fft, cfg.hop, window=win, center=True, return_complex=True)
    z = torch.view_as_real(z).permute(0, 3, 1, 2)                                   # [(b c), 2, n_bins, T]
    T = z.shape[-1]
    x = z.reshape(B, C * 2, -1, T)[..., : cfg.dim_f, :]                              # [B, 4, dim_f, T]
    k = cfg.num_subbands
    f = cfg.dim_f // k
    mix = x = x.reshape(B, C * 2 * k, f, T)                                          # cac2cws
    first = x = F.conv2d(x, w["first_conv.weight"])
    x = x.transpose(-1, -2)
    enc = []
    for i in range(cfg.num_scales):
        x = _tfc_tdf(w, f"encoder_blocks.{i}.tfc_tdf", x, cfg.num_blocks_per_scale, half)
        enc.append(x)
        x = F.conv2d(_h(_norm_act(w, f"encoder_blocks.{i}.downscale.conv.0", x), half), _h(w[f"encoder_blocks.{i}.downscale.conv.2.weight"], half),
                     stride=cfg.scale)
    x = _tfc_tdf(w, "bottleneck_block", x, cfg.num_blocks_per_scale, half)
    for i in range(cfg.num_scales):
        x = F.conv_transpose2d(_h(_norm_act(w, f"decoder_blocks.{i}.upscale.conv.0", x), half),
                               _h(w[f"decoder_blocks.{i}.upscale.conv.2.weight"], half), stride=cfg.scale)
        x = torch.cat([x, enc.pop()], 1)
        x = _tfc_tdf(w, f"decoder_blocks.{i}.tfc_tdf", x, cfg.num_blocks_per_scale, half)
    x = x.transpose(-1, -2)
    x = x * first
    x = F.conv2d(F.gelu(F.conv2d(torch.cat([mix, x], 1), w["final_conv.0.weight"])), w["final_conv.2.weight"])
    S = cfg.num_stems
    x = x.reshape(B, S * C * 2, k * f, T)                                            # cws2cac, instruments on the channel axis
    x = x.reshape(B, S, C * 2, cfg.dim_f, T)
    n_bins = cfg.n_fft // 2 + 1
    x = torch.cat([x, torch.zeros(B, S, C * 2, n_bins - cfg.dim_f, T)], dim=-2)
    zc = torch.view_as_complex(x.reshape(B * S * C, 2, n_bins, T).permute(0, 2, 3, 1).contiguous())
    y = torch.istft(zc, cfg.n_fft, cfg.hop, window=win, center=True)
    return y.reshape(B, S, C, -1)


def synthetic_state_dict(cfg: MDX23CConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def conv(p, cout, cin, kh, kw, transposed=False, gain=1.0):
        fan = cin * kh * kw
        shape = (cin, cout, kh, kw) if transposed else (cout, cin, kh, kw)
        sd[p + ".weight"] = (torch.rand(shape, generator=g) * 2 - 1) * (gain * math.sqrt(3.0 / fan))

    def lin(p, out, inp, gain=1.0):
        sd[p + ".weight"] = (torch.rand(out, inp, generator=g) * 2 - 1) * (gain * math.sqrt(3.0 / inp))

    def norm(p, c):
        sd[p + ".weight"] = 1.0 + 0.1 * (torch.rand(c, generator=g) * 2 - 1)
        sd[p + ".bias"] = 0.05 * (torch.rand(c, generator=g) * 2 - 1)

    def tfc_tdf(p, in_c, c, f):
        for i in range(cfg.num_blocks_per_scale):
            q = f"{p}.blocks.{i}"
            norm(q + ".tfc1.0", in_c)
            conv(q + ".tfc1.2", c, in_c, 3, 3, gain=1.4)
            norm(q + ".tdf.0", c)
            lin(q + ".tdf.2", f // cfg.bottleneck_factor, f, gain=1.4)
            norm(q + ".tdf.3", c)
            lin(q + ".tdf.5", f, f // cfg.bottleneck_factor, gain=1.4)
            norm(q + ".tfc2.0", c)
            conv(q + ".tfc2.2", c, c, 3, 3, gain=1.4)
            conv(q + ".shortcut", c, in_c, 1, 1)
            in_c = c

    c, gr, f = cfg.num_channels, cfg.growth, cfg.dim_f // cfg.num_subbands
    conv("first_conv", c, cfg.dim_c, 1, 1)
    for i in range(cfg.num_scales):
        tfc_tdf(f"encoder_blocks.{i}.tfc_tdf", c, c, f)
        norm(f"encoder_blocks.{i}.downscale.conv.0", c)
        conv(f"encoder_blocks.{i}.downscale.conv.2", c + gr, c, cfg.scale[0], cfg.scale[1], gain=1.4)
        f //= cfg.scale[1]
        c += gr
    tfc_tdf("bottleneck_block", c, c, f)
    for i in range(cfg.num_scales):
        norm(f"decoder_blocks.{i}.upscale.conv.0", c)
        conv(f"decoder_blocks.{i}.upscale.conv.2", c - gr, c, cfg.scale[0], cfg.scale[1], transposed=True, gain=1.4)
        f *= cfg.scale[1]
        c -= gr
        tfc_tdf(f"decoder_blocks.{i}.tfc_tdf", 2 * c, c, f)
    conv("final_conv.0", c, c + cfg.dim_c, 1, 1)
    conv("final_conv.2", cfg.num_stems * cfg.dim_c, c, 1, 1)
    return sd
