"""Plain torch-CPU fp32 restatement of the Roformer separation networks (BS-RoFormer / Mel-Band RoFormer) and of their chunked runner.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Not imported by the product.

PARITY UNPINNED.  These are the models the reference's DEFAULT ensemble and its de-reverb / de-echo transforms run
(/root/reference/modules/separator/stem_separator.py:379-382: ``vocals_mel_band_roformer.ckpt``,
``model_bs_roformer_ep_368_sdr_12.9628.ckpt``, ``melband_roformer_big_beta4.ckpt``; :796-797
``dereverb_mel_band_roformer_anvuew_sdr_19.1729.ckpt``, ``dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt``), loaded through
``audio-separator[gpu]>=0.32.0`` (setup.sh:96), whose vendored ``bs_roformer.py`` / ``mel_band_roformer.py`` (lucidrains' BS-RoFormer as
trained by the Music-Source-Separation-Training project) are not in /root/reference or in this image.  Restated from the published design,
with that code's parameter names (``band_split.to_features.3.1.weight``, ``layers.0.1.layers.0.0.to_qkv.weight``,
``mask_estimators.0.to_freqs.5.0.net.2.bias`` ...) so that a real ``state_dict`` would load:

  * complex STFT (n_fft 2048, hop 441, Hann) of both channels, frequency and channel merged ``(f s)``, real / imaginary as features;
  * band split: BS = contiguous bands of ``freqs_per_bands`` bins; Mel = the bins under each filter of a slaney mel filter bank
    (overlapping), gathered; per band RMSNorm + Linear to ``dim``;
  * ``depth`` x (transformer over time per band, transformer over bands per frame): pre-RMSNorm attention with rotary embeddings and
    per-head sigmoid gates, GELU feed-forward x4, residuals; final RMSNorm;
  * mask estimator per stem: per band MLP (Linear, Tanh, Linear) + GLU -> complex mask; Mel: masks of overlapping bands averaged per bin;
  * mask x STFT (complex), iSTFT.

Runner: the chunked inference of the training project (``demix_track``): chunks of ``chunk_size`` every ``chunk_size / num_overlap`` samples
of the reflect-padded track, linear fades of ``chunk_size / 10`` at the chunk edges (none at the track's first / last chunk), sum / counter.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

BS_FREQS_PER_BANDS = (2,) * 24 + (4,) * 12 + (12,) * 8 + (24,) * 8 + (48,) * 8 + (128, 129)      # 62 bands, 1025 bins


@dataclass(frozen=True)
class RoformerConfig:
    kind: str = "bs"                       # "bs" (band-split) | "mel" (mel-band)
    dim: int = 384
    depth: int = 12
    heads: int = 8
    dim_head: int = 64
    num_stems: int = 1
    n_fft: int = 2048
    hop: int = 441
    num_bands: int = 60                    # mel only
    freqs_per_bands: Tuple[int, ...] = BS_FREQS_PER_BANDS     # bs only
    sample_rate: int = 44100
    mask_estimator_depth: int = 2
    mlp_expansion_factor: int = 4
    chunk_size: int = 352800               # 8 s = 800 hops
    num_overlap: int = 4

    @property
    def n_freq(self) -> int:
        return self.n_fft // 2 + 1


# ---- band layout -------------------------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-10) / min_log_hz) / logstep, mels)


def _mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    min_log_hz, min_log_mel, logstep = 1000.0, 1000.0 / f_sp, np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filter_bank(sr: int, n_fft: int, n_mels: int) -> np.ndarray:
    """librosa.filters.mel(sr, n_fft, n_mels) (slaney scale, slaney norm, fmax = sr / 2), float32 [n_mels, n_fft / 2 + 1]"""
    fftfreqs = np.linspace(0, sr / 2.0, n_fft // 2 + 1)
    mel_f = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, n_fft // 2 + 1))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    w *= (2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels]))[:, None]
    return w.astype(np.float32)


def band_layout(cfg: RoformerConfig) -> Tuple[List[np.ndarray], np.ndarray]:
    """-> ([indices into the merged (f s) axis, per band], how many bands cover each of the 2 * n_freq merged bins)"""
    nf = cfg.n_freq
    if cfg.kind == "bs":
        assert sum(cfg.freqs_per_bands) == nf
        edges = np.cumsum((0,) + tuple(cfg.freqs_per_bands))
        per_band_f = [np.arange(edges[i], edges[i + 1]) for i in range(len(cfg.freqs_per_bands))]
    else:
        fb = mel_filter_bank(cfg.sample_rate, cfg.n_fft, cfg.num_bands)
        fb[0][0] = 1.0
        fb[-1, -1] = 1.0
        mask = fb > 0
        assert mask.any(axis=0).all(), "every frequency must be covered by a band"
        per_band_f = [np.nonzero(mask[i])[0] for i in range(cfg.num_bands)]
    bands = [(f[:, None] * 2 + np.arange(2)[None]).reshape(-1) for f in per_band_f]       # stereo: (f s) index = 2 f + s
    cover = np.zeros(2 * nf, dtype=np.int64)
    for b in bands:
        cover[b] += 1
    return bands, cover


# ---- layers ------------------------------------------------------------------------------------------------------------------
def _rmsnorm(w, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * w[p + ".gamma"]


def _rotary(t: torch.Tensor, dim_head: int) -> torch.Tensor:
    """rotary_embedding_torch.RotaryEmbedding(dim=dim_head).rotate_queries_or_keys on [..., n, d]: interleaved pairs"""
    n = t.shape[-2]
    inv = 1.0 / (10000 ** (torch.arange(0, dim_head, 2)[: dim_head // 2].float() / dim_head))
    freqs = torch.einsum("n,f->nf", torch.arange(n).float(), inv).repeat_interleave(2, dim=-1)     # [n, d]
    x = t.reshape(*t.shape[:-1], dim_head // 2, 2)
    rot = torch.stack((-x[..., 1], x[..., 0]), dim=-1).reshape(t.shape)
    return t * freqs.cos() + rot * freqs.sin()


def _h(t: torch.Tensor) -> torch.Tensor:
    """an MFMA operand of the half-precision mode: the float32 value rounded to IEEE half (products and sums stay float32)"""
    return t.half().float()


def _linear(x: torch.Tensor, wt: torch.Tensor, bias, half: bool, padded: bool = False) -> torch.Tensor:
    """a Linear layer; ``half``: its input (stored as IEEE half by the producing kernel) and its weights are half, products exact,
    accumulation and output float32 (an output that is stored as half is rounded by the caller).  ``padded`` is documentation: the
    per-band layers reach the kernel zero-padded, which changes nothing in the arithmetic."""
    if half:
        return F.linear(_h(x), _h(wt), bias)
    return F.linear(x, wt, bias)


def _attention(cfg, w, p: str, x: torch.Tensor, half: bool = False) -> torch.Tensor:
    h, d = cfg.heads, cfg.dim_head
    xn = _rmsnorm(w, p + ".norm", x)
    qkv = _linear(xn, w[p + ".to_qkv.weight"], None, half)
    if half:
        qkv = _h(qkv)                                              # the projection is stored as IEEE half (what the attention kernel reads)
    b, n, _ = qkv.shape
    q, k, v = qkv.view(b, n, 3, h, d).permute(2, 0, 3, 1, 4)
    q, k = _rotary(q, d), _rotary(k, d)
    if half:                                                       # the one-pass kernel: q d^-1/2, k, v and the un-normalised probabilities are the
        s = _h(q * d ** -0.5) @ _h(k).transpose(-1, -2)            # f16 operands; row statistics and the final division are float32
        e = torch.exp(s - s.amax(dim=-1, keepdim=True))
        out = (_h(e) @ _h(v)) / e.sum(dim=-1, keepdim=True)
    else:
        att = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1)
        out = att @ v
    gates = _linear(xn, w[p + ".to_gates.weight"], w[p + ".to_gates.bias"], half)
    out = out * gates.permute(0, 2, 1)[..., None].sigmoid()            # (stored as IEEE half in the half mode: the rounding of _linear's input)
    return _linear(out.permute(0, 2, 1, 3).reshape(b, n, h * d), w[p + ".to_out.0.weight"], None, half)


def _feedforward(w, p: str, x: torch.Tensor, half: bool = False) -> torch.Tensor:
    h = _rmsnorm(w, p + ".net.0", x)
    h = F.gelu(_linear(h, w[p + ".net.1.weight"], w[p + ".net.1.bias"], half))
    return _linear(h, w[p + ".net.4.weight"], w[p + ".net.4.bias"], half)


def _transformer(cfg, w, p: str, x: torch.Tensor, half: bool = False) -> torch.Tensor:
    """Transformer(depth=1, norm_output=False): x = attn(x) + x; x = ff(x) + x"""
    x = _attention(cfg, w, p + ".layers.0.0", x, half) + x
    return _feedforward(w, p + ".layers.0.1", x, half) + x


@torch.no_grad()
def forward(cfg: RoformerConfig, w: Dict[str, torch.Tensor], audio: torch.Tensor, half: bool = False) -> torch.Tensor:
    """audio [B, 2, L] (L a multiple of hop) -> [B, num_stems, 2, L].  ``half=True`` restates the build's half-precision mode (the
    arithmetic of the reference's use_autocast=True as csrc/nn_half.hip runs it): the Linear layers of the transformer blocks and the
    mask estimators and the two attention products take operands rounded to IEEE half, everything else -- including every accumulation
    and every stored activation -- stays float32."""
    B, S, L = audio.shape
    win = torch.hann_window(cfg.n_fft)
    z = torch.stft(audio.reshape(B * S, L), cfg.n_fft, cfg.hop, win_length=cfg.n_fft, window=win, return_complex=True)
    Fq, T = z.shape[-2:]
    z = torch.view_as_real(z).view(B, S, Fq, T, 2).permute(0, 2, 1, 3, 4).reshape(B, Fq * S, T, 2)      # 'b s f t c -> b (f s) t c'
    bands, cover = band_layout(cfg)
    x_bands = []
    for i, idx in enumerate(bands):
        feat = z[:, torch.from_numpy(idx)].permute(0, 2, 1, 3).reshape(B, T, -1)                     # 'b f t c -> b t (f c)'
        feat = _rmsnorm(w, f"band_split.to_features.{i}.0", feat)
        x_bands.append(_linear(feat, w[f"band_split.to_features.{i}.1.weight"], w[f"band_split.to_features.{i}.1.bias"], half, padded=True))
    x = torch.stack(x_bands, dim=-2)                                                                   # [B, T, bands, dim]
    nb = len(bands)
    for li in range(cfg.depth):
        x = x.permute(0, 2, 1, 3).reshape(B * nb, T, cfg.dim)
        x = _transformer(cfg, w, f"layers.{li}.0", x, half)
        x = x.view(B, nb, T, cfg.dim).permute(0, 2, 1, 3).reshape(B * T, nb, cfg.dim)
        x = _transformer(cfg, w, f"layers.{li}.1", x, half).view(B, T, nb, cfg.dim)
    x = _rmsnorm(w, "final_norm", x)
    zc = torch.view_as_complex(z.contiguous())                                                         # [B, (f s), T]
    outs = []
    for s in range(cfg.num_stems):
        summed = torch.zeros(B, 2 * Fq, T, dtype=torch.complex64)
        for i, idx in enumerate(bands):
            p = f"mask_estimators.{s}.to_freqs.{i}.0.net"
            hcur = x[:, :, i]
            nl = cfg.mask_estimator_depth
            for j in range(nl):
                hcur = _linear(hcur, w[f"{p}.{2 * j}.weight"], w[f"{p}.{2 * j}.bias"], half, padded=True)
                if j + 1 < nl:
                    hcur = torch.tanh(hcur)
            m = F.glu(hcur, dim=-1)                                                                    # [B, T, len(idx) * 2]
            m = torch.view_as_complex(m.view(B, T, len(idx), 2).permute(0, 2, 1, 3).contiguous())   # 'b t (f c) -> b f t c'
            summed[:, torch.from_numpy(idx)] += m
        mask = summed / torch.from_numpy(cover).clamp(min=1).view(1, -1, 1)
        zs = (zc * mask).view(B, Fq, S, T).permute(0, 2, 1, 3).reshape(B * S, Fq, T)                # 'b (f s) t -> (b s) f t'
        y = torch.istft(zs, cfg.n_fft, cfg.hop, win_length=cfg.n_fft, window=win, return_complex=False, length=L)
        outs.append(y.view(B, S, L))
    return torch.stack(outs, dim=1)


# ---- runner ------------------------------------------------------------------------------------------------------------------
def demix_plan(cfg: RoformerConfig, length_init: int):
    C = cfg.chunk_size
    step = C // cfg.num_overlap
    border = C - step
    padded = length_init > 2 * border and border > 0
    total = length_init + 2 * border if padded else length_init
    starts = list(range(0, total, step))
    return C, step, border, padded, total, starts


def windows(C: int) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    fade = C // 10
    fadein, fadeout = torch.linspace(0, 1, fade), torch.linspace(1, 0, fade)
    start, middle, finish = torch.ones(C), torch.ones(C), torch.ones(C)
    start[-fade:] *= fadeout
    finish[:fade] *= fadein
    middle[-fade:] *= fadeout
    middle[:fade] *= fadein
    return start, middle, finish


@torch.no_grad()
def demix_track(cfg: RoformerConfig, w: Dict[str, torch.Tensor], mix: torch.Tensor, fwd=None) -> torch.Tensor:
    """mix [2, L] -> [num_stems, 2, L]"""
    fwd = fwd or (lambda x: forward(cfg, w, x))
    L0 = mix.shape[-1]
    C, step, border, padded, total, starts = demix_plan(cfg, L0)
    if padded:
        mix = F.pad(mix[None], (border, border), mode="reflect")[0]
    w_start, w_middle, w_finish = windows(C)
    result = torch.zeros(cfg.num_stems, 2, total)
    counter = torch.zeros(total)
    for i in starts:
        part = mix[:, i:i + C]
        length = part.shape[-1]
        if length < C:
            mode = "reflect" if length > C // 2 + 1 else "constant"
            part = F.pad(part[None], (0, C - length), mode=mode)[0]
        y = fwd(part[None])[0]
        win = w_middle
        if i == 0:
            win = w_start
        elif i + step >= total:
            win = w_finish
        result[..., i:i + length] += y[..., :length] * win[:length]
        counter[i:i + length] += win[:length]
    out = torch.nan_to_num(result / counter, nan=0.0)
    return out[..., border:-border] if padded else out


# ---- synthetic weights (data only) ---------------------------------------------------------------------------------------------
def synthetic_state_dict(cfg: RoformerConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def lin(p, out, inp, bias=True, gain=1.0):
        b = gain / math.sqrt(inp)
        sd[p + ".weight"] = (torch.rand(out, inp, generator=g) * 2 - 1) * b
        if bias:
            sd[p + ".bias"] = (torch.rand(out, generator=g) * 2 - 1) * b

    def gamma(p, n):
        sd[p + ".gamma"] = 1.0 + 0.1 * (torch.rand(n, generator=g) * 2 - 1)

    bands, _ = band_layout(cfg)
    inner = cfg.heads * cfg.dim_head
    for i, idx in enumerate(bands):
        gamma(f"band_split.to_features.{i}.0", 2 * len(idx))
        lin(f"band_split.to_features.{i}.1", cfg.dim, 2 * len(idx))
    for li in range(cfg.depth):
        for tr in (0, 1):
            p = f"layers.{li}.{tr}.layers.0"
            gamma(p + ".0.norm", cfg.dim)
            lin(p + ".0.to_qkv", 3 * inner, cfg.dim, bias=False, gain=2.0)
            lin(p + ".0.to_gates", cfg.heads, cfg.dim)
            lin(p + ".0.to_out.0", cfg.dim, inner, bias=False)
            gamma(p + ".1.net.0", cfg.dim)
            lin(p + ".1.net.1", 4 * cfg.dim, cfg.dim)
            lin(p + ".1.net.4", cfg.dim, 4 * cfg.dim)
    gamma("final_norm", cfg.dim)
    hidden = cfg.dim * cfg.mlp_expansion_factor
    for s in range(cfg.num_stems):
        for i, idx in enumerate(bands):
            p = f"mask_estimators.{s}.to_freqs.{i}.0.net"
            dims = (cfg.dim,) + (hidden,) * (cfg.mask_estimator_depth - 1) + (2 * len(idx) * 2,)
            for j in range(cfg.mask_estimator_depth):
                lin(f"{p}.{2 * j}", dims[j + 1], dims[j], gain=1.5)
    return sd
