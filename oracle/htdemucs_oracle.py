"""Plain torch-CPU fp32 restatement of HTDemucs (Hybrid Transformer Demucs) and of demucs.apply.apply_model.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Not imported by the product.

PARITY UNPINNED.  The reference reaches this network only through the third-party packages
``audio-separator[gpu]>=0.32.0`` (setup.sh:96) -> ``demucs>=4.0.1`` (requirements.txt:19):
``self.separator.load_model("htdemucs_6s.yaml")`` / ``.separate(tmp_wav)`` at
/root/reference/modules/separator/stem_separator.py:466,479, outputs mapped to drums / bass / guitar / piano / other by
file-name substring (:491-500).  Neither package is in /root/reference or in this image and there is no network, so the
network below restates the published demucs v4 design (``demucs/htdemucs.py``, ``hdemucs.py``, ``demucs.py``,
``transformer.py``, ``apply.py``) with that package's parameter names, so that a real ``state_dict`` would load:

  * time branch: HEncLayer(Conv1d K=8 S=4 pad 2, GELU, DConv residual branch, 1x1 rewrite + GLU) x depth, mirrored by
    HDecLayer(3-tap rewrite + GLU, ConvTranspose1d K=8 S=4, crop, GELU);
  * frequency branch on the n_fft 4096 / hop 1024 complex-as-channels spectrogram: the same layers with [8,1] / [4,1]
    kernels along frequency, a scaled frequency embedding after the first layer;
  * DConv: depth x (dilated Conv1d 3 -> GroupNorm(1) -> GELU -> Conv1d 1 -> GroupNorm(1) -> GLU -> LayerScale), residual;
  * cross-transformer at the bottleneck (channels up-sampled to ``bottom_channels``): 5 layers alternating self-attention and
    cross-attention between the branches, pre-norm, LayerScale, GELU FFN x4, a 1-group GroupNorm on each layer's output,
    2-D / 1-D sinusoidal position embeddings;
  * outputs: iSTFT of the (de-normalised) complex-as-channels output + the time branch.

apply_model: ``shifts`` passes with (here: fixed, seeded) time offsets, each split into ``segment``-long chunks every
``(1 - overlap) * segment`` samples, triangular weights, zero-padded to the training length inside the network.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class HTDemucsConfig:
    sources: Tuple[str, ...] = ("drums", "bass", "other", "vocals", "guitar", "piano")     # htdemucs_6s
    audio_channels: int = 2
    channels: int = 48
    growth: int = 2
    nfft: int = 4096
    depth: int = 4
    kernel_size: int = 8
    stride: int = 4
    context: int = 1
    context_enc: int = 0
    dconv_depth: int = 2
    dconv_comp: int = 8
    dconv_init: float = 1e-3
    freq_emb: float = 0.2
    emb_scale: float = 10.0
    bottom_channels: int = 512
    t_layers: int = 5
    t_heads: int = 8
    t_hidden_scale: float = 4.0
    t_max_period: float = 10000.0
    t_weight_pos_embed: float = 1.0
    samplerate: int = 44100
    segment_samples: int = 343980            # 7.8 s (Fraction(39, 5) * 44100): the training length every chunk is padded to

    @property
    def hop(self) -> int:
        return self.nfft // 4

    @property
    def S(self) -> int:
        return len(self.sources)

    def widths(self) -> List[int]:
        return [self.channels * self.growth ** i for i in range(self.depth)]


def _pad1d_reflect(x: torch.Tensor, left: int, right: int) -> torch.Tensor:
    """demucs.hdemucs.pad1d: reflect padding that tolerates inputs shorter than the padding"""
    length = x.shape[-1]
    max_pad = max(left, right)
    if length <= max_pad:
        extra = max_pad - length + 1
        extra_right = min(right, extra)
        extra_left = extra - extra_right
        x = F.pad(x, (extra_left, extra_right))
        left, right = left - extra_left, right - extra_right
    return F.pad(x, (left, right), mode="reflect")


def spec(cfg: HTDemucsConfig, x: torch.Tensor) -> torch.Tensor:
    """HTDemucs._spec: [B,C,L] -> complex [B,C,nfft/2,le]"""
    hl, nfft = cfg.hop, cfg.nfft
    le = int(math.ceil(x.shape[-1] / hl))
    pad = hl // 2 * 3
    x = _pad1d_reflect(x, pad, pad + le * hl - x.shape[-1])
    b, c, n = x.shape
    z = torch.stft(x.reshape(-1, n), nfft, hl, window=torch.hann_window(nfft).to(x), win_length=nfft, normalized=True, center=True,
                   return_complex=True, pad_mode="reflect")
    z = z.reshape(b, c, z.shape[-2], z.shape[-1])[..., :-1, :]
    return z[..., 2:2 + le]


def ispec(cfg: HTDemucsConfig, z: torch.Tensor, length: int) -> torch.Tensor:
    """HTDemucs._ispec: complex [...,nfft/2,le] -> [...,length]"""
    hl = cfg.hop
    z = F.pad(z, (0, 0, 0, 1))
    z = F.pad(z, (2, 2))
    pad = hl // 2 * 3
    le = hl * int(math.ceil(length / hl)) + 2 * pad
    lead = z.shape[:-2]
    zz = z.reshape(-1, z.shape[-2], z.shape[-1])
    x = torch.istft(zz, cfg.nfft, hl, window=torch.hann_window(cfg.nfft).to(z.real), win_length=cfg.nfft, normalized=True, length=le,
                    center=True)
    return x.reshape(*lead, le)[..., pad:pad + length]


def create_sin_embedding(length: int, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """demucs.transformer.create_sin_embedding (shift 0): [length, 1, dim]"""
    pos = torch.arange(length).view(-1, 1, 1).float()
    half = dim // 2
    adim = torch.arange(half).view(1, 1, -1).float()
    phase = pos / (max_period ** (adim / (half - 1)))
    return torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1)


def create_2d_sin_embedding(d_model: int, height: int, width: int, max_period: float = 10000.0) -> torch.Tensor:
    """demucs.transformer.create_2d_sin_embedding: [1, d_model, height, width]"""
    pe = torch.zeros(d_model, height, width)
    dm = d_model // 2
    div_term = torch.exp(torch.arange(0.0, dm, 2) * -(math.log(max_period) / dm))
    pos_w = torch.arange(0.0, width).unsqueeze(1)
    pos_h = torch.arange(0.0, height).unsqueeze(1)
    pe[0:dm:2] = torch.sin(pos_w * div_term).transpose(0, 1).unsqueeze(1).repeat(1, height, 1)
    pe[1:dm:2] = torch.cos(pos_w * div_term).transpose(0, 1).unsqueeze(1).repeat(1, height, 1)
    pe[dm::2] = torch.sin(pos_h * div_term).transpose(0, 1).unsqueeze(2).repeat(1, 1, width)
    pe[dm + 1::2] = torch.cos(pos_h * div_term).transpose(0, 1).unsqueeze(2).repeat(1, 1, width)
    return pe[None]


def _dconv(w: Dict[str, torch.Tensor], p: str, x: torch.Tensor, depth: int) -> torch.Tensor:
    """demucs.demucs.DConv on [N, C, T]"""
    for d in range(depth):
        q = f"{p}.layers.{d}"
        dil = 2 ** d
        y = F.conv1d(x, w[q + ".0.weight"], w[q + ".0.bias"], dilation=dil, padding=dil)
        y = F.gelu(F.group_norm(y, 1, w[q + ".1.weight"], w[q + ".1.bias"], eps=1e-5))
        y = F.conv1d(y, w[q + ".3.weight"], w[q + ".3.bias"])
        y = F.glu(F.group_norm(y, 1, w[q + ".4.weight"], w[q + ".4.bias"], eps=1e-5), dim=1)
        x = x + w[q + ".6.scale"][:, None] * y
    return x


def _enc_layer(cfg, w, p: str, x: torch.Tensor, inject, freq: bool) -> torch.Tensor:
    """demucs.hdemucs.HEncLayer.forward (norm off, dconv on, rewrite on, context_enc 0)"""
    K, S = cfg.kernel_size, cfg.stride
    if freq:
        y = F.conv2d(x, w[p + ".conv.weight"], w[p + ".conv.bias"], stride=(S, 1), padding=(K // 4, 0))
    else:
        le = x.shape[-1]
        if le % S:
            x = F.pad(x, (0, S - le % S))
        y = F.conv1d(x, w[p + ".conv.weight"], w[p + ".conv.bias"], stride=S, padding=K // 4)
    if inject is not None:
        y = y + (inject[:, :, None] if inject.dim() == 3 and y.dim() == 4 else inject)
    y = F.gelu(y)
    if freq:
        b, c, fr, t = y.shape
        y = _dconv(w, p + ".dconv", y.permute(0, 2, 1, 3).reshape(-1, c, t), cfg.dconv_depth).view(b, fr, c, t).permute(0, 2, 1, 3)
        z = F.conv2d(y, w[p + ".rewrite.weight"], w[p + ".rewrite.bias"], padding=cfg.context_enc)
    else:
        y = _dconv(w, p + ".dconv", y, cfg.dconv_depth)
        z = F.conv1d(y, w[p + ".rewrite.weight"], w[p + ".rewrite.bias"], padding=cfg.context_enc)
    return F.glu(z, dim=1)


def _dec_layer(cfg, w, p: str, x: torch.Tensor, skip: torch.Tensor, length: int, freq: bool, last: bool) -> torch.Tensor:
    """demucs.hdemucs.HDecLayer.forward (norm off, dconv off, rewrite on, context 1)"""
    K, S = cfg.kernel_size, cfg.stride
    pad = K // 4
    x = x + skip
    if freq:
        y = F.glu(F.conv2d(x, w[p + ".rewrite.weight"], w[p + ".rewrite.bias"], padding=cfg.context), dim=1)
        z = F.conv_transpose2d(y, w[p + ".conv_tr.weight"], w[p + ".conv_tr.bias"], stride=(S, 1))
        z = z[..., pad:-pad, :]
    else:
        y = F.glu(F.conv1d(x, w[p + ".rewrite.weight"], w[p + ".rewrite.bias"], padding=cfg.context), dim=1)
        z = F.conv_transpose1d(y, w[p + ".conv_tr.weight"], w[p + ".conv_tr.bias"], stride=S)
        z = z[..., pad:pad + length]
        assert z.shape[-1] == length, (z.shape, length)
    return z if last else F.gelu(z)


def _my_group_norm(x: torch.Tensor, wt: torch.Tensor, bs: torch.Tensor) -> torch.Tensor:
    """demucs.transformer.MyGroupNorm (1 group) on [B, T, C]"""
    return F.group_norm(x.transpose(1, 2), 1, wt, bs, eps=1e-5).transpose(1, 2)


def _mha(w, p: str, q: torch.Tensor, kv: torch.Tensor, heads: int) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True), no masks, eval mode"""
    c = q.shape[-1]
    wi, bi = w[p + ".in_proj_weight"], w[p + ".in_proj_bias"]
    qq = F.linear(q, wi[:c], bi[:c])
    kk = F.linear(kv, wi[c:2 * c], bi[c:2 * c])
    vv = F.linear(kv, wi[2 * c:], bi[2 * c:])
    b, tq, _ = qq.shape
    tk = kk.shape[1]
    dh = c // heads
    qh = qq.view(b, tq, heads, dh).transpose(1, 2)
    kh = kk.view(b, tk, heads, dh).transpose(1, 2)
    vh = vv.view(b, tk, heads, dh).transpose(1, 2)
    att = torch.softmax(qh @ kh.transpose(-1, -2) / math.sqrt(dh), dim=-1)
    out = (att @ vh).transpose(1, 2).reshape(b, tq, c)
    return F.linear(out, w[p + ".out_proj.weight"], w[p + ".out_proj.bias"])


def _ln(w, p: str, x: torch.Tensor) -> torch.Tensor:
    return F.layer_norm(x, (x.shape[-1],), w[p + ".weight"], w[p + ".bias"], eps=1e-5)


def _tlayer(w, p: str, x: torch.Tensor, other, heads: int) -> torch.Tensor:
    """MyTransformerEncoderLayer (other is None) / CrossTransformerEncoderLayer: norm_first, layer_scale, norm_out, GELU"""
    if other is None:
        h = _ln(w, p + ".norm1", x)
        x = x + w[p + ".gamma_1.scale"] * _mha(w, p + ".self_attn", h, h, heads)
        h = _ln(w, p + ".norm2", x)
    else:
        x = x + w[p + ".gamma_1.scale"] * _mha(w, p + ".cross_attn", _ln(w, p + ".norm1", x), _ln(w, p + ".norm2", other), heads)
        h = _ln(w, p + ".norm3", x)
    ff = F.linear(F.gelu(F.linear(h, w[p + ".linear1.weight"], w[p + ".linear1.bias"])), w[p + ".linear2.weight"], w[p + ".linear2.bias"])
    x = x + w[p + ".gamma_2.scale"] * ff
    return _my_group_norm(x, w[p + ".norm_out.weight"], w[p + ".norm_out.bias"])


def _cross_transformer(cfg, w, x: torch.Tensor, xt: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """demucs.transformer.CrossTransformerEncoder.forward (emb 'sin', cross_first False, norm_in, no sparsity)"""
    p = "crosstransformer"
    b, c, fr, t1 = x.shape
    pos2d = create_2d_sin_embedding(c, fr, t1, cfg.t_max_period).permute(0, 3, 2, 1).reshape(1, t1 * fr, c)
    x = x.permute(0, 3, 2, 1).reshape(b, t1 * fr, c)                  # "b c fr t1 -> b (t1 fr) c"
    x = _ln(w, p + ".norm_in", x) + cfg.t_weight_pos_embed * pos2d
    t2 = xt.shape[-1]
    xt = xt.transpose(1, 2)
    xt = _ln(w, p + ".norm_in_t", xt) + cfg.t_weight_pos_embed * create_sin_embedding(t2, c, cfg.t_max_period).transpose(0, 1)
    for idx in range(cfg.t_layers):
        if idx % 2 == 0:
            x = _tlayer(w, f"{p}.layers.{idx}", x, None, cfg.t_heads)
            xt = _tlayer(w, f"{p}.layers_t.{idx}", xt, None, cfg.t_heads)
        else:
            old_x = x
            x = _tlayer(w, f"{p}.layers.{idx}", x, xt, cfg.t_heads)
            xt = _tlayer(w, f"{p}.layers_t.{idx}", xt, old_x, cfg.t_heads)
    x = x.reshape(b, t1, fr, c).permute(0, 3, 2, 1)
    return x, xt.transpose(1, 2)


@torch.no_grad()
def forward(cfg: HTDemucsConfig, w: Dict[str, torch.Tensor], mix: torch.Tensor) -> torch.Tensor:
    """HTDemucs.forward in eval mode: mix [B, 2, L] (L <= segment_samples) -> [B, S, 2, L]"""
    length_pre_pad = None
    if mix.shape[-1] < cfg.segment_samples:
        length_pre_pad = mix.shape[-1]
        mix = F.pad(mix, (0, cfg.segment_samples - length_pre_pad))
    length = mix.shape[-1]
    z = spec(cfg, mix)
    m = torch.view_as_real(z).permute(0, 1, 4, 2, 3)
    B, C, _, Fq, T = m.shape
    x = m.reshape(B, C * 2, Fq, T)
    mean = x.mean(dim=(1, 2, 3), keepdim=True)
    std = x.std(dim=(1, 2, 3), keepdim=True)
    x = (x - mean) / (1e-5 + std)
    xt = mix
    meant = xt.mean(dim=(1, 2), keepdim=True)
    stdt = xt.std(dim=(1, 2), keepdim=True)
    xt = (xt - meant) / (1e-5 + stdt)
    saved, saved_t, lengths, lengths_t = [], [], [], []
    for idx in range(cfg.depth):
        lengths.append(x.shape[-1])
        lengths_t.append(xt.shape[-1])
        xt = _enc_layer(cfg, w, f"tencoder.{idx}", xt, None, freq=False)
        saved_t.append(xt)
        x = _enc_layer(cfg, w, f"encoder.{idx}", x, None, freq=True)
        if idx == 0:
            emb = (w["freq_emb.embedding.weight"] * cfg.emb_scale).t()[None, :, :, None].expand_as(x)
            x = x + cfg.freq_emb * emb
        saved.append(x)
    b, c, f, t = x.shape
    x = F.conv1d(x.reshape(b, c, f * t), w["channel_upsampler.weight"], w["channel_upsampler.bias"]).reshape(b, -1, f, t)
    xt = F.conv1d(xt, w["channel_upsampler_t.weight"], w["channel_upsampler_t.bias"])
    x, xt = _cross_transformer(cfg, w, x, xt)
    x = F.conv1d(x.reshape(b, -1, f * t), w["channel_downsampler.weight"], w["channel_downsampler.bias"]).reshape(b, c, f, t)
    xt = F.conv1d(xt, w["channel_downsampler_t.weight"], w["channel_downsampler_t.bias"])
    for idx in range(cfg.depth):
        last = idx == cfg.depth - 1
        x = _dec_layer(cfg, w, f"decoder.{idx}", x, saved.pop(-1), lengths.pop(-1), freq=True, last=last)
        xt = _dec_layer(cfg, w, f"tdecoder.{idx}", xt, saved_t.pop(-1), lengths_t.pop(-1), freq=False, last=last)
    S = cfg.S
    x = x.view(B, S, -1, Fq, T) * std[:, None] + mean[:, None]
    zout = torch.view_as_complex(x.view(B, S, -1, 2, Fq, T).permute(0, 1, 2, 4, 5, 3).contiguous())
    x = ispec(cfg, zout, length)
    xt = xt.view(B, S, -1, length) * stdt[:, None] + meant[:, None]
    x = xt + x
    if length_pre_pad:
        x = x[..., :length_pre_pad]
    return x


def shift_offsets(shifts: int, max_shift: int, seed: int = 0) -> List[int]:
    """the time offsets of the ``shifts`` passes.  demucs draws them with random.randint per call; this build fixes them by a
    seeded generator so that a separation is reproducible"""
    g = torch.Generator().manual_seed(seed)
    return [int(torch.randint(0, max_shift + 1, (1,), generator=g)) for _ in range(shifts)]


def segment_plan(length: int, segment: int, overlap: float) -> Tuple[List[int], torch.Tensor]:
    """offsets of the chunks and the triangular weight (demucs.apply.apply_model, split=True, transition_power 1)"""
    stride = int((1 - overlap) * segment)
    offsets = list(range(0, length, stride))
    weight = torch.cat([torch.arange(1, segment // 2 + 1), torch.arange(segment - segment // 2, 0, -1)]).float()
    return offsets, weight / weight.max()


def _run_split(cfg, root: torch.Tensor, base: int, length: int, overlap: float, fwd) -> torch.Tensor:
    """split=True over the view root[..., base : base + length] (a demucs TensorChunk): -> [B,S,2,length].  A chunk shorter than
    the segment is padded to it the way TensorChunk.padded does: centred, with the REAL samples of ``root`` around the chunk
    where they exist and zeros beyond its ends; the network output is centre-trimmed back to the chunk."""
    B, C, total = root.shape
    seg = cfg.segment_samples
    offsets, weight = segment_plan(length, seg, overlap)
    out = torch.zeros(B, cfg.S, C, length)
    sum_weight = torch.zeros(length)
    for off in offsets:
        cl = min(length - off, seg)
        delta = seg - cl
        start = base + off - delta // 2
        end = start + seg
        cs, ce = max(0, start), min(total, end)
        padded = F.pad(root[..., cs:ce], (cs - start, end - ce))
        y = fwd(padded)
        y = y[..., delta // 2: delta // 2 + cl]                 # center_trim
        out[..., off:off + cl] += weight[:cl] * y
        sum_weight[off:off + cl] += weight[:cl]
    return out / sum_weight


@torch.no_grad()
def apply_model(cfg: HTDemucsConfig, w: Dict[str, torch.Tensor], mix: torch.Tensor, shifts: int = 2, overlap: float = 0.25,
                seed: int = 0, fwd=None) -> torch.Tensor:
    """demucs.apply.apply_model(model, mix, shifts, split=True, overlap): mix [B,2,L] -> [B,S,2,L]"""
    fwd = fwd or (lambda x: forward(cfg, w, x))
    length = mix.shape[-1]
    if not shifts:
        return _run_split(cfg, mix, 0, length, overlap, fwd)
    max_shift = int(0.5 * cfg.samplerate)
    padded = F.pad(mix, (max_shift, max_shift))               # TensorChunk(mix).padded(length + 2 * max_shift)
    out = 0.0
    for offset in shift_offsets(shifts, max_shift, seed):
        # shifted = TensorChunk(padded_mix, offset, length + max_shift - offset); out += shifted_out[..., max_shift - offset:]
        res = _run_split(cfg, padded, offset, length + max_shift - offset, overlap, fwd)
        out = out + res[..., max_shift - offset:]
    return out / shifts


def separate(cfg: HTDemucsConfig, w: Dict[str, torch.Tensor], mix: torch.Tensor, shifts: int = 2, overlap: float = 0.25, seed: int = 0,
             fwd=None) -> torch.Tensor:
    """audio_separator DemucsSeparator.demix_demucs around apply_model: whole-track normalisation by the mono reference.
    mix [2, L] -> [S, 2, L]"""
    ref = mix.mean(0)
    m, s = ref.mean(), ref.std()
    out = apply_model(cfg, w, ((mix - m) / s)[None], shifts=shifts, overlap=overlap, seed=seed, fwd=fwd)[0]
    return out * s + m


# ---- synthetic weights (data only: random-init parameters with demucs' names and shapes) ------------------------------------
def synthetic_state_dict(cfg: HTDemucsConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}

    def uni(shape, fan_in):
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b

    def conv(p, cout, cin, *k, transposed=False, gain=1.0):
        fan = cin * int(torch.tensor(k).prod()) if k else cin
        shape = (cin, cout, *k) if transposed else (cout, cin, *k)
        sd[p + ".weight"] = uni(shape, fan if not transposed else cout * int(torch.tensor(k).prod())) * gain
        sd[p + ".bias"] = uni((cout,), fan)

    def norm(p, c):
        sd[p + ".weight"] = 1.0 + 0.1 * (torch.rand(c, generator=g) * 2 - 1)
        sd[p + ".bias"] = 0.05 * (torch.rand(c, generator=g) * 2 - 1)

    def dconv(p, c):
        hidden = c // cfg.dconv_comp
        for d in range(cfg.dconv_depth):
            q = f"{p}.layers.{d}"
            conv(q + ".0", hidden, c, 3)
            norm(q + ".1", hidden)
            conv(q + ".3", 2 * c, hidden, 1)
            norm(q + ".4", 2 * c)
            sd[q + ".6.scale"] = torch.full((c,), 0.2) * (0.5 + torch.rand(c, generator=g))   # demucs inits 1e-3; larger so the branch matters

    chin = cfg.audio_channels
    chin_z = chin * 2
    for idx, chout in enumerate(cfg.widths()):
        K = cfg.kernel_size
        conv(f"encoder.{idx}.conv", chout, chin_z, K, 1)
        conv(f"encoder.{idx}.rewrite", 2 * chout, chout, 1 + 2 * cfg.context_enc, 1 + 2 * cfg.context_enc)
        dconv(f"encoder.{idx}.dconv", chout)
        conv(f"tencoder.{idx}.conv", chout, chin, K)
        conv(f"tencoder.{idx}.rewrite", 2 * chout, chout, 1 + 2 * cfg.context_enc)
        dconv(f"tencoder.{idx}.dconv", chout)
        if idx == 0:
            chin = cfg.audio_channels * cfg.S
            chin_z = chin * 2
        di = cfg.depth - 1 - idx                                  # decoders are stored outermost-last
        conv(f"decoder.{di}.conv_tr", chin_z, chout, K, 1, transposed=True)
        conv(f"decoder.{di}.rewrite", 2 * chout, chout, 1 + 2 * cfg.context, 1 + 2 * cfg.context)
        conv(f"tdecoder.{di}.conv_tr", chin, chout, K, transposed=True)
        conv(f"tdecoder.{di}.rewrite", 2 * chout, chout, 1 + 2 * cfg.context)
        chin = chout
        chin_z = chout
    freqs = cfg.nfft // 2 // cfg.stride
    sd["freq_emb.embedding.weight"] = torch.randn(freqs, cfg.channels, generator=g) / cfg.emb_scale
    cb, cd = cfg.widths()[-1], cfg.bottom_channels
    for name, (co, ci) in {"channel_upsampler": (cd, cb), "channel_upsampler_t": (cd, cb), "channel_downsampler": (cb, cd),
                           "channel_downsampler_t": (cb, cd)}.items():
        conv(name, co, ci, 1)
    p = "crosstransformer"
    norm(p + ".norm_in", cd)
    norm(p + ".norm_in_t", cd)
    hidden = int(cd * cfg.t_hidden_scale)
    for branch in ("layers", "layers_t"):
        for idx in range(cfg.t_layers):
            q = f"{p}.{branch}.{idx}"
            att = "self_attn" if idx % 2 == 0 else "cross_attn"
            sd[f"{q}.{att}.in_proj_weight"] = uni((3 * cd, cd), cd)
            sd[f"{q}.{att}.in_proj_bias"] = uni((3 * cd,), cd)
            conv(f"{q}.{att}.out_proj", cd, cd)
            conv(f"{q}.linear1", hidden, cd)
            conv(f"{q}.linear2", cd, hidden)
            for n in ("norm1", "norm2") + (("norm3",) if idx % 2 else ()):
                norm(f"{q}.{n}", cd)
            norm(f"{q}.norm_out", cd)
            for gname in ("gamma_1", "gamma_2"):
                sd[f"{q}.{gname}.scale"] = torch.full((cd,), 0.3) * (0.5 + torch.rand(cd, generator=g))   # demucs inits 1e-4
    return sd
