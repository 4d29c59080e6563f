"""HTDemucs (Hybrid Transformer Demucs, ``htdemucs_6s.yaml``) on the GPU -- the network and the chunked runner behind
``self.separator.load_model("htdemucs_6s.yaml")`` / ``.separate(...)`` of the reference's multi-stem stage
(modules/separator/stem_separator.py:459-503: the model runs on the FULL mix, its outputs are mapped to
drums / bass / guitar / piano / other by file-name substring, the vocals output is ignored).

The network lives in the un-vendored ``demucs>=4.0.1`` (requirements.txt:19), reached through
``audio-separator[gpu]>=0.32.0`` (setup.sh:96): PARITY UNPINNED -- restated from the published design
(oracle/htdemucs_oracle.py is the torch-CPU fp32 twin the kernels are checked against).  Parameter names are demucs'
(``encoder.0.conv.weight``, ``tdecoder.3.conv_tr.bias``, ``crosstransformer.layers_t.1.cross_attn.in_proj_weight`` ...),
so a real ``state_dict`` loads as it is.

Everything runs in libalsep.so on channels-last float32 tensors: frequency branch ``[Fr, T, C]``, time branch ``[L, C]``;
convolutions and linears through ``alsep_nn_conv2d`` (exact-f32 MFMA), transposed convolutions as its 1x1 case +
``alsep_nn_tconv_fold``, GroupNorm / LayerNorm (+ GELU / GLU) through ``alsep_nn_norm``, attention as two strided batched
GEMMs around a row softmax, STFT / iSTFT by the n_fft 4096 three-pass kernels of the MDX path (fft_r16.h).  torch supplies
device memory only; the sinusoidal position tables are data computed once per shape.

Runner: ``demucs.apply.apply_model`` (shifts, split into 7.8 s segments every 75 %, triangular weights) inside
``DemucsSeparator``'s whole-track normalisation.  demucs draws the shift offsets with ``random.randint`` per call; this build
fixes them with a seeded generator.  With ``sharded=True`` the (shift, segment) units are split over the ranks of a
``torch.distributed`` group and the weighted partial sums are exchanged once (audiolab_amd.dist).
"""
from __future__ import annotations

import ctypes as C
import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import AlsepError, Context

ACT_NONE, ACT_GELU, ACT_GLU = 0, 3, 4


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
    segment_samples: int = 343980

    @property
    def hop(self) -> int:
        return self.nfft // 4

    @property
    def S(self) -> int:
        return len(self.sources)

    def widths(self) -> List[int]:
        return [self.channels * self.growth ** i for i in range(self.depth)]


def expected_shapes(cfg: HTDemucsConfig) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, hyper-parameter that fixes it) of every tensor an HTDemucs with this configuration reads (demucs' names)"""
    exp: Dict[str, Tuple[Tuple[int, ...], str]] = {}
    wch = f"channels={cfg.channels} / growth={cfg.growth} / depth={cfg.depth}"

    def conv(p, cout, cin, *k, transposed=False, hyper=wch):
        exp[p + ".weight"] = (((cin, cout) if transposed else (cout, cin)) + tuple(k), hyper)
        exp[p + ".bias"] = ((cout,), hyper)

    def norm(p, c, hyper=wch):
        exp[p + ".weight"] = ((c,), hyper)
        exp[p + ".bias"] = ((c,), hyper)

    def dconv(p, c):
        hidden = c // cfg.dconv_comp
        hy = f"dconv_comp={cfg.dconv_comp} / dconv_depth={cfg.dconv_depth} / {wch}"
        for d in range(cfg.dconv_depth):
            q = f"{p}.layers.{d}"
            conv(q + ".0", hidden, c, 3, hyper=hy)
            norm(q + ".1", hidden, hy)
            conv(q + ".3", 2 * c, hidden, 1, hyper=hy)
            norm(q + ".4", 2 * c, hy)
            exp[q + ".6.scale"] = ((c,), hy)

    K, S = cfg.kernel_size, cfg.S
    chin, chin_z = cfg.audio_channels, cfg.audio_channels * 2
    src = f"sources ({S}) / audio_channels={cfg.audio_channels}"
    for idx, chout in enumerate(cfg.widths()):
        ke = 1 + 2 * cfg.context_enc
        conv(f"encoder.{idx}.conv", chout, chin_z, K, 1, hyper=f"kernel_size={K} / {wch}")
        conv(f"encoder.{idx}.rewrite", 2 * chout, chout, ke, ke, hyper=f"context_enc={cfg.context_enc} / {wch}")
        dconv(f"encoder.{idx}.dconv", chout)
        conv(f"tencoder.{idx}.conv", chout, chin, K, hyper=f"kernel_size={K} / {wch}")
        conv(f"tencoder.{idx}.rewrite", 2 * chout, chout, ke, hyper=f"context_enc={cfg.context_enc} / {wch}")
        dconv(f"tencoder.{idx}.dconv", chout)
        if idx == 0:
            chin, chin_z = cfg.audio_channels * S, cfg.audio_channels * S * 2
        di = cfg.depth - 1 - idx
        kd = 1 + 2 * cfg.context
        hy = src if idx == 0 else wch
        conv(f"decoder.{di}.conv_tr", chin_z, chout, K, 1, transposed=True, hyper=f"kernel_size={K} / {hy}")
        conv(f"decoder.{di}.rewrite", 2 * chout, chout, kd, kd, hyper=f"context={cfg.context} / {wch}")
        conv(f"tdecoder.{di}.conv_tr", chin, chout, K, transposed=True, hyper=f"kernel_size={K} / {hy}")
        conv(f"tdecoder.{di}.rewrite", 2 * chout, chout, kd, hyper=f"context={cfg.context} / {wch}")
        chin = chin_z = chout
    exp["freq_emb.embedding.weight"] = ((cfg.nfft // 2 // cfg.stride, cfg.channels), f"nfft={cfg.nfft} / stride={cfg.stride} / channels={cfg.channels}")
    cb, cd = cfg.widths()[-1], cfg.bottom_channels
    bc = f"bottom_channels={cd}"
    for name, (co, ci) in {"channel_upsampler": (cd, cb), "channel_upsampler_t": (cd, cb), "channel_downsampler": (cb, cd),
                           "channel_downsampler_t": (cb, cd)}.items():
        conv(name, co, ci, 1, hyper=f"{bc} / {wch}")
    p = "crosstransformer"
    norm(p + ".norm_in", cd, bc)
    norm(p + ".norm_in_t", cd, bc)
    hidden = int(cd * cfg.t_hidden_scale)
    th = f"{bc} / t_hidden_scale={cfg.t_hidden_scale} / t_layers={cfg.t_layers}"
    for branch in ("layers", "layers_t"):
        for idx in range(cfg.t_layers):
            q = f"{p}.{branch}.{idx}"
            att = "self_attn" if idx % 2 == 0 else "cross_attn"
            exp[f"{q}.{att}.in_proj_weight"] = ((3 * cd, cd), th)
            exp[f"{q}.{att}.in_proj_bias"] = ((3 * cd,), th)
            conv(f"{q}.{att}.out_proj", cd, cd, hyper=th)
            conv(f"{q}.linear1", hidden, cd, hyper=th)
            conv(f"{q}.linear2", cd, hidden, hyper=th)
            for n in ("norm1", "norm2") + (("norm3",) if idx % 2 else ()) + ("norm_out",):
                norm(f"{q}.{n}", cd, th)
            for gname in ("gamma_1", "gamma_2"):
                exp[f"{q}.{gname}.scale"] = ((cd,), th)
    return exp


class _Conv:
    """weights of one convolution / linear in the layout alsep_nn_conv2d reads: [KH][KW][Cin][Cout], scale 1, shift = bias"""

    def __init__(self, ctx: Context, w4: torch.Tensor, bias: Optional[torch.Tensor]):
        self.kh, self.kw, self.cin, self.cout = (int(v) for v in w4.shape)
        self.w = w4.detach().float().contiguous().to(ctx.device)
        self.scale = torch.ones(self.cout, device=ctx.device)
        self.shift = (bias.detach().float() if bias is not None else torch.zeros(self.cout)).contiguous().to(ctx.device)


class HTDemucs:
    def __init__(self, cfg: HTDemucsConfig, state_dict: Dict[str, torch.Tensor], ctx: Optional[Context] = None):
        self.cfg = cfg
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        if cfg.kernel_size != 2 * cfg.stride or cfg.audio_channels != 2:
            raise AlsepError("HTDemucs: kernel_size must be 2 * stride and the input stereo")
        if cfg.bottom_channels % cfg.t_heads:
            raise AlsepError("HTDemucs: bottom_channels must be divisible by the head count")
        sd = state_dict
        dev = self.ctx.device
        from .roformer import check_shapes
        check_shapes(sd, expected_shapes(cfg), "HTDemucs",
                     ((f"encoder.{cfg.depth}.", f"depth={cfg.depth}"), (f"crosstransformer.layers.{cfg.t_layers}.", f"t_layers={cfg.t_layers}"),
                      ("encoder.0.dconv.layers.%d." % cfg.dconv_depth, f"dconv_depth={cfg.dconv_depth}")))
        try:
            self._build(sd)
        except KeyError as e:
            raise AlsepError(f"state_dict is missing {e} for this HTDemucsConfig") from e
        self._plans: Dict[int, object] = {}
        self._pos: Dict[tuple, torch.Tensor] = {}
        self._ws: Optional[torch.Tensor] = None
        self.dtype = torch.float32
        _ = dev

    def on_stream(self, ctx: Context) -> "HTDemucs":
        """A view of this network that launches on another context (= another HIP stream of the same device): the weights are shared
        (read-only device tensors), the per-call caches (FFT plans, workspace, position tables) are the view's own."""
        import copy
        if ctx.device != self.ctx.device:
            raise AlsepError("HTDemucs.on_stream: the other context must be on the same device")
        v = copy.copy(self)
        v.ctx = ctx
        v._plans, v._pos, v._ws = {}, {}, None
        return v

    # -- parameters ---------------------------------------------------------------------------------------
    def _vec(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().float().contiguous().to(self.ctx.device)

    def _build(self, sd) -> None:
        cfg, ctx = self.cfg, self.ctx
        K = cfg.kernel_size

        def conv_freq(p):       # Conv2d [Cout, Cin, KH, KW] -> [KH][KW][Cin][Cout]
            return _Conv(ctx, sd[p + ".weight"].permute(2, 3, 1, 0), sd.get(p + ".bias"))

        def conv_time(p):       # Conv1d [Cout, Cin, K]: the kernel runs along H (= samples), W = 1
            return _Conv(ctx, sd[p + ".weight"].permute(2, 1, 0)[:, None], sd.get(p + ".bias"))

        def conv_along_w(p):    # Conv1d applied along T of [Fr, T, C] (the DConv of the frequency branch): kernel along W
            return _Conv(ctx, sd[p + ".weight"].permute(2, 1, 0)[None], sd.get(p + ".bias"))

        def linear(wt, bs):     # [out, in] -> 1x1
            return _Conv(ctx, wt.t()[None, None], bs)

        def tconv(p):           # ConvTranspose [Cin, Cout, K(,1)] -> 1x1 conv to K * Cout columns (k-major), bias kept for the fold
            w = sd[p + ".weight"]
            w = w[..., 0] if w.dim() == 4 else w
            cin, cout, k = w.shape
            return _Conv(ctx, w.permute(0, 2, 1).reshape(cin, k * cout)[None, None], None), self._vec(sd[p + ".bias"]), cout

        def dconv(p, freq):
            layers = []
            for d in range(cfg.dconv_depth):
                q = f"{p}.layers.{d}"
                mk = conv_along_w if freq else conv_time
                layers.append(dict(c1=mk(q + ".0"), g1=self._vec(sd[q + ".1.weight"]), b1=self._vec(sd[q + ".1.bias"]),
                                   c2=mk(q + ".3"), g2=self._vec(sd[q + ".4.weight"]), b2=self._vec(sd[q + ".4.bias"]),
                                   scale=self._vec(sd[q + ".6.scale"]), dil=2 ** d))
            return layers

        self.enc, self.tenc, self.dec, self.tdec = [], [], [], []
        for idx in range(cfg.depth):
            self.enc.append(dict(conv=conv_freq(f"encoder.{idx}.conv"), dconv=dconv(f"encoder.{idx}.dconv", True),
                                 rewrite=conv_freq(f"encoder.{idx}.rewrite")))
            self.tenc.append(dict(conv=conv_time(f"tencoder.{idx}.conv"), dconv=dconv(f"tencoder.{idx}.dconv", False),
                                  rewrite=conv_time(f"tencoder.{idx}.rewrite")))
            tr, bias, cout = tconv(f"decoder.{idx}.conv_tr")
            self.dec.append(dict(rewrite=conv_freq(f"decoder.{idx}.rewrite"), tr=tr, bias=bias, cout=cout))
            tr, bias, cout = tconv(f"tdecoder.{idx}.conv_tr")
            self.tdec.append(dict(rewrite=conv_time(f"tdecoder.{idx}.rewrite"), tr=tr, bias=bias, cout=cout))
        self.freq_emb = self._vec(sd["freq_emb.embedding.weight"] * cfg.emb_scale)          # ScaledEmbedding: weight * scale
        self.up = linear(sd["channel_upsampler.weight"][..., 0], sd["channel_upsampler.bias"])
        self.up_t = linear(sd["channel_upsampler_t.weight"][..., 0], sd["channel_upsampler_t.bias"])
        self.down = linear(sd["channel_downsampler.weight"][..., 0], sd["channel_downsampler.bias"])
        self.down_t = linear(sd["channel_downsampler_t.weight"][..., 0], sd["channel_downsampler_t.bias"])
        p = "crosstransformer"
        self.norm_in = (self._vec(sd[p + ".norm_in.weight"]), self._vec(sd[p + ".norm_in.bias"]))
        self.norm_in_t = (self._vec(sd[p + ".norm_in_t.weight"]), self._vec(sd[p + ".norm_in_t.bias"]))
        cd = cfg.bottom_channels

        def tlayer(q, cross):
            att = q + (".cross_attn" if cross else ".self_attn")
            wi, bi = sd[att + ".in_proj_weight"], sd[att + ".in_proj_bias"]
            d = dict(cross=cross, out=linear(sd[att + ".out_proj.weight"], sd[att + ".out_proj.bias"]),
                     l1=linear(sd[q + ".linear1.weight"], sd[q + ".linear1.bias"]),
                     l2=linear(sd[q + ".linear2.weight"], sd[q + ".linear2.bias"]),
                     g1=self._vec(sd[q + ".gamma_1.scale"]), g2=self._vec(sd[q + ".gamma_2.scale"]),
                     norm_out=(self._vec(sd[q + ".norm_out.weight"]), self._vec(sd[q + ".norm_out.bias"])))
            for n in ("norm1", "norm2") + (("norm3",) if cross else ()):
                d[n] = (self._vec(sd[f"{q}.{n}.weight"]), self._vec(sd[f"{q}.{n}.bias"]))
            if cross:
                d["wq"] = linear(wi[:cd], bi[:cd])
                d["wkv"] = linear(wi[cd:], bi[cd:])
            else:
                d["wqkv"] = linear(wi, bi)
            return d
        self.layers = [tlayer(f"{p}.layers.{i}", i % 2 == 1) for i in range(cfg.t_layers)]
        self.layers_t = [tlayer(f"{p}.layers_t.{i}", i % 2 == 1) for i in range(cfg.t_layers)]

    # -- thin wrappers over the C ABI -----------------------------------------------------------------------
    def _conv(self, x: torch.Tensor, H: int, W: int, cv: _Conv, stride=(1, 1), pad=(0, 0), dil=(1, 1), act=ACT_NONE,
              out: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, int, int]:
        """x holds [H, W, cin] (possibly with zero rows beyond H*W); -> (y [Ho*Wo (+ spare), cout], Ho, Wo)"""
        ctx = self.ctx
        Ho = (H + 2 * pad[0] - dil[0] * (cv.kh - 1) - 1) // stride[0] + 1
        Wo = (W + 2 * pad[1] - dil[1] * (cv.kw - 1) - 1) // stride[1] + 1
        y = out if out is not None else ctx.empty((Ho * Wo, cv.cout))
        ctx.check(ctx.lib.alsep_nn_conv2d(ctx.handle, _lib.ptr(x), _lib.ptr(cv.w), _lib.ptr(cv.scale), _lib.ptr(cv.shift), _lib.ptr(y), 1,
                                          H, W, cv.cin, cv.cout, cv.kh, cv.kw, stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], act,
                                          cv.cout, 0), "alsep_nn_conv2d")
        return y, Ho, Wo

    def _workspace(self, G: int, per_group: int) -> torch.Tensor:
        need = int(self.ctx.lib.alsep_nn_stats_workspace_bytes(G, per_group))
        if self._ws is None or self._ws.numel() < need:
            self._ws = self.ctx.empty((max(need, 1 << 16),), torch.uint8)
        return self._ws

    def _norm(self, x: torch.Tensor, G: int, R: int, Cn: int, gamma, beta, act=ACT_NONE, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        ctx = self.ctx
        co = Cn // 2 if act == ACT_GLU else Cn
        y = out if out is not None else ctx.empty((G * R, co))
        ctx.check(ctx.lib.alsep_nn_norm(ctx.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), _lib.ptr(beta), G, R, Cn, 1e-5, act,
                                        _lib.ptr(self._workspace(G, R * Cn))), "alsep_nn_norm")
        return y

    def _act(self, x: torch.Tensor, rows: int, Cn: int, act: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        ctx = self.ctx
        y = out if out is not None else ctx.empty((rows, Cn // 2 if act == ACT_GLU else Cn))
        ctx.check(ctx.lib.alsep_nn_act(ctx.handle, _lib.ptr(x), _lib.ptr(y), rows, Cn, act), "alsep_nn_act")
        return y

    def _scale_add(self, a: torch.Tensor, b: torch.Tensor, scale, rows: int, Cn: int) -> torch.Tensor:
        ctx = self.ctx
        y = ctx.empty((rows, Cn))
        ctx.check(ctx.lib.alsep_nn_scale_add(ctx.handle, _lib.ptr(a), _lib.ptr(b), _lib.ptr(scale) if scale is not None else None,
                                             _lib.ptr(y), rows, Cn), "alsep_nn_scale_add")
        return y

    def _meanstd(self, x: torch.Tensor, n: int) -> torch.Tensor:
        ctx = self.ctx
        stats = ctx.empty((2,))
        ctx.check(ctx.lib.alsep_nn_meanstd(ctx.handle, _lib.ptr(x), 1, n, _lib.ptr(stats), _lib.ptr(self._workspace(1, n))), "alsep_nn_meanstd")
        return stats

    def _affine(self, x: torch.Tensor, stats: torch.Tensor, n: int, inverse: bool, eps: float = 1e-5) -> torch.Tensor:
        ctx = self.ctx
        y = torch.empty_like(x)
        ctx.check(ctx.lib.alsep_nn_affine_stats(ctx.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(stats), 1, n, eps, 1 if inverse else 0),
                  "alsep_nn_affine_stats")
        return y

    # -- layers ---------------------------------------------------------------------------------------------
    def _dconv(self, y: torch.Tensor, H: int, W: int, Cn: int, layers, freq: bool) -> torch.Tensor:
        """demucs DConv: residual branches on [H, W, C]; the 1-D convolution runs along T (W for the frequency branch, H for the
        time branch); GroupNorm(1) normalises over (C, T) of one sequence: one (b, f) row of the spectrogram, or the whole waveform"""
        rows = H * W
        G, R = (H, W) if freq else (1, H)
        for L in layers:
            d = L["dil"]
            h, _, _ = self._conv(y, H, W, L["c1"], pad=(0, d) if freq else (d, 0), dil=(1, d) if freq else (d, 1))
            h = self._norm(h, G, R, L["c1"].cout, L["g1"], L["b1"], ACT_GELU)
            h, _, _ = self._conv(h, H, W, L["c2"])
            h = self._norm(h, G, R, L["c2"].cout, L["g2"], L["b2"], ACT_GLU)
            y = self._scale_add(y, h, L["scale"], rows, Cn)
        return y

    def _enc_layer(self, x: torch.Tensor, H: int, W: int, P, freq: bool, out_rows: Optional[int] = None):
        """HEncLayer.forward; x [H, W, Cin] -> (z [Ho, W, C], Ho).  ``out_rows`` > Ho*W: z is allocated with zeroed spare rows (the
        right zero padding of the NEXT time-branch layer, F.pad(x, (0, stride - le % stride)))"""
        cfg = self.cfg
        cv = P["conv"]
        y, Ho, Wo = self._conv(x, H, W, cv, stride=(cfg.stride, 1), pad=(cfg.kernel_size // 4, 0), act=ACT_GELU)
        y = self._dconv(y, Ho, Wo, cv.cout, P["dconv"], freq)
        z, _, _ = self._conv(y, Ho, Wo, P["rewrite"])
        rows = Ho * Wo
        out = self.ctx.zeros((out_rows, cv.cout)) if out_rows and out_rows > rows else None
        return self._act(z, rows, 2 * cv.cout, ACT_GLU, out=out), Ho

    def _dec_layer(self, x: torch.Tensor, skip: torch.Tensor, H: int, W: int, P, freq: bool, length: int, last: bool):
        """HDecLayer.forward; x, skip [H, W, C] -> z [length, W, Cout]"""
        ctx, cfg = self.ctx, self.cfg
        cin = P["rewrite"].cin
        rows = H * W
        x = self._scale_add(x, skip, None, rows, cin)
        r, _, _ = self._conv(x, H, W, P["rewrite"], pad=(cfg.context, cfg.context if freq else 0))
        y = self._act(r, rows, 2 * cin, ACT_GLU)
        g, _, _ = self._conv(y, H, W, P["tr"])
        z = ctx.empty((length * W, P["cout"]))
        ctx.check(ctx.lib.alsep_nn_tconv_fold(ctx.handle, _lib.ptr(g), _lib.ptr(P["bias"]), _lib.ptr(z), 1, H, W, P["cout"], cfg.stride,
                                              cfg.kernel_size // 4, length, ACT_NONE if last else ACT_GELU), "alsep_nn_tconv_fold")
        return z

    def _attention(self, q: torch.Tensor, q_off: int, q_ld: int, kv: torch.Tensor, k_off: int, v_off: int, kv_ld: int, Nq: int, Nk: int) -> torch.Tensor:
        """softmax(Q K^T / sqrt(dh)) V per head; Q / K / V are column blocks of the packed projections (row strides q_ld / kv_ld)"""
        ctx, cfg = self.ctx, self.cfg
        Cd, Hh = cfg.bottom_channels, cfg.t_heads
        dh = Cd // Hh
        Np = -(-Nk // 4) * 4                                     # score rows padded to 16 bytes: `P V` then runs on the tiled GEMM
        scores = ctx.empty((Hh, Nq, Np))
        arr = C.c_int64 * 4
        ctx.check(ctx.lib.alsep_nn_bgemm(ctx.handle, C.c_void_p(q.data_ptr() + 4 * q_off), C.c_void_p(kv.data_ptr() + 4 * k_off), _lib.ptr(scores),
                                         1, Hh, Nq, Nk, dh, arr(0, dh, q_ld, 1), arr(0, dh, kv_ld, 1), arr(0, Nq * Np, Np, 1),
                                         1.0 / math.sqrt(dh)), "alsep_nn_bgemm")
        ctx.check(ctx.lib.alsep_nn_softmax_rows_ld(ctx.handle, _lib.ptr(scores), Hh * Nq, Nk, Np), "alsep_nn_softmax_rows_ld")
        out = ctx.empty((Nq, Cd))
        ctx.check(ctx.lib.alsep_nn_bgemm(ctx.handle, _lib.ptr(scores), C.c_void_p(kv.data_ptr() + 4 * v_off), _lib.ptr(out), 1, Hh, Nq, dh, Nk,
                                         arr(0, Nq * Np, Np, 1), arr(0, dh, 1, kv_ld), arr(0, dh, Cd, 1), 1.0), "alsep_nn_bgemm")
        return out

    def _tlayer(self, x: torch.Tensor, N: int, other: Optional[torch.Tensor], No: int, P) -> torch.Tensor:
        """MyTransformerEncoderLayer / CrossTransformerEncoderLayer (norm_first, LayerScale, norm_out) on tokens [N, C]"""
        Cd = self.cfg.bottom_channels
        if not P["cross"]:
            h = self._norm(x, N, 1, Cd, *P["norm1"])
            qkv, _, _ = self._conv(h, N, 1, P["wqkv"])
            a = self._attention(qkv, 0, 3 * Cd, qkv, Cd, 2 * Cd, 3 * Cd, N, N)
        else:
            q, _, _ = self._conv(self._norm(x, N, 1, Cd, *P["norm1"]), N, 1, P["wq"])
            kv, _, _ = self._conv(self._norm(other, No, 1, Cd, *P["norm2"]), No, 1, P["wkv"])
            a = self._attention(q, 0, Cd, kv, 0, Cd, 2 * Cd, N, No)
        a, _, _ = self._conv(a, N, 1, P["out"])
        x = self._scale_add(x, a, P["g1"], N, Cd)
        h = self._norm(x, N, 1, Cd, *(P["norm3"] if P["cross"] else P["norm2"]))
        f, _, _ = self._conv(h, N, 1, P["l1"], act=ACT_GELU)
        f, _, _ = self._conv(f, N, 1, P["l2"])
        x = self._scale_add(x, f, P["g2"], N, Cd)
        return self._norm(x, 1, N, Cd, *P["norm_out"])           # MyGroupNorm(1 group): over all tokens and channels

    def _pos_tables(self, Fr: int, T1: int, T2: int):
        """sinusoidal position tables (demucs.transformer.create_2d_sin_embedding / create_sin_embedding), token order (fr, t1)"""
        key = (Fr, T1, T2)
        if key not in self._pos:
            cfg = self.cfg
            Cd, mp = cfg.bottom_channels, cfg.t_max_period
            dm = Cd // 2
            div = torch.exp(torch.arange(0.0, dm, 2) * -(math.log(mp) / dm))
            pe = torch.zeros(Cd, Fr, T1)
            pw, ph = torch.arange(0.0, T1).unsqueeze(1), torch.arange(0.0, Fr).unsqueeze(1)
            pe[0:dm:2] = torch.sin(pw * div).transpose(0, 1).unsqueeze(1).repeat(1, Fr, 1)
            pe[1:dm:2] = torch.cos(pw * div).transpose(0, 1).unsqueeze(1).repeat(1, Fr, 1)
            pe[dm::2] = torch.sin(ph * div).transpose(0, 1).unsqueeze(2).repeat(1, 1, T1)
            pe[dm + 1::2] = torch.cos(ph * div).transpose(0, 1).unsqueeze(2).repeat(1, 1, T1)
            e2d = pe.permute(1, 2, 0).reshape(Fr * T1, Cd).contiguous()
            half = Cd // 2
            phase = torch.arange(T2).view(-1, 1).float() / (mp ** (torch.arange(half).view(1, -1).float() / (half - 1)))
            e1d = torch.cat([torch.cos(phase), torch.sin(phase)], dim=-1).contiguous()
            self._pos[key] = (e2d.to(self.ctx.device), e1d.to(self.ctx.device))
        return self._pos[key]

    def _plan(self, dim_t: int):
        from .mdx import StftPlan
        if dim_t not in self._plans:
            self._plans[dim_t] = StftPlan(self.ctx, self.cfg.nfft, self.cfg.hop, self.cfg.nfft // 2, dim_t)
        return self._plans[dim_t]

    # -- forward --------------------------------------------------------------------------------------------
    def forward(self, mix: torch.Tensor) -> torch.Tensor:
        """HTDemucs.forward (eval): mix [2, L] float32 on the device, L <= segment_samples -> [S, 2, L]"""
        ctx, cfg = self.ctx, self.cfg
        lib, h = ctx.lib, ctx.handle
        if mix.dim() != 2 or mix.shape[0] != 2 or mix.dtype != torch.float32:
            raise AlsepError("HTDemucs.forward expects a float32 [2, L] tensor")
        length_pre_pad = None
        if mix.shape[-1] < cfg.segment_samples:                # HTDemucs pads short inputs to its training length on the right
            length_pre_pad = mix.shape[-1]
            padded = ctx.zeros((2, cfg.segment_samples))
            padded[:, :length_pre_pad] = mix
            mix = padded
        mix = mix.contiguous()
        L = mix.shape[-1]
        hl, nfft, S = cfg.hop, cfg.nfft, cfg.S
        le = -(-L // hl)
        pad = hl // 2 * 3
        Lp = le * hl + 2 * pad
        xp = ctx.empty((2, Lp))
        ctx.check(lib.alsep_nn_reflect_pad(h, _lib.ptr(mix), _lib.ptr(xp), 2, L, pad, pad + le * hl - L), "alsep_nn_reflect_pad")
        Tt, Fq = le + 4, nfft // 2
        plan = self._plan(Tt)
        spec = plan.stft_strided(xp, Lp, 2 * Lp, 1, torch.float32, _lib.LAYOUT_REF)            # [1, 4, Fq, Tt]
        x = ctx.empty((Fq * le, 4))
        ctx.check(lib.alsep_demucs_spec_in(h, _lib.ptr(spec), _lib.ptr(x), 1, Fq, Tt, le, 2, 1.0 / math.sqrt(nfft)), "alsep_demucs_spec_in")
        stats = self._meanstd(x, Fq * le * 4)
        x = self._affine(x, stats, Fq * le * 4, inverse=False)
        xt = ctx.empty((L, 2))
        ctx.check(lib.alsep_nn_swap_last2(h, _lib.ptr(mix), _lib.ptr(xt), 1, 2, L), "alsep_nn_swap_last2")
        stats_t = self._meanstd(xt, 2 * L)
        xt = self._affine(xt, stats_t, 2 * L, inverse=False)

        saved, saved_t, lengths_t = [], [], []
        Fr, Lt = Fq, L
        St = cfg.stride
        for idx in range(cfg.depth):
            lengths_t.append(Lt)
            Lin = -(-Lt // St) * St                             # this layer sees its input zero-padded to a multiple of the stride
            Lt_next = Lin // St
            spare = -(-Lt_next // St) * St if idx + 1 < cfg.depth else Lt_next
            if idx == 0 and Lin != Lt:
                buf = ctx.zeros((Lin, 2))
                buf[:Lt] = xt
                xt = buf
            xt, Lt = self._enc_layer(xt, Lin, 1, self.tenc[idx], freq=False, out_rows=spare)
            saved_t.append((xt, Lt))
            x, Fr = self._enc_layer(x, Fr, le, self.enc[idx], freq=True)
            if idx == 0:
                c0 = self.enc[0]["conv"].cout
                ctx.check(lib.alsep_nn_add_bcast(h, _lib.ptr(x), _lib.ptr(self.freq_emb), cfg.freq_emb, Fr * le * c0, le * c0, Fr, c0),
                          "alsep_nn_add_bcast")
            saved.append((x, Fr))
        # cross-transformer on bottom_channels
        Cd = cfg.bottom_channels
        N1, N2 = Fr * le, Lt
        x, _, _ = self._conv(x, N1, 1, self.up)
        xt, _, _ = self._conv(xt, N2, 1, self.up_t)
        e2d, e1d = self._pos_tables(Fr, le, N2)
        x = self._norm(x, N1, 1, Cd, *self.norm_in)
        ctx.check(lib.alsep_nn_add_bcast(h, _lib.ptr(x), _lib.ptr(e2d), cfg.t_weight_pos_embed, N1 * Cd, Cd, N1, Cd), "alsep_nn_add_bcast")
        xt = self._norm(xt, N2, 1, Cd, *self.norm_in_t)
        ctx.check(lib.alsep_nn_add_bcast(h, _lib.ptr(xt), _lib.ptr(e1d), cfg.t_weight_pos_embed, N2 * Cd, Cd, N2, Cd), "alsep_nn_add_bcast")
        for idx in range(cfg.t_layers):
            if idx % 2 == 0:
                x = self._tlayer(x, N1, None, 0, self.layers[idx])
                xt = self._tlayer(xt, N2, None, 0, self.layers_t[idx])
            else:
                old_x = x
                x = self._tlayer(x, N1, xt, N2, self.layers[idx])
                xt = self._tlayer(xt, N2, old_x, N1, self.layers_t[idx])
        x, _, _ = self._conv(x, N1, 1, self.down)
        xt, _, _ = self._conv(xt, N2, 1, self.down_t)
        # decoders
        for idx in range(cfg.depth):
            last = idx == cfg.depth - 1
            skip, Fs = saved.pop(-1)
            x = self._dec_layer(x, skip, Fs, le, self.dec[idx], True, Fs * cfg.stride, last)
            Fr = Fs * cfg.stride
            skip_t, Ls = saved_t.pop(-1)
            xt = self._dec_layer(xt, skip_t, Ls, 1, self.tdec[idx], False, lengths_t.pop(-1), last)
        # outputs: mask (complex-as-channels) + iSTFT, plus the time branch
        spec_out = ctx.empty((S, 4, Fq, Tt))
        ctx.check(lib.alsep_demucs_spec_out(h, _lib.ptr(x), _lib.ptr(stats), _lib.ptr(spec_out), 1, S, Fq, Tt, le, 2, math.sqrt(nfft)),
                  "alsep_demucs_spec_out")
        xs = ctx.empty((S, 2, L))
        plan.istft_strided(spec_out, _lib.LAYOUT_REF, xs, L, 2 * L, pad, pad + L, (S - 1) * 2 * L + L)
        out = ctx.empty((S, 2, L))
        ctx.check(lib.alsep_demucs_mix_out(h, _lib.ptr(xt), _lib.ptr(stats_t), _lib.ptr(xs), _lib.ptr(out), 1, S, L), "alsep_demucs_mix_out")
        return out[..., :length_pre_pad] if length_pre_pad else out

    __call__ = forward


def shift_offsets(shifts: int, max_shift: int, seed: int = 0) -> List[int]:
    g = torch.Generator().manual_seed(seed)
    return [int(torch.randint(0, max_shift + 1, (1,), generator=g)) for _ in range(shifts)]


class DemucsRunner:
    """``demucs.apply.apply_model(model, mix, shifts, split=True, overlap)`` inside DemucsSeparator's whole-track normalisation
    (audio_separator defaults: shifts 2, overlap 0.25, segments of the model's training length), on the device."""

    def __init__(self, net: HTDemucs, shifts: int = 2, overlap: float = 0.25, seed: int = 0, sharded: bool = False, group=None,
                 lanes: Optional[int] = None, graphs: Optional[bool] = None, contraction: str = "exact"):
        """``lanes``: (shift, segment) units in flight at once, each on a HIP stream of its own (default: 4 on a GPU, 1 elsewhere).  One
        segment of htdemucs_6s is ~450 launches of mostly small kernels (grids of 42-170 workgroups on 256 CUs): units are independent,
        so running a few side by side fills the chip; the weighted sums are kept per lane and added at the end."""
        self.net, self.ctx = net, net.ctx
        # ``contraction="split"`` (opt-in): the network's float32 convolutions / GEMMs run as split-half products on the f16 matrix pipe
        # (csrc/nn_f32s.h: float32 in and out, 2^-22 per product) for the duration of a track, on ONE lane; a track during which an operand
        # left the half range is run again on the exact f32 MFMA kernels.  Measured (10 min, one GPU): exact with four lanes 1.66 s; split
        # with four lanes 1.38 s but WRONG (the FFT launches of the other lanes are corrupted beside the f16 MFMA waves, 3e-4 at peak
        # 0.014 and not reproducible); hence one lane, and the default stays exact.
        if contraction not in ("split", "exact"):
            raise AlsepError("contraction must be 'split' or 'exact'")
        self.contraction = contraction
        if contraction == "split":
            lanes = 1          # the split kernels issue f16 MFMA: FFT launches must not share the GPU with them (roformer.RoformerRunner)
        self.shifts, self.overlap, self.seed = shifts, overlap, seed
        self.sharded, self.group = sharded, group
        if lanes is None:
            lanes = int(os.environ.get("ALSEP_DEMUCS_LANES", "4")) if self.ctx.device.type == "cuda" else 1
        self.lanes = max(1, int(lanes)) if self.ctx.device.type == "cuda" else 1
        # ``graphs`` (default off; ``ALSEP_DEMUCS_GRAPH=1``): every lane captures ONE segment forward into a HIP graph on its own stream and
        # replays it per unit, as the Roformer / MDX23C runner does.  Measured here it LOSES: htdemucs_6s, 10 min, 4 lanes: 1.70 s with plain
        # launches, 2.05 s with graph replays (a segment is ~450 small kernels; the replay's per-node cost exceeds what the host saves).
        if graphs is None:
            graphs = os.environ.get("ALSEP_DEMUCS_GRAPH", "0") != "0"
        self.graphs = bool(graphs) and self.ctx.device.type == "cuda"
        self._lane_nets: List[tuple] = []                      # [(HTDemucs view, torch stream)], built on first use
        self._graphs: Dict[int, list] = {}                     # lane index -> [(graph, static input, static output)] x graphs_per_lane
        self._graph_turn: Dict[int, int] = {}
        self.graphs_per_lane = max(1, int(os.environ.get("ALSEP_RUNNER_GRAPHS_PER_LANE", "2")))

    def _lanes(self):
        if not self._lane_nets:
            first = 0 if self.graphs else 1                    # a capture needs a non-default stream: with graphs lane 0 gets its own too
            if not self.graphs:
                self._lane_nets = [(self.net, None)]
            for _ in range(first, self.lanes):
                st = torch.cuda.Stream(device=self.ctx.device)
                self._lane_nets.append((self.net.on_stream(Context(self.ctx.device, stream=st.cuda_stream)), st))
        return self._lane_nets

    def _forward_unit(self, k: int, lane_net, st, chunk: torch.Tensor) -> torch.Tensor:
        """the network on one [2, seg] segment on lane k's stream: eagerly, or -- from the lane's second unit on -- as a graph replay"""
        if not self.graphs:
            return lane_net.forward(chunk)
        slots = self._graphs.setdefault(k, [])                  # two captures per lane, replayed in turn (roformer.RoformerRunner._forward_chunk)
        if len(slots) < self.graphs_per_lane:
            y = lane_net.forward(chunk)                         # eager first: plans, tables, workspaces outside a capture
            st.synchronize()
            static_in = chunk.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                static_out = lane_net.forward(static_in)
            slots.append((g, static_in, static_out))
            return y
        turn = self._graph_turn.get(k, 0)
        self._graph_turn[k] = (turn + 1) % len(slots)
        g, static_in, static_out = slots[turn]
        static_in.copy_(chunk)
        g.replay()
        return static_out

    def units(self, length: int):
        """[(root offset of the view, view length, chunk offset in the view, chunk length, out offset)] over all shifts"""
        cfg = self.net.cfg
        seg = cfg.segment_samples
        stride = int((1 - self.overlap) * seg)
        max_shift = int(0.5 * cfg.samplerate) if self.shifts else 0
        passes = shift_offsets(self.shifts, max_shift, self.seed) if self.shifts else [0]
        out = []
        for p, offset in enumerate(passes):
            view_len = length + max_shift - offset if self.shifts else length
            for off in range(0, view_len, stride):
                out.append((p, offset, view_len, off, min(view_len - off, seg)))
        return out, max_shift, len(passes)

    def separate(self, mix: torch.Tensor) -> Dict[str, torch.Tensor]:
        """mix [2, L] on the device -> {source name: [2, L]} (sources in the model's order)"""
        if self.contraction != "split":
            return self._separate(mix)
        ctxs = [self.ctx] + [ln.ctx for ln, _ in self._lanes() if ln.ctx is not self.ctx]
        try:
            for c in ctxs:
                c.set_nn_contraction(True)
            out = self._separate(mix)
            exceeded = [c.nn_range_exceeded() for c in ctxs]
        finally:
            for c in ctxs:
                c.set_nn_contraction(False)
        if any(exceeded):
            import logging
            logging.getLogger(__name__).warning("DemucsRunner: an operand left the half range (|x| > 65504) during this track -- running it "
                                                "again on the exact float32 kernels")
            self._graphs.clear()                               # captured with the split kernels
            out = self._separate(mix)
        return out

    def _separate(self, mix: torch.Tensor) -> Dict[str, torch.Tensor]:
        ctx, net = self.ctx, self.net
        cfg = net.cfg
        lib, h = ctx.lib, ctx.handle
        mix = mix.contiguous().float()
        L = mix.shape[-1]
        S, seg = cfg.S, cfg.segment_samples
        # ref = mix.mean(0); mix = (mix - ref.mean()) / ref.std()
        ref, right = mix[0].clone(), mix[1].clone()             # own (16-byte aligned) buffers: alsep_axpby moves float4
        ctx.check(lib.alsep_axpby(h, 0.5, _lib.ptr(right), 0.5, _lib.ptr(ref), L), "alsep_axpby")
        stats = net._meanstd(ref, L)
        norm = ctx.empty((2, L))
        stats2 = torch.cat([stats, stats])                      # the same (mean, std) for both channels; kept alive across the call
        ctx.check(lib.alsep_nn_affine_stats(h, _lib.ptr(mix), _lib.ptr(norm), _lib.ptr(stats2), 2, L, 0.0, 0), "alsep_nn_affine_stats")
        units, max_shift, n_pass = self.units(L)
        root = ctx.zeros((2, L + 2 * max_shift))
        root[:, max_shift:max_shift + L] = norm
        total = root.shape[-1]
        weight = torch.cat([torch.arange(1, seg // 2 + 1), torch.arange(seg - seg // 2, 0, -1)]).float()
        weight = (weight / weight.max()).to(ctx.device)
        rank, world = 0, 1
        if self.sharded:
            import torch.distributed as tdist
            rank, world = tdist.get_rank(self.group), tdist.get_world_size(self.group)
        from . import dist as adist
        lo, hi = adist.window_range(len(units), world, rank)
        lanes = self._lanes()[: max(1, min(self.lanes, hi - lo))]
        # one weighted sum per (lane, shift pass), view coordinates; lane 0 runs on this context's stream, the others on their own
        accs = [[ctx.zeros((S * 2, L + max_shift)) for _ in range(n_pass)] for _ in lanes]
        main = torch.cuda.current_stream(ctx.device) if (len(lanes) > 1 or self.graphs) else None
        for _, st in lanes:
            if st is not None:
                st.wait_stream(main)                             # root, weight and the zeroed sums are ready

        def run_unit(k, lane_net, st, lane_acc, unit):
            p, offset, view_len, off, cl = unit
            lctx = lane_net.ctx
            delta = seg - cl
            start = offset + off - delta // 2
            end = start + seg
            cs, ce = max(0, start), min(total, end)
            chunk = lctx.zeros((2, seg))
            chunk[:, cs - start: cs - start + (ce - cs)] = root[:, cs:ce]
            y = self._forward_unit(k, lane_net, st, chunk)                           # [S, 2, seg]
            src = C.c_void_p(y.data_ptr() + 4 * (delta // 2))
            dst = C.c_void_p(lane_acc[p].data_ptr() + 4 * off)
            lctx.check(lctx.lib.alsep_nn_vec_fma(lctx.handle, dst, src, _lib.ptr(weight), S * 2, cl, L + max_shift, seg), "alsep_nn_vec_fma")

        for i, unit in enumerate(units[lo:hi]):
            k = i % len(lanes)
            lane_net, st = lanes[k]
            if st is None:
                run_unit(k, lane_net, st, accs[0], unit)
            else:
                with torch.cuda.stream(st):                      # torch's allocator ties the lane's temporaries to its stream
                    run_unit(k, lane_net, st, accs[k], unit)
        acc = accs[0]
        for k, (_, st) in enumerate(lanes):
            if st is None:
                continue
            main.wait_stream(st)
            if k == 0:
                continue
            for p in range(n_pass):
                ctx.check(lib.alsep_axpby(h, 1.0, _lib.ptr(accs[k][p]), 1.0, _lib.ptr(acc[p]), acc[p].numel()), "alsep_axpby")
        # per pass: divide by the summed weights of that pass, cut the view back to the track, average the passes
        stride = int((1 - self.overlap) * seg)
        passes = shift_offsets(self.shifts, max_shift, self.seed) if self.shifts else [0]
        width = L + max_shift
        sws = []
        for p, offset in enumerate(passes):
            view_len = L + max_shift - offset if self.shifts else L
            sw = torch.zeros(width)
            for off in range(0, view_len, stride):
                cl = min(view_len - off, seg)
                sw[off:off + cl] += _tri(seg)[:cl]
            sws.append(sw.to(ctx.device))
        if self.sharded and world > 1:
            # a rank's units are a contiguous run, so in every pass it owns the span from its first unit's offset to the next
            # rank's (the last one: to the end).  Its units reach at most seg - stride samples beyond that span: those seam sums travel in
            # one small all-gather and are added by the span's owner; every rank divides ITS spans by the summed weights, and ONE
            # all-gather of the finished spans (both passes side by side) gives every rank the whole passes -- no full-length all-reduce.
            seam = max(seg - stride, 1)
            bounds = [adist.window_range(len(units), world, q) for q in range(world)]
            ranges, tails = [], []
            for p in range(n_pass):
                firsts = []                                                     # first offset of rank q's units in pass p (None: none there)
                for q_lo, q_hi in bounds:
                    offs = [u[3] for u in units[q_lo:q_hi] if u[0] == p]
                    firsts.append(offs[0] if offs else None)
                owners = [q for q in range(world) if firsts[q] is not None]
                rg = [None] * world
                for i, q in enumerate(owners):
                    rg[q] = (0 if i == 0 else firsts[q], firsts[owners[i + 1]] if i + 1 < len(owners) else width)
                for q in range(world):                                          # a rank without units in this pass: an empty span, placed
                    if rg[q] is None:                                           # so that the spans stay ascending and cover [0, width)
                        at = next((rg[r][0] for r in range(q + 1, world) if rg[r] is not None and firsts[r] is not None), width)
                        rg[q] = (at, at)
                ranges.append(rg)
                own_lo, own_hi = rg[rank]
                tail = ctx.zeros((S * 2, seam))
                n_tail = max(0, min(seam, width - own_hi)) if own_hi > own_lo else 0
                if n_tail:
                    tail[:, :n_tail] = acc[p][:, own_hi: own_hi + n_tail]
                tails.append(tail)
            got = adist.all_gather_fixed(torch.stack(tails), self.group)        # [world, n_pass, 2 S, seam]
            pieces = []
            for p in range(n_pass):
                own_lo, own_hi = ranges[p][rank]
                own = acc[p][:, own_lo:own_hi].contiguous()
                for q in range(rank):                                           # earlier owners whose units reach into this span
                    q_lo, q_hi = ranges[p][q]
                    if q_hi <= q_lo:
                        continue
                    lo_, hi_ = max(q_hi, own_lo), min(q_hi + seam, own_hi)
                    if hi_ > lo_:
                        own[:, lo_ - own_lo: hi_ - own_lo] += got[q, p][:, lo_ - q_hi: hi_ - q_hi]
                if own_hi > own_lo:
                    swp = sws[p][own_lo:own_hi].contiguous()
                    ctx.check(lib.alsep_nn_vec_div(h, _lib.ptr(own), _lib.ptr(swp), S * 2, own_hi - own_lo), "alsep_nn_vec_div")
                pieces.append(own)
            acc = adist.all_gather_multi_ranges(pieces, ranges, [width] * n_pass, self.group)
        out = ctx.zeros((S * 2, L))
        for p, offset in enumerate(passes):
            view_len = L + max_shift - offset if self.shifts else L
            a = acc[p][:, :view_len].contiguous()
            if not (self.sharded and world > 1):
                sw = sws[p][:view_len].contiguous()
                ctx.check(lib.alsep_nn_vec_div(h, _lib.ptr(a), _lib.ptr(sw), S * 2, view_len), "alsep_nn_vec_div")
            cut = a[:, max_shift - offset: max_shift - offset + L].contiguous()
            ctx.check(lib.alsep_axpby(h, 1.0 / n_pass, _lib.ptr(cut), 1.0, _lib.ptr(out), S * 2 * L), "alsep_axpby")
        # sources * ref.std() + ref.mean()
        res = ctx.empty((S * 2, L))
        ctx.check(lib.alsep_nn_affine_stats(h, _lib.ptr(out), _lib.ptr(res), _lib.ptr(stats), 1, S * 2 * L, 0.0, 1), "alsep_nn_affine_stats")
        res = res.view(S, 2, L)
        return {name: res[i] for i, name in enumerate(cfg.sources)}


_TRI: Dict[int, torch.Tensor] = {}


def _tri(seg: int) -> torch.Tensor:
    if seg not in _TRI:
        w = torch.cat([torch.arange(1, seg // 2 + 1), torch.arange(seg - seg // 2, 0, -1)]).float()
        _TRI[seg] = w / w.max()
    return _TRI[seg]


# ---- synthetic weights (data only: random-init parameters with demucs' names and shapes; bench / tests, allow_synthetic=True) ----
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
