"""MDX23C (TFC-TDF v3) on the GPU -- ensemble slot 4 of the reference (``MDX23C-8KFFT-InstVoc_HQ.ckpt``,
modules/separator/stem_separator.py:383) and its drum-kit splitter (``MDX23C-DrumSep-aufr33-jarredou.ckpt``, :541: six outputs matched by
``(kick)`` ... ``(crash)``, :563-574).

The network code (``tfc_tdf_v3.py``) lives in the un-vendored ``audio-separator[gpu]>=0.32.0`` (setup.sh:96): PARITY UNPINNED -- restated from
the published design (oracle/mdx23c_oracle.py is the torch-CPU fp32 twin), with that code's parameter names so that a real ``state_dict``
loads as it is.  float32, channels-last ``[T, f, C]`` (frames x sub-band bins x channels): 3x3 / 1x1 / strided convolutions through
``alsep_nn_conv2d`` (exact-f32 MFMA), the TDF linears as strided batched GEMMs over the frequency axis (no transposes), InstanceNorm + GELU
fused (``alsep_nn_instnorm``), the 2x2 transposed convolution as a 1x1 conv + depth-to-space straight into the concatenation buffer;
STFT / iSTFT by the n_fft 8192 kernels of csrc/fft.hip.  Runner: the chunked inference shared with the Roformer models."""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import AlsepError, Context

ACT_NONE, ACT_GELU = 0, 3


@dataclass(frozen=True)
class MDX23CConfig:
    instruments: Tuple[str, ...] = ("vocals", "other")
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


def expected_shapes(cfg: MDX23CConfig) -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, hyper-parameter that fixes it) of every tensor an MDX23C with this configuration reads (tfc_tdf_v3 names)"""
    exp: Dict[str, Tuple[Tuple[int, ...], str]] = {}
    ch = f"num_channels={cfg.num_channels} / growth={cfg.growth}"

    def norm(p, c, hyper=ch):
        exp[p + ".weight"] = ((c,), hyper)
        exp[p + ".bias"] = ((c,), hyper)

    def tfc_tdf(p, in_c, c, f):
        fh = f"audio.dim_f={cfg.dim_f} / num_subbands={cfg.num_subbands} / bottleneck_factor={cfg.bottleneck_factor} (TDF width {f})"
        for i in range(cfg.num_blocks_per_scale):
            q = f"{p}.blocks.{i}"
            norm(q + ".tfc1.0", in_c)
            exp[q + ".tfc1.2.weight"] = ((c, in_c, 3, 3), ch)
            norm(q + ".tdf.0", c)
            exp[q + ".tdf.2.weight"] = ((f // cfg.bottleneck_factor, f), fh)
            norm(q + ".tdf.3", c)
            exp[q + ".tdf.5.weight"] = ((f, f // cfg.bottleneck_factor), fh)
            norm(q + ".tfc2.0", c)
            exp[q + ".tfc2.2.weight"] = ((c, c, 3, 3), ch)
            exp[q + ".shortcut.weight"] = ((c, in_c, 1, 1), ch)
            in_c = c

    c, gr, f = cfg.num_channels, cfg.growth, cfg.dim_f // cfg.num_subbands
    exp["first_conv.weight"] = ((c, cfg.dim_c, 1, 1), f"num_channels={cfg.num_channels} / num_subbands={cfg.num_subbands}")
    for i in range(cfg.num_scales):
        tfc_tdf(f"encoder_blocks.{i}.tfc_tdf", c, c, f)
        norm(f"encoder_blocks.{i}.downscale.conv.0", c)
        exp[f"encoder_blocks.{i}.downscale.conv.2.weight"] = ((c + gr, c, 2, 2), ch)
        f //= 2
        c += gr
    tfc_tdf("bottleneck_block", c, c, f)
    for i in range(cfg.num_scales):
        norm(f"decoder_blocks.{i}.upscale.conv.0", c)
        exp[f"decoder_blocks.{i}.upscale.conv.2.weight"] = ((c, c - gr, 2, 2), ch)
        f *= 2
        c -= gr
        tfc_tdf(f"decoder_blocks.{i}.tfc_tdf", 2 * c, c, f)
    exp["final_conv.0.weight"] = ((c, c + cfg.dim_c, 1, 1), ch)
    exp["final_conv.2.weight"] = ((cfg.num_stems * cfg.dim_c, c, 1, 1),
                                  f"training.instruments ({cfg.num_stems} stems) / num_subbands={cfg.num_subbands}")
    return exp


class _W:
    """[KH][KW][Cin][Cout] convolution weights (no bias anywhere in this network)"""

    def __init__(self, ctx: Context, w4: torch.Tensor, half: bool = False, what: str = ""):
        self.kh, self.kw, self.cin, self.cout = (int(v) for v in w4.shape)
        self.w = w4.detach().float().contiguous().to(ctx.device)
        self.scale = torch.ones(self.cout, device=ctx.device)
        self.shift = torch.zeros(self.cout, device=ctx.device)
        self.wh = None
        if half:                                            # [Cout][KH][KW][Cin] IEEE half: the image alsep_nn_conv2d_f16 reads
            if self.cin % 64 or self.cout % 4:
                raise AlsepError(f"MDX23C precision='f16': {what} has {self.cin} -> {self.cout} channels; the half-precision convolution needs "
                                 f"input channels in multiples of 64 (num_channels / growth of the .yaml)")
            self.wh = self.w.permute(3, 0, 1, 2).contiguous().to(torch.float16)


class MDX23C:
    def __init__(self, cfg: MDX23CConfig, state_dict: Dict[str, torch.Tensor], ctx: Optional[Context] = None, precision: str = "f32"):
        """``precision="f16"``: the half-precision mode (the reference runs this network under torch autocast, stem_separator.py:106
        ``use_autocast=True``).  The convolutions of the TFC-TDF blocks and of the down- / up-scaling layers -- 85 % of the arithmetic --
        read IEEE-half activations (written in that type by the InstanceNorm + GELU kernel in front of them; the shortcut branch's raw
        input by a conversion pass) and IEEE-half weights on v_mfma_f32_16x16x32_f16; their results, the residual stream, the statistics,
        the first / final 1x1 convolutions stay float32.  The TDF linears over the frequency axis run on the f16 GEMM as well (operands
        rounded to half, float32 result with the residual fused) wherever the level's frequency count is a multiple of 32 (every level of
        the published models).  Storage-mode oracle: oracle/mdx23c_oracle.forward(half=True)."""
        if precision not in ("f32", "f16"):
            raise AlsepError(f"MDX23C: precision {precision!r} (f32, f16)")
        self.half = precision == "f16"
        if tuple(cfg.scale) != (2, 2):
            raise AlsepError("MDX23C: only scale (2, 2) is implemented")
        if cfg.dim_f % cfg.num_subbands or (cfg.dim_f // cfg.num_subbands) % (2 ** cfg.num_scales * cfg.bottleneck_factor):
            raise AlsepError("MDX23C: dim_f / num_subbands must be divisible by 2**num_scales * bottleneck_factor")
        self.cfg = cfg
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        self.dtype = torch.float32
        sd, dev = state_dict, self.ctx.device
        from .roformer import check_shapes
        check_shapes(sd, expected_shapes(cfg), "MDX23C",
                     ((f"encoder_blocks.{cfg.num_scales}.", f"num_scales={cfg.num_scales}"),
                      (f"decoder_blocks.{cfg.num_scales}.", f"num_scales={cfg.num_scales}"),
                      ("encoder_blocks.0.tfc_tdf.blocks.%d." % cfg.num_blocks_per_scale, f"num_blocks_per_scale={cfg.num_blocks_per_scale}")))
        v = lambda k: sd[k].detach().float().contiguous().to(dev)

        def conv(k, half=None):           # Conv2d [Cout, Cin, KH, KW] over (T, F) -> H = T, W = F
            return _W(self.ctx, sd[k].permute(2, 3, 1, 0), self.half if half is None else half, k)

        def tconv(k):                     # ConvTranspose2d [Cin, Cout, 2, 2] -> 1x1 conv to (dy*2+dx)*Cout + co
            w = sd[k]
            cin, cout = w.shape[:2]
            return _W(self.ctx, w.permute(0, 2, 3, 1).reshape(cin, 4 * cout)[None, None], self.half, k), int(cout)

        def block(p):
            out = []
            for i in range(cfg.num_blocks_per_scale):
                q = f"{p}.blocks.{i}"
                lh = (lambda k: v(k).to(torch.float16)) if self.half else (lambda k: None)
                out.append(dict(short=conv(q + ".shortcut.weight"), n1=(v(q + ".tfc1.0.weight"), v(q + ".tfc1.0.bias")), c1=conv(q + ".tfc1.2.weight"),
                                l1h=lh(q + ".tdf.2.weight"), l2h=lh(q + ".tdf.5.weight"),
                                nt1=(v(q + ".tdf.0.weight"), v(q + ".tdf.0.bias")), l1=v(q + ".tdf.2.weight"),
                                nt2=(v(q + ".tdf.3.weight"), v(q + ".tdf.3.bias")), l2=v(q + ".tdf.5.weight"),
                                n2=(v(q + ".tfc2.0.weight"), v(q + ".tfc2.0.bias")), c2=conv(q + ".tfc2.2.weight")))
            return out
        try:
            self.first = conv("first_conv.weight", False)
            self.enc = [dict(blk=block(f"encoder_blocks.{i}.tfc_tdf"),
                             dn=(v(f"encoder_blocks.{i}.downscale.conv.0.weight"), v(f"encoder_blocks.{i}.downscale.conv.0.bias")),
                             dconv=conv(f"encoder_blocks.{i}.downscale.conv.2.weight")) for i in range(cfg.num_scales)]
            self.bott = block("bottleneck_block")
            self.dec = []
            for i in range(cfg.num_scales):
                tw, cout = tconv(f"decoder_blocks.{i}.upscale.conv.2.weight")
                self.dec.append(dict(un=(v(f"decoder_blocks.{i}.upscale.conv.0.weight"), v(f"decoder_blocks.{i}.upscale.conv.0.bias")), up=tw,
                                     cout=cout, blk=block(f"decoder_blocks.{i}.tfc_tdf")))
            self.final0, self.final2 = conv("final_conv.0.weight", False), conv("final_conv.2.weight", False)
        except KeyError as e:
            raise AlsepError(f"state_dict is missing {e} for this MDX23CConfig") from e
        self._plans: Dict[int, object] = {}
        self._ws: Optional[torch.Tensor] = None
        self._cws: Optional[torch.Tensor] = None

    # -- wrappers ---------------------------------------------------------------------------------------------
    def _conv(self, x, H, W, cv: _W, stride=(1, 1), pad=(0, 0), act=ACT_NONE, out=None, ctotal=None, coff=0):
        ctx = self.ctx
        Ho = (H + 2 * pad[0] - (cv.kh - 1) - 1) // stride[0] + 1
        Wo = (W + 2 * pad[1] - (cv.kw - 1) - 1) // stride[1] + 1
        ct = ctotal or cv.cout
        y = out if out is not None else ctx.empty((Ho * Wo, ct))
        ctx.check(ctx.lib.alsep_nn_conv2d(ctx.handle, _lib.ptr(x), _lib.ptr(cv.w), _lib.ptr(cv.scale), _lib.ptr(cv.shift), _lib.ptr(y), 1, H, W,
                                          cv.cin, cv.cout, cv.kh, cv.kw, stride[0], stride[1], pad[0], pad[1], 1, 1, act, ct, coff), "alsep_nn_conv2d")
        return y, Ho, Wo

    def _norm_act(self, x, P, Cn, gb, half=False):
        """InstanceNorm2d(affine) + GELU; ``half``: the result as IEEE half (what a half-precision convolution reads)"""
        ctx = self.ctx
        need = int(ctx.lib.alsep_nn_instnorm_workspace_bytes(P, Cn))
        if self._ws is None or self._ws.numel() < need:
            self._ws = ctx.empty((max(need, 1 << 16),), torch.uint8)
        if half:
            y = ctx.empty((P, Cn), torch.float16)
            ctx.check(ctx.lib.alsep_nn_instnorm_f16(ctx.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gb[0]), _lib.ptr(gb[1]), P, Cn, 1e-5, ACT_GELU,
                                                    _lib.ptr(self._ws)), "alsep_nn_instnorm_f16")
            return y
        y = ctx.empty((P, Cn))
        ctx.check(ctx.lib.alsep_nn_instnorm(ctx.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gb[0]), _lib.ptr(gb[1]), P, Cn, 1e-5, ACT_GELU,
                                            _lib.ptr(self._ws)), "alsep_nn_instnorm")
        return y

    def _conv_h(self, xh, H, W, cv: _W, stride=(1, 1), pad=(0, 0), res=None):
        """half-precision convolution of an IEEE-half activation; float32 result (+ ``res``, float32 [pixels, Cout])"""
        ctx = self.ctx
        Ho = (H + 2 * pad[0] - cv.kh) // stride[0] + 1
        Wo = (W + 2 * pad[1] - cv.kw) // stride[1] + 1
        y = ctx.empty((Ho * Wo, cv.cout))
        need = int(ctx.lib.alsep_nn_conv2d_f16_workspace_bytes(1, H, W, cv.cin, cv.cout, cv.kh, cv.kw, stride[0], stride[1], pad[0], pad[1]))
        if need < 0:
            raise AlsepError("MDX23C: alsep_nn_conv2d_f16_workspace_bytes rejected the layer's geometry")
        if need and (self._cws is None or self._cws.numel() < need):     # split-K partials of the deep levels' layers
            self._cws = ctx.empty((need,), torch.uint8)
        ctx.check(ctx.lib.alsep_nn_conv2d_f16(ctx.handle, _lib.ptr(xh), _lib.ptr(cv.wh), _lib.ptr(y), _lib.ptr(res) if res is not None else None,
                                              cv.cout, 1, H, W, cv.cin, cv.cout, cv.kh, cv.kw, stride[0], stride[1], pad[0], pad[1], cv.cout, 0,
                                              _lib.ptr(self._cws) if need else None, need), "alsep_nn_conv2d_f16")
        return y, Ho, Wo

    def _norm_act_t(self, x, T, Fw, Cn, gb):
        """InstanceNorm2d + GELU of x [T * Fw, Cn] with the IEEE-half result stored [T][Cn][Fw] (the TDF Linear's operand layout)"""
        ctx = self.ctx
        need = int(ctx.lib.alsep_nn_instnorm_workspace_bytes(T * Fw, Cn))
        if self._ws is None or self._ws.numel() < need:
            self._ws = ctx.empty((max(need, 1 << 16),), torch.uint8)
        y = ctx.empty((T, Cn, Fw), torch.float16)
        ctx.check(ctx.lib.alsep_nn_instnorm_f16_t(ctx.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gb[0]), _lib.ptr(gb[1]), T, Fw, Cn, 1e-5, ACT_GELU,
                                                  _lib.ptr(self._ws)), "alsep_nn_instnorm_f16_t")
        return y

    def _linear_f_h(self, xt, T, Fin, Cn, wh, res=None):
        """Linear over the frequency axis in half precision: y[t, f', c] = sum_f w[f', f] xt[t, c, f] (+ res[t, f', c]) as ONE batched f16
        GEMM -- per frame A = the shared weight matrix [Fo, Fin], "W" = the frame's [Cn, Fin] rows, float32 result [Fo, Cn]"""
        ctx = self.ctx
        Fo = int(wh.shape[0])
        y = ctx.empty((T * Fo, Cn))
        ctx.check(ctx.lib.alsep_nn_gemm_f16(ctx.handle, _lib.ptr(wh), Fin, 0, _lib.ptr(xt), Fin, Cn * Fin, _lib.ptr(y), 0, Cn, Fo * Cn, None, 0,
                                            _lib.ptr(res) if res is not None else None, Cn, Fo * Cn, T, Fo, Cn, Fin, 1.0, 0, None), "alsep_nn_gemm_f16")
        return y, Fo

    def _to_half(self, x):
        ctx = self.ctx
        y = ctx.empty(tuple(x.shape), torch.float16)
        ctx.check(ctx.lib.alsep_nn_to_f16(ctx.handle, _lib.ptr(x), _lib.ptr(y), x.numel()), "alsep_nn_to_f16")
        return y

    def _linear_f(self, x, T, Fin, Cn, w):
        """Linear over the frequency axis of x [T, Fin, C]: y[t, f', c] = sum_f w[f', f] x[t, f, c] (one GEMM per frame, strided)"""
        ctx = self.ctx
        Fo = int(w.shape[0])
        y = ctx.empty((T * Fo, Cn))
        arr = C.c_int64 * 4
        ctx.check(ctx.lib.alsep_nn_bgemm(ctx.handle, _lib.ptr(w), _lib.ptr(x), _lib.ptr(y), T, 1, Fo, Cn, Fin, arr(0, 0, Fin, 1),
                                         arr(Fin * Cn, 0, 1, Cn), arr(Fo * Cn, 0, Cn, 1), 1.0), "alsep_nn_bgemm")
        return y, Fo

    def _add(self, a, b, n):
        ctx = self.ctx
        y = torch.empty_like(a)
        ctx.check(ctx.lib.alsep_nn_scale_add(ctx.handle, _lib.ptr(a), _lib.ptr(b), None, _lib.ptr(y), n, 1), "alsep_nn_scale_add")
        return y

    def _block(self, x, T, Fw, blk):
        P = T * Fw
        for L in blk:
            cin, c = L["c1"].cin, L["c1"].cout
            if self.half:
                s, _, _ = self._conv_h(self._to_half(x), T, Fw, L["short"])
                x, _, _ = self._conv_h(self._norm_act(x, P, cin, L["n1"], True), T, Fw, L["c1"], pad=(1, 1))
                if Fw % 32 == 0 and (Fw // self.cfg.bottleneck_factor) % 8 == 0:   # TDF linears on the f16 GEMM (both K -- the frequency count and the bottleneck width -- in whole 8-groups)
                    t, Fh = self._linear_f_h(self._norm_act_t(x, T, Fw, c, L["nt1"]), T, Fw, c, L["l1h"])
                    x, _ = self._linear_f_h(self._norm_act_t(t, T, Fh, c, L["nt2"]), T, Fh, c, L["l2h"], res=x)       # x + tdf(x), fused
                else:
                    t, Fh = self._linear_f(self._norm_act(x, P, c, L["nt1"]), T, Fw, c, L["l1"])
                    t, _ = self._linear_f(self._norm_act(t, T * Fh, c, L["nt2"]), T, Fh, c, L["l2"])
                    x = self._add(x, t, P * c)
                x, _, _ = self._conv_h(self._norm_act(x, P, c, L["n2"], True), T, Fw, L["c2"], pad=(1, 1), res=s)      # + shortcut, fused
                continue
            s, _, _ = self._conv(x, T, Fw, L["short"])
            x, _, _ = self._conv(self._norm_act(x, P, cin, L["n1"]), T, Fw, L["c1"], pad=(1, 1))
            t, Fh = self._linear_f(self._norm_act(x, P, c, L["nt1"]), T, Fw, c, L["l1"])
            t, _ = self._linear_f(self._norm_act(t, T * Fh, c, L["nt2"]), T, Fh, c, L["l2"])
            x = self._add(x, t, P * c)
            x, _, _ = self._conv(self._norm_act(x, P, c, L["n2"]), T, Fw, L["c2"], pad=(1, 1))
            x = self._add(x, s, P * c)
        return x

    def _plan(self, dim_t: int):
        from .mdx import StftPlan
        if dim_t not in self._plans:
            self._plans[dim_t] = StftPlan(self.ctx, self.cfg.n_fft, self.cfg.hop, self.cfg.dim_f, dim_t)
        return self._plans[dim_t]

    # -- forward --------------------------------------------------------------------------------------------
    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        """audio [2, L] float32 on the device, L = hop * (T - 1), T divisible by 2**num_scales -> [num_stems, 2, L]"""
        ctx, cfg = self.ctx, self.cfg
        lib, h = ctx.lib, ctx.handle
        if audio.dim() != 2 or audio.shape[0] != 2 or audio.dtype != torch.float32:
            raise AlsepError("MDX23C.forward expects a float32 [2, L] tensor")
        L = audio.shape[-1]
        if L % cfg.hop or (L // cfg.hop + 1) % (2 ** cfg.num_scales):
            raise AlsepError(f"MDX23C.forward: {L} samples do not give a frame count divisible by 2**{cfg.num_scales}")
        audio = audio.contiguous()
        T, k = L // cfg.hop + 1, cfg.num_subbands
        f = cfg.dim_f // k
        plan = self._plan(T)
        spec = plan.stft_strided(audio, L, 2 * L, 1, torch.float32, _lib.LAYOUT_REF)          # [1, 4, dim_f, T]
        dc = cfg.dim_c
        mix = ctx.empty((T * f, dc))
        ctx.check(lib.alsep_mdx23c_spec_in(h, _lib.ptr(spec), _lib.ptr(mix), f, k, T), "alsep_mdx23c_spec_in")
        first, _, _ = self._conv(mix, T, f, self.first)
        x, Tc, Fc, c = first, T, f, cfg.num_channels
        skips = []
        for E in self.enc:
            x = self._block(x, Tc, Fc, E["blk"])
            skips.append((x, Tc, Fc, c))
            if self.half:
                x, Tc, Fc = self._conv_h(self._norm_act(x, Tc * Fc, c, E["dn"], True), Tc, Fc, E["dconv"], stride=(2, 2))
            else:
                x, Tc, Fc = self._conv(self._norm_act(x, Tc * Fc, c, E["dn"]), Tc, Fc, E["dconv"], stride=(2, 2))
            c = E["dconv"].cout
        x = self._block(x, Tc, Fc, self.bott)
        for D in self.dec:
            if self.half:
                g, _, _ = self._conv_h(self._norm_act(x, Tc * Fc, c, D["un"], True), Tc, Fc, D["up"])
            else:
                g, _, _ = self._conv(self._norm_act(x, Tc * Fc, c, D["un"]), Tc, Fc, D["up"])
            skip, Ts, Fs, cs = skips.pop()
            c = D["cout"]
            cat = ctx.empty((Ts * Fs, c + cs))
            ctx.check(lib.alsep_nn_depth_to_space2(h, _lib.ptr(g), _lib.ptr(cat), Tc, Fc, c, c + cs, 0), "alsep_nn_depth_to_space2")
            ctx.check(lib.alsep_vr_copy_slice(h, _lib.ptr(skip), _lib.ptr(cat), Ts, Fs, cs, 0, Fs, c + cs, c), "alsep_vr_copy_slice")
            Tc, Fc = Ts, Fs
            x = self._block(cat, Tc, Fc, D["blk"])
        P = T * f
        xm = ctx.empty((P, c))
        ctx.check(lib.alsep_nn_mul(h, _lib.ptr(x), _lib.ptr(first), _lib.ptr(xm), P * c), "alsep_nn_mul")
        cat = ctx.empty((P, dc + c))                                        # torch.cat([mix, x], 1)
        ctx.check(lib.alsep_vr_copy_slice(h, _lib.ptr(mix), _lib.ptr(cat), T, f, dc, 0, f, dc + c, 0), "alsep_vr_copy_slice")
        ctx.check(lib.alsep_vr_copy_slice(h, _lib.ptr(xm), _lib.ptr(cat), T, f, c, 0, f, dc + c, dc), "alsep_vr_copy_slice")
        y, _, _ = self._conv(cat, T, f, self.final0, act=ACT_GELU)
        y, _, _ = self._conv(y, T, f, self.final2)
        S = cfg.num_stems
        spec_out = ctx.empty((S, 4, cfg.dim_f, T))
        ctx.check(lib.alsep_mdx23c_spec_out(h, _lib.ptr(y), _lib.ptr(spec_out), S, f, k, T), "alsep_mdx23c_spec_out")
        out = ctx.empty((S, 2, L))
        plan.istft_strided(spec_out, _lib.LAYOUT_REF, out, L, 2 * L, 0, L, (S - 1) * 2 * L + L)
        return out

    __call__ = forward


# ---- synthetic weights (data only; bench / tests, allow_synthetic=True) -----------------------------------------------------------
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
