"""VR-architecture network (``CascadedASPPNet``) on the HIP kernels of csrc/vrnet.hip.

Mirrors the in-tree reference modules -- modules/rvc/infer/lib/uvr5_pack/lib_v5/nets*.py (``BaseASPPNet``,
``CascadedASPPNet.forward`` / ``predict``) and layers*.py (``Conv2DBNActiv``, ``SeperableConv2DBNActiv``, ``Encoder``,
``Decoder``, ``ASPPModule``) -- with the same parameter names, so a state_dict of the reference module loads unchanged.
Inference only: BatchNorm uses its running statistics (folded into a per-channel scale / shift), Dropout2d is the identity.

This is the network of the VR models the orchestrator names (woodwinds: stem_separator.py:596).  It is
pinned against the reference module itself (oracle/make_golden_vr.py).  The multi-band STFT front / back end of the VR
models (spec_utils.py, librosa resampling) is not built, so the network is reachable through this class only.

Tensors are channels-last on the device: ``[B, bins, frames, C]``; the reference layout ``[B, C, bins, frames]`` exists
at the ``forward`` / ``predict`` boundary.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import AlsepError, Context

# (stage-1 band net width, stage-2 bridge out, stage-2 width, stage-3 bridge out, stage-3 width) per reference file
WIDTHS = {"nets": (16, 8, 16, 16, 32), "nets_61968KB": (32, 16, 32, 32, 64), "nets_123821KB": (32, 16, 32, 32, 64),
          "nets_123812KB": (32, 16, 32, 32, 64)}
ACT = {"none": 0, "relu": 1, "leaky": 2}


def base_aspp_param_shapes(prefix: str, nin: int, ch: int):
    """(name, shape) of every tensor of a ``BaseASPPNet(nin, ch)`` in module order (nets*.py:9-21)."""
    out = []

    def cba(p, ci, co, k):                                   # Conv2DBNActiv: conv.0 (Conv2d), conv.1 (BatchNorm2d)
        out.append((f"{p}.conv.0.weight", (co, ci, k, k)))
        for n in ("weight", "bias", "running_mean", "running_var"):
            out.append((f"{p}.conv.1.{n}", (co,)))

    def sep(p, ci, co):                                      # SeperableConv2DBNActiv: depthwise, pointwise, BN
        out.append((f"{p}.conv.0.weight", (ci, 1, 3, 3)))
        out.append((f"{p}.conv.1.weight", (co, ci, 1, 1)))
        for n in ("weight", "bias", "running_mean", "running_var"):
            out.append((f"{p}.conv.2.{n}", (co,)))

    chans = [(nin, ch), (ch, 2 * ch), (2 * ch, 4 * ch), (4 * ch, 8 * ch)]
    for i, (ci, co) in enumerate(chans, 1):
        cba(f"{prefix}.enc{i}.conv1", ci, co, 3)
        cba(f"{prefix}.enc{i}.conv2", co, co, 3)
    c8 = 8 * ch
    cba(f"{prefix}.aspp.conv1.1", c8, c8, 1)
    cba(f"{prefix}.aspp.conv2", c8, c8, 1)
    for j in (3, 4, 5):
        sep(f"{prefix}.aspp.conv{j}", c8, c8)
    cba(f"{prefix}.aspp.bottleneck.0", 5 * c8, 16 * ch, 1)
    for i, (ci, co) in zip((4, 3, 2, 1), ((24 * ch, 8 * ch), (12 * ch, 4 * ch), (6 * ch, 2 * ch), (3 * ch, ch))):
        cba(f"{prefix}.dec{i}.conv", ci, co, 3)
    return out


def cascaded_param_shapes(widths: Sequence[int]):
    w1, b2, w2, b3, w3 = widths
    out = base_aspp_param_shapes("stg1_low_band_net", 2, w1) + base_aspp_param_shapes("stg1_high_band_net", 2, w1)
    out.append(("stg2_bridge.conv.0.weight", (b2, 2 + w1, 1, 1)))
    out += [(f"stg2_bridge.conv.1.{n}", (b2,)) for n in ("weight", "bias", "running_mean", "running_var")]
    out += base_aspp_param_shapes("stg2_full_band_net", b2, w2)
    out.append(("stg3_bridge.conv.0.weight", (b3, 2 + w1 + w2, 1, 1)))
    out += [(f"stg3_bridge.conv.1.{n}", (b3,)) for n in ("weight", "bias", "running_mean", "running_var")]
    out += base_aspp_param_shapes("stg3_full_band_net", b3, w3)
    out += [("out.weight", (2, w3, 1, 1)), ("aux1_out.weight", (2, w1, 1, 1)), ("aux2_out.weight", (2, w2, 1, 1))]
    return out


def random_state_dict(widths: Sequence[int], seed: int = 0) -> Dict[str, torch.Tensor]:
    """Seeded random weights with the reference's parameter names (fixtures, smoke runs): Kaiming-scaled convolutions,
    BatchNorm statistics away from the identity so that the folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in cascaded_param_shapes(widths):
        if name.endswith("running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean") or name.endswith(".bias"):
            t = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                                # BatchNorm gamma
            t = 0.8 + 0.4 * torch.rand(shape, generator=g)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            t = torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
        sd[name] = t.float()
    return sd


class _Conv:
    """Conv2d(bias=False) [+ BatchNorm2d eval] with weights packed [KH][KW][Cin][Cout] on the device."""

    def __init__(self, ctx: Context, sd, conv_key: str, bn_prefix: Optional[str], act: str, stride=1, pad=0, dil=1, eps=1e-5):
        w = sd[conv_key].float()
        self.cout, self.cin, self.kh, self.kw = w.shape
        self.w = w.permute(2, 3, 1, 0).contiguous().to(ctx.device)
        if bn_prefix is not None:
            gamma, beta = sd[bn_prefix + ".weight"].float(), sd[bn_prefix + ".bias"].float()
            mean, var = sd[bn_prefix + ".running_mean"].float(), sd[bn_prefix + ".running_var"].float()
            scale = gamma / torch.sqrt(var + eps)
            shift = beta - mean * scale
        else:
            scale, shift = torch.ones(self.cout), torch.zeros(self.cout)
        self.scale, self.shift = scale.contiguous().to(ctx.device), shift.contiguous().to(ctx.device)
        pair = lambda v: tuple(v) if isinstance(v, (tuple, list)) else (v, v)
        self.act, self.stride, self.pad, self.dil = ACT[act], stride, pair(pad), pair(dil)

    def out_hw(self, h, w):
        return ((h + 2 * self.pad[0] - self.dil[0] * (self.kh - 1) - 1) // self.stride + 1,
                (w + 2 * self.pad[1] - self.dil[1] * (self.kw - 1) - 1) // self.stride + 1)


class VRNet:
    """``CascadedASPPNet(n_fft)`` of the reference; ``variant`` picks the channel widths of one of its files."""

    def __init__(self, n_fft: int, state_dict: Dict[str, torch.Tensor], variant: str = "nets_61968KB", ctx: Optional[Context] = None):
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        if variant not in WIDTHS:
            raise AlsepError(f"unknown VR net variant '{variant}' (have {sorted(WIDTHS)})")
        self.widths = WIDTHS[variant]
        missing = [n for n, _ in cascaded_param_shapes(self.widths) if n not in state_dict]
        if missing:
            raise AlsepError(f"VR state_dict lacks {len(missing)} tensors, e.g. {missing[:3]}")
        for n, shape in cascaded_param_shapes(self.widths):
            if tuple(state_dict[n].shape) != tuple(shape):
                raise AlsepError(f"VR state_dict: {n} has shape {tuple(state_dict[n].shape)}, expected {shape}")
        self.max_bin, self.output_bin, self.offset = n_fft // 2, n_fft // 2 + 1, 128
        sd = state_dict
        w1, b2, w2, b3, w3 = self.widths
        self.nets = {p: self._build_base(sd, p, nin, ch) for p, nin, ch in (
            ("stg1_low_band_net", 2, w1), ("stg1_high_band_net", 2, w1), ("stg2_full_band_net", b2, w2), ("stg3_full_band_net", b3, w3))}
        self.stg2_bridge = _Conv(self.ctx, sd, "stg2_bridge.conv.0.weight", "stg2_bridge.conv.1", "relu")
        self.stg3_bridge = _Conv(self.ctx, sd, "stg3_bridge.conv.0.weight", "stg3_bridge.conv.1", "relu")
        self.out = _Conv(self.ctx, sd, "out.weight", None, "none")

    # -- construction ------------------------------------------------------------------------------
    def _build_base(self, sd, p, nin, ch):
        c = self.ctx
        net = {"ch": ch}
        for i in (1, 2, 3, 4):                                # Encoder: LeakyReLU, conv2 has stride 2 (nets*.py:12-15)
            net[f"enc{i}.conv1"] = _Conv(c, sd, f"{p}.enc{i}.conv1.conv.0.weight", f"{p}.enc{i}.conv1.conv.1", "leaky", 1, 1)
            net[f"enc{i}.conv2"] = _Conv(c, sd, f"{p}.enc{i}.conv2.conv.0.weight", f"{p}.enc{i}.conv2.conv.1", "leaky", 2, 1)
        net["aspp.conv1"] = _Conv(c, sd, f"{p}.aspp.conv1.1.conv.0.weight", f"{p}.aspp.conv1.1.conv.1", "relu")
        net["aspp.conv2"] = _Conv(c, sd, f"{p}.aspp.conv2.conv.0.weight", f"{p}.aspp.conv2.conv.1", "relu")
        for j, d in zip((3, 4, 5), (4, 8, 16)):              # separable: depthwise 3x3 dilated, pointwise + BN + ReLU
            dw = sd[f"{p}.aspp.conv{j}.conv.0.weight"].float()
            net[f"aspp.conv{j}.dw"] = (dw.reshape(dw.shape[0], 3, 3).contiguous().to(c.device), d)
            net[f"aspp.conv{j}.pw"] = _Conv(c, sd, f"{p}.aspp.conv{j}.conv.1.weight", f"{p}.aspp.conv{j}.conv.2", "relu")
        net["aspp.bottleneck"] = _Conv(c, sd, f"{p}.aspp.bottleneck.0.conv.0.weight", f"{p}.aspp.bottleneck.0.conv.1", "relu")
        for i in (4, 3, 2, 1):
            net[f"dec{i}"] = _Conv(c, sd, f"{p}.dec{i}.conv.conv.0.weight", f"{p}.dec{i}.conv.conv.1", "relu", 1, 1)
        return net

    # -- kernels -------------------------------------------------------------------------------------
    def _conv(self, L: _Conv, x: torch.Tensor, y: Optional[torch.Tensor] = None, c0: int = 0) -> torch.Tensor:
        b, h, w, cin = x.shape
        if cin != L.cin:
            raise AlsepError(f"conv expects {L.cin} input channels, got {cin}")
        ho, wo = L.out_hw(h, w)
        if y is None:
            y = self.ctx.empty((b, ho, wo, L.cout))
        ctx = self.ctx
        ctx.check(ctx.lib.alsep_vr_conv2d(ctx.handle, _lib.ptr(x), _lib.ptr(L.w), _lib.ptr(L.scale), _lib.ptr(L.shift), _lib.ptr(y),
                                          b, h, w, L.cin, L.cout, L.kh, L.kw, L.stride, L.pad[0], L.pad[1], L.dil[0], L.dil[1], L.act,
                                          y.shape[3], c0),
                  "alsep_vr_conv2d")
        return y

    def _resize(self, x, ho, wo, y, c0):
        b, h, w, c = x.shape
        ctx = self.ctx
        ctx.check(ctx.lib.alsep_vr_resize_bilinear(ctx.handle, _lib.ptr(x), _lib.ptr(y), b, h, w, c, ho, wo, y.shape[3], c0),
                  "alsep_vr_resize_bilinear")

    def _copy(self, x, y, c0, w_off=0):
        b, h, wx, c = x.shape
        ctx = self.ctx
        ctx.check(ctx.lib.alsep_vr_copy_slice(ctx.handle, _lib.ptr(x), _lib.ptr(y), b * h, wx, c, w_off, y.shape[2], y.shape[3], c0),
                  "alsep_vr_copy_slice")

    # -- modules ---------------------------------------------------------------------------------------
    def _decoder(self, L: _Conv, x, skip):
        """layers*.py:83-93: upsample x2 (bilinear, align_corners), crop_center(skip) along frames, cat, conv."""
        b, h, w, c = x.shape
        ho, wo = 2 * h, 2 * w
        if skip.shape[1] != ho or skip.shape[2] < wo:
            raise AlsepError(f"decoder: skip {tuple(skip.shape)} does not fit the upsampled {ho}x{wo} map")
        cat = self.ctx.empty((b, ho, wo, c + skip.shape[3]))
        self._resize(x, ho, wo, cat, 0)
        self._copy(skip, cat, c, (skip.shape[2] - wo) // 2)
        return self._conv(L, cat)

    def _aspp(self, net, x):
        """layers*.py:96-125."""
        ctx = self.ctx
        b, h, w, c = x.shape
        cat = ctx.empty((b, h, w, 5 * c))
        pooled = ctx.empty((b, 1, w, c))
        ctx.check(ctx.lib.alsep_vr_mean_h(ctx.handle, _lib.ptr(x), _lib.ptr(pooled), b, h, w, c), "alsep_vr_mean_h")
        self._resize(self._conv(net["aspp.conv1"], pooled), h, w, cat, 0)
        self._conv(net["aspp.conv2"], x, cat, c)
        for k, j in enumerate((3, 4, 5)):
            dw, d = net[f"aspp.conv{j}.dw"]
            t = ctx.empty((b, h, w, c))
            ctx.check(ctx.lib.alsep_vr_depthwise(ctx.handle, _lib.ptr(x), _lib.ptr(dw), _lib.ptr(t), b, h, w, c, 3, 3, d, d),
                      "alsep_vr_depthwise")
            self._conv(net[f"aspp.conv{j}.pw"], t, cat, (2 + k) * c)
        return self._conv(net["aspp.bottleneck"], cat)

    def _base(self, name, x):
        """nets*.py:23-37."""
        net = self.nets[name]
        skips = []
        h = x
        for i in (1, 2, 3, 4):
            s = self._conv(net[f"enc{i}.conv1"], h)
            h = self._conv(net[f"enc{i}.conv2"], s)
            skips.append(s)
        h = self._aspp(net, h)
        for i in (4, 3, 2, 1):
            h = self._decoder(net[f"dec{i}"], h, skips[i - 1])
        return h

    # -- forward -------------------------------------------------------------------------------------------
    def forward_nhwc(self, x: torch.Tensor, aggressiveness: Optional[dict] = None) -> torch.Tensor:
        """x [B, bins >= max_bin, frames, 2] magnitudes -> mask * x, [B, output_bin, frames, 2] (nets*.py:59-111, eval)."""
        ctx = self.ctx
        if x.dim() != 4 or x.shape[3] != 2 or x.shape[1] < self.output_bin or x.dtype != torch.float32:
            raise AlsepError(f"VRNet input must be float32 [B, >= {self.output_bin} bins, frames, 2], got {tuple(x.shape)}")
        x = x.contiguous()
        mix = x[:, : self.output_bin].contiguous()
        xin = x[:, : self.max_bin].contiguous()
        b, hh, w, _ = xin.shape
        bandw = hh // 2
        w1, b2, w2, b3, w3 = self.widths
        low = self._base("stg1_low_band_net", xin[:, :bandw].contiguous())
        high = self._base("stg1_high_band_net", xin[:, bandw:].contiguous())
        h1 = ctx.empty((b, hh, w, 2 + w1))                      # cat([x, aux1], channels)
        self._copy(xin, h1, 0)
        aux1 = torch.cat([low, high], dim=1).contiguous()       # along bins: plain tensor plumbing
        self._copy(aux1, h1, 2)
        aux2 = self._base("stg2_full_band_net", self._conv(self.stg2_bridge, h1))
        h2 = ctx.empty((b, hh, w, 2 + w1 + w2))
        self._copy(h1, h2, 0)
        self._copy(aux2, h2, 2 + w1)
        h3 = self._base("stg3_full_band_net", self._conv(self.stg3_bridge, h2))
        logit = self._conv(self.out, h3)
        out = ctx.empty(tuple(mix.shape))
        split, aggr = (int(aggressiveness["split_bin"]), float(aggressiveness["value"])) if aggressiveness else (0, -1.0)
        ctx.check(ctx.lib.alsep_vr_mask(ctx.handle, _lib.ptr(logit), _lib.ptr(mix), _lib.ptr(out), b, logit.shape[1], self.output_bin,
                                        w, 2, split, C.c_float(aggr)), "alsep_vr_mask")
        return out

    def forward(self, x_mag: torch.Tensor, aggressiveness: Optional[dict] = None) -> torch.Tensor:
        """Reference layout [B, 2, bins, frames] in and out."""
        x = torch.as_tensor(x_mag, dtype=torch.float32).permute(0, 2, 3, 1).contiguous().to(self.ctx.device)
        return self.forward_nhwc(x, aggressiveness).permute(0, 3, 1, 2).contiguous()

    def predict(self, x_mag: torch.Tensor, aggressiveness: Optional[dict] = None) -> torch.Tensor:
        """nets*.py:113-121: forward, then drop ``offset`` frames on both sides."""
        h = self.forward(x_mag, aggressiveness)
        if self.offset > 0:
            h = h[:, :, :, self.offset:-self.offset]
            if h.shape[3] <= 0:
                raise AlsepError("predict: fewer than 2 * offset frames")
        return h


def make_padding(width: int, cropsize: int, offset: int) -> Tuple[int, int, int]:
    """utils.py:15-22."""
    left = offset
    roi_size = cropsize - left * 2
    if roi_size == 0:
        roi_size = cropsize
    right = roi_size - (width % roi_size) + left
    return left, right, roi_size


def vr_inference(net: VRNet, x_spec: torch.Tensor, aggressiveness: Optional[dict], window_size: int, tta: bool = False,
                 max_batch: int = 4):
    """The VR runner: reference modules/rvc/infer/lib/uvr5_pack/utils.py:25-100 (``inference``) on the device.

    x_spec complex64 [2, bins, frames] -> (pred * coef [2, bins, frames], |X|, exp(i angle X)).  The magnitude is
    normalised by its maximum, zero-padded by ``offset`` frames (plus the remainder of the last window), cut into windows
    of ``window_size`` frames every ``roi_size = window_size - 2 offset`` and each window goes through ``predict`` (the
    network, then ``offset`` frames dropped on both sides); with ``tta`` a second pass shifted by half a window is
    averaged in.  Windows are independent: here they run ``max_batch`` at a time instead of one by one."""
    ctx = net.ctx
    x_spec = x_spec.to(ctx.device)
    x_mag = x_spec.abs().float()
    phase = torch.polar(torch.ones_like(x_mag), torch.angle(x_spec))
    coef = x_mag.max()
    pre = x_mag / coef
    n_frame = pre.shape[2]
    off = net.offset

    def execute(pad_l, pad_r, roi, n_window):
        padded = torch.nn.functional.pad(pre, (pad_l, pad_r))                      # [2, bins, frames]
        nhwc = padded.permute(1, 2, 0).contiguous()                                # [bins, frames, 2]
        preds = []
        for w0 in range(0, n_window, max_batch):
            nb = min(max_batch, n_window - w0)
            win = torch.stack([nhwc[:, (w0 + i) * roi:(w0 + i) * roi + window_size] for i in range(nb)]).contiguous()
            if win.shape[2] != window_size:
                raise AlsepError("vr_inference: window runs past the padded spectrogram")
            out = net.forward_nhwc(win, aggressiveness)                            # [nb, output_bin, window, 2]
            out = out[:, :, off:window_size - off] if off > 0 else out
            preds.extend(out[i] for i in range(nb))
        return torch.cat(preds, dim=1).permute(2, 0, 1)                            # [2, bins, n_window * roi]

    pad_l, pad_r, roi = make_padding(n_frame, window_size, off)
    n_window = -(-n_frame // roi)
    pred = execute(pad_l, pad_r, roi, n_window)[:, :, :n_frame]
    if tta:
        pred_t = execute(pad_l + roi // 2, pad_r + roi // 2, roi, n_window + 1)[:, :, roi // 2:][:, :, :n_frame]
        pred = (pred + pred_t) * 0.5
    return pred * coef, x_mag, phase


# ==================================================================================================
# nets_new.py: CascadedNet (BaseNet with an LSTM branch) -- the architecture of the in-tree DeEcho / DeReverb runner
# (modules/rvc/infer/modules/uvr5/vr.py AudioPreDeEcho) and of the "VR 5.1" models the orchestrator names for noise / echo
# removal and the BG-vocal split (stem_separator.py:752, 798-799).
# ==================================================================================================
def _cba_shapes(out, p, ci, co, k):
    out.append((f"{p}.conv.0.weight", (co, ci, k, k)))
    for n in ("weight", "bias", "running_mean", "running_var"):
        out.append((f"{p}.conv.1.{n}", (co,)))


def basenet_param_shapes(prefix: str, nin: int, nout: int, nin_lstm: int, nout_lstm: int):
    """(name, shape) of a ``BaseNet`` (nets_new.py:9-29) in module order."""
    out = []
    _cba_shapes(out, f"{prefix}.enc1", nin, nout, 3)
    for i, (ci, co) in zip((2, 3, 4, 5), ((nout, 2 * nout), (2 * nout, 4 * nout), (4 * nout, 6 * nout), (6 * nout, 8 * nout))):
        _cba_shapes(out, f"{prefix}.enc{i}.conv1", ci, co, 3)
        _cba_shapes(out, f"{prefix}.enc{i}.conv2", co, co, 3)
    c8 = 8 * nout
    _cba_shapes(out, f"{prefix}.aspp.conv1.1", c8, c8, 1)
    _cba_shapes(out, f"{prefix}.aspp.conv2", c8, c8, 1)
    for j in (3, 4, 5):
        _cba_shapes(out, f"{prefix}.aspp.conv{j}", c8, c8, 3)
    _cba_shapes(out, f"{prefix}.aspp.bottleneck", 5 * c8, c8, 1)
    for i, (ci, co) in zip((4, 3, 2), ((14 * nout, 6 * nout), (10 * nout, 4 * nout), (6 * nout, 2 * nout))):
        _cba_shapes(out, f"{prefix}.dec{i}.conv1", ci, co, 3)
    _cba_shapes(out, f"{prefix}.lstm_dec2.conv", 2 * nout, 1, 1)
    hd = nout_lstm // 2
    for sfx in ("", "_reverse"):
        out += [(f"{prefix}.lstm_dec2.lstm.weight_ih_l0{sfx}", (4 * hd, nin_lstm)), (f"{prefix}.lstm_dec2.lstm.weight_hh_l0{sfx}", (4 * hd, hd)),
                (f"{prefix}.lstm_dec2.lstm.bias_ih_l0{sfx}", (4 * hd,)), (f"{prefix}.lstm_dec2.lstm.bias_hh_l0{sfx}", (4 * hd,))]
    out += [(f"{prefix}.lstm_dec2.dense.0.weight", (nin_lstm, nout_lstm)), (f"{prefix}.lstm_dec2.dense.0.bias", (nin_lstm,))]
    out += [(f"{prefix}.lstm_dec2.dense.1.{n}", (nin_lstm,)) for n in ("weight", "bias", "running_mean", "running_var")]
    _cba_shapes(out, f"{prefix}.dec1.conv1", 3 * nout + 1, nout, 3)
    return out


def cascaded_new_param_shapes(n_fft: int, nout: int, nout_lstm: int):
    nin_lstm = (n_fft // 2) // 2
    out = basenet_param_shapes("stg1_low_band_net.0", 2, nout // 2, nin_lstm // 2, nout_lstm)
    _cba_shapes(out, "stg1_low_band_net.1", nout // 2, nout // 4, 1)
    out += basenet_param_shapes("stg1_high_band_net", 2, nout // 4, nin_lstm // 2, nout_lstm // 2)
    out += basenet_param_shapes("stg2_low_band_net.0", nout // 4 + 2, nout, nin_lstm // 2, nout_lstm)
    _cba_shapes(out, "stg2_low_band_net.1", nout, nout // 2, 1)
    out += basenet_param_shapes("stg2_high_band_net", nout // 4 + 2, nout // 2, nin_lstm // 2, nout_lstm // 2)
    out += basenet_param_shapes("stg3_full_band_net", 3 * nout // 4 + 2, nout, nin_lstm, nout_lstm)
    out += [("out.weight", (2, nout, 1, 1)), ("aux_out.weight", (2, 3 * nout // 4, 1, 1))]
    return out


def random_state_dict_new(n_fft: int, nout: int, nout_lstm: int, seed: int = 0) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in cascaded_new_param_shapes(n_fft, nout, nout_lstm):
        if name.endswith("running_var"):
            t = 0.5 + torch.rand(shape, generator=g)
        elif name.endswith("running_mean") or ".bias" in name:
            t = 0.1 * torch.randn(shape, generator=g)
        elif len(shape) == 1:
            t = 0.8 + 0.4 * torch.rand(shape, generator=g)
        elif len(shape) == 2:                                # LSTM / Linear matrices
            t = torch.randn(shape, generator=g) * (1.0 / shape[1]) ** 0.5
        else:
            t = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
        sd[name] = t.float()
    return sd


class VRNetNew(VRNet):
    """``CascadedNet(n_fft, nout, nout_lstm)`` of nets_new.py; parameter names as in the reference module."""

    def __init__(self, n_fft: int, state_dict: Dict[str, torch.Tensor], nout: int = 32, nout_lstm: int = 128,
                 ctx: Optional[Context] = None):
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        shapes = cascaded_new_param_shapes(n_fft, nout, nout_lstm)
        for n, shape in shapes:
            if n not in state_dict:
                raise AlsepError(f"VR (new) state_dict lacks {n}")
            if tuple(state_dict[n].shape) != tuple(shape):
                raise AlsepError(f"VR (new) state_dict: {n} has shape {tuple(state_dict[n].shape)}, expected {shape}")
        self.max_bin, self.output_bin, self.offset = n_fft // 2, n_fft // 2 + 1, 64
        self.nout = nout
        sd = state_dict
        self.base = {p: self._build_basenet(sd, p) for p in ("stg1_low_band_net.0", "stg1_high_band_net", "stg2_low_band_net.0",
                                                             "stg2_high_band_net", "stg3_full_band_net")}
        self.stg1_low_tail = _Conv(self.ctx, sd, "stg1_low_band_net.1.conv.0.weight", "stg1_low_band_net.1.conv.1", "relu")
        self.stg2_low_tail = _Conv(self.ctx, sd, "stg2_low_band_net.1.conv.0.weight", "stg2_low_band_net.1.conv.1", "relu")
        self.out = _Conv(self.ctx, sd, "out.weight", None, "none")

    def _build_basenet(self, sd, p):
        c = self.ctx
        net = {"enc1": _Conv(c, sd, f"{p}.enc1.conv.0.weight", f"{p}.enc1.conv.1", "relu", 1, 1)}
        for i in (2, 3, 4, 5):                                # Encoder (layers_new.py:30-40): conv1 strided, LeakyReLU
            net[f"enc{i}.conv1"] = _Conv(c, sd, f"{p}.enc{i}.conv1.conv.0.weight", f"{p}.enc{i}.conv1.conv.1", "leaky", 2, 1)
            net[f"enc{i}.conv2"] = _Conv(c, sd, f"{p}.enc{i}.conv2.conv.0.weight", f"{p}.enc{i}.conv2.conv.1", "leaky", 1, 1)
        net["aspp.conv1"] = _Conv(c, sd, f"{p}.aspp.conv1.1.conv.0.weight", f"{p}.aspp.conv1.1.conv.1", "relu")
        net["aspp.conv2"] = _Conv(c, sd, f"{p}.aspp.conv2.conv.0.weight", f"{p}.aspp.conv2.conv.1", "relu")
        for j, d in zip((3, 4, 5), ((4, 2), (8, 4), (12, 6))):           # nets_new.py:11, layers_new.py:83-91
            net[f"aspp.conv{j}"] = _Conv(c, sd, f"{p}.aspp.conv{j}.conv.0.weight", f"{p}.aspp.conv{j}.conv.1", "relu", 1, d, d)
        net["aspp.bottleneck"] = _Conv(c, sd, f"{p}.aspp.bottleneck.conv.0.weight", f"{p}.aspp.bottleneck.conv.1", "relu")
        for i in (4, 3, 2, 1):
            net[f"dec{i}"] = _Conv(c, sd, f"{p}.dec{i}.conv1.conv.0.weight", f"{p}.dec{i}.conv1.conv.1", "relu", 1, 1)
        # LSTMModule (layers_new.py:108-125)
        net["lstm.conv"] = _Conv(c, sd, f"{p}.lstm_dec2.conv.conv.0.weight", f"{p}.lstm_dec2.conv.conv.1", "relu")
        lstm = {}
        for d, sfx in enumerate(("", "_reverse")):
            wih = sd[f"{p}.lstm_dec2.lstm.weight_ih_l0{sfx}"].float()
            lstm[d] = {"wih": wih.t().contiguous().to(c.device),                       # [nin, 4 Hd] = 1x1 conv weights
                       "bias": (sd[f"{p}.lstm_dec2.lstm.bias_ih_l0{sfx}"] + sd[f"{p}.lstm_dec2.lstm.bias_hh_l0{sfx}"]).float().contiguous().to(c.device),
                       "ones": torch.ones(wih.shape[0]).to(c.device),
                       "whh": sd[f"{p}.lstm_dec2.lstm.weight_hh_l0{sfx}"].float().contiguous().to(c.device)}
        net["lstm"] = lstm
        net["lstm.hidden"] = lstm[0]["whh"].shape[1]
        wd, bd = sd[f"{p}.lstm_dec2.dense.0.weight"].float(), sd[f"{p}.lstm_dec2.dense.0.bias"].float()
        gamma, beta = sd[f"{p}.lstm_dec2.dense.1.weight"].float(), sd[f"{p}.lstm_dec2.dense.1.bias"].float()
        mean, var = sd[f"{p}.lstm_dec2.dense.1.running_mean"].float(), sd[f"{p}.lstm_dec2.dense.1.running_var"].float()
        scale = gamma / torch.sqrt(var + 1e-5)
        net["dense"] = {"w": wd.t().contiguous().to(c.device), "scale": scale.contiguous().to(c.device),
                        "shift": ((bd - mean) * scale + beta).contiguous().to(c.device)}
        return net

    def _linear(self, x2d: torch.Tensor, w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor, act: int) -> torch.Tensor:
        """rows x Cin -> rows x Cout through the 1x1 case of the conv kernel."""
        rows, cin = x2d.shape
        cout = w.shape[1]
        y = self.ctx.empty((rows, cout))
        ctx = self.ctx
        ctx.check(ctx.lib.alsep_vr_conv2d(ctx.handle, _lib.ptr(x2d), _lib.ptr(w), _lib.ptr(scale), _lib.ptr(shift), _lib.ptr(y),
                                          1, rows, 1, cin, cout, 1, 1, 1, 0, 0, 1, 1, act, cout, 0), "alsep_vr_conv2d")
        return y

    def _lstm_module(self, net, h: torch.Tensor) -> torch.Tensor:
        """h [N, bins, frames, C] -> [N, bins, frames, 1]."""
        ctx = self.ctx
        n, nbins, nframes, _ = h.shape
        seq = self._conv(net["lstm.conv"], h)[..., 0].permute(2, 0, 1).contiguous()         # [frames, N, bins]
        hd = net["lstm.hidden"]
        both = ctx.empty((nframes, n, 2 * hd))
        for d in (0, 1):
            L = net["lstm"][d]
            if L["wih"].shape[0] != nbins:
                raise AlsepError(f"LSTM input size {L['wih'].shape[0]} != {nbins} bins")
            pre = self._linear(seq.reshape(nframes * n, nbins), L["wih"], L["ones"], L["bias"], 0)
            ctx.check(ctx.lib.alsep_vr_lstm(ctx.handle, _lib.ptr(pre), _lib.ptr(L["whh"]), _lib.ptr(both), nframes, n, hd, 2 * hd,
                                            d * hd, d), "alsep_vr_lstm")
        D = net["dense"]
        out = self._linear(both.reshape(nframes * n, 2 * hd), D["w"], D["scale"], D["shift"], 1)   # [frames * N, bins]
        return out.reshape(nframes, n, nbins).permute(1, 2, 0).unsqueeze(-1).contiguous()

    def _basenet(self, name, x):
        """nets_new.py:31-47."""
        net = self.base[name]
        e1 = self._conv(net["enc1"], x)
        es = [e1]
        h = e1
        for i in (2, 3, 4, 5):
            h = self._conv(net[f"enc{i}.conv2"], self._conv(net[f"enc{i}.conv1"], h))
            es.append(h)
        e1, e2, e3, e4, e5 = es
        # ASPP with full dilated convolutions (layers_new.py:73-105); Dropout2d is the identity in eval
        ctx = self.ctx
        b, hh, ww, c = e5.shape
        cat = ctx.empty((b, hh, ww, 5 * c))
        pooled = ctx.empty((b, 1, ww, c))
        ctx.check(ctx.lib.alsep_vr_mean_h(ctx.handle, _lib.ptr(e5), _lib.ptr(pooled), b, hh, ww, c), "alsep_vr_mean_h")
        self._resize(self._conv(net["aspp.conv1"], pooled), hh, ww, cat, 0)
        self._conv(net["aspp.conv2"], e5, cat, c)
        for k, j in enumerate((3, 4, 5)):
            self._conv(net[f"aspp.conv{j}"], e5, cat, (2 + k) * c)
        h = self._conv(net["aspp.bottleneck"], cat)
        h = self._decoder(net["dec4"], h, e4)
        h = self._decoder(net["dec3"], h, e3)
        h = self._decoder(net["dec2"], h, e2)
        h = torch.cat([h, self._lstm_module(net, h)], dim=3).contiguous()
        return self._decoder(net["dec1"], h, e1)

    def mask_nhwc(self, x: torch.Tensor):
        """x [B, bins >= max_bin, frames, 2] -> (logits [B, max_bin, frames, 2], the cropped input) (nets_new.py:81-105)."""
        xin = x[:, : self.max_bin].contiguous()
        bandw = xin.shape[1] // 2
        l1_in, h1_in = xin[:, :bandw].contiguous(), xin[:, bandw:].contiguous()
        l1 = self._conv(self.stg1_low_tail, self._basenet("stg1_low_band_net.0", l1_in))
        h1 = self._basenet("stg1_high_band_net", h1_in)
        aux1 = torch.cat([l1, h1], dim=1)
        l2 = self._conv(self.stg2_low_tail, self._basenet("stg2_low_band_net.0", torch.cat([l1_in, l1], dim=3).contiguous()))
        h2 = self._basenet("stg2_high_band_net", torch.cat([h1_in, h1], dim=3).contiguous())
        aux2 = torch.cat([l2, h2], dim=1)
        f3 = self._basenet("stg3_full_band_net", torch.cat([xin, aux1, aux2], dim=3).contiguous())
        return self._conv(self.out, f3)

    def forward_nhwc(self, x: torch.Tensor, aggressiveness: Optional[dict] = None) -> torch.Tensor:
        """x [B, bins, frames, 2] -> x * mask [B, output_bin, frames, 2] (``predict`` before its offset crop; the new nets
        ignore ``aggressiveness``, nets_new.py:124-132)."""
        ctx = self.ctx
        if x.dim() != 4 or x.shape[3] != 2 or x.shape[1] < self.output_bin or x.dtype != torch.float32:
            raise AlsepError(f"VRNetNew input must be float32 [B, >= {self.output_bin} bins, frames, 2], got {tuple(x.shape)}")
        x = x.contiguous()
        mix = x[:, : self.output_bin].contiguous()
        logit = self.mask_nhwc(x)
        out = ctx.empty(tuple(mix.shape))
        ctx.check(ctx.lib.alsep_vr_mask(ctx.handle, _lib.ptr(logit), _lib.ptr(mix), _lib.ptr(out), mix.shape[0], logit.shape[1],
                                        self.output_bin, mix.shape[2], 2, 0, C.c_float(-1.0)), "alsep_vr_mask")
        return out
