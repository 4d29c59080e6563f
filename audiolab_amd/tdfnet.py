"""TFC-TDF U-Net (MDX-Net "ConvTDFNet") on the GPU: host-side model object over alsep_net.

This is what the reference reaches through ``MDXSeparator.model_run(spek)``
(handlers/patch_separate.py:52,58-62) after ``Separator.load_model('*.onnx')``
(modules/separator/stem_separator.py:394,512).  Weights come as a torch ``state_dict``-style
dict (``first_conv.0.weight``, ``encoding_blocks.0.tfc.H.0.1.running_var`` ...); BatchNorm is
folded here into per-channel (scale, shift) and everything else happens in libalsep.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import AlsepError, Context

BN_EPS = 1e-5


@dataclass(frozen=True)
class TDFNetConfig:
    dim_f: int = 3072
    dim_t: int = 256          # frames (2**dim_t_arg of mdxnet.py:22)
    n_fft: int = 6144
    hop: int = 1024
    num_blocks: int = 11      # L
    l: int = 3
    g: int = 48
    k: int = 3
    bn: int = 8
    bias: bool = True
    dim_c: int = 4

    @property
    def n(self) -> int:
        return self.num_blocks // 2

    def levels(self):
        """[(channels, frames, bins)] for encoder levels 0..n (n = bottleneck)."""
        out = []
        c, t, f = self.g, self.dim_t, self.dim_f
        for _ in range(self.n + 1):
            out.append((c, t, f))
            c += self.g
            t //= 2
            f //= 2
        return out

    def flops_per_chunk(self) -> float:
        """MACs*2 of one forward over one chunk (convs + TDF + ds/us + 1x1)."""
        lv = self.levels()
        total = 0.0

        def block(c, t, f):
            s = self.l * 2.0 * 9 * c * c * t * f
            if self.bn is not None:
                if self.bn == 0:
                    s += 2.0 * c * t * f * f
                else:
                    s += 2 * 2.0 * c * t * f * (f // self.bn)
            return s
        for i, (c, t, f) in enumerate(lv):
            total += block(c, t, f) * (1 if i == self.n else 2)
        for i in range(self.n):
            c, t, f = lv[i]
            total += 2 * 2.0 * 4 * c * (c + self.g) * (t // 2) * (f // 2)      # ds + us
        total += 2 * 2.0 * 4 * self.g * self.dim_t * self.dim_f                # first + final 1x1
        return total


def fold_batchnorm(sd: Dict[str, torch.Tensor], conv: str, bn: str, conv_bias_per_channel: bool = True):
    """(scale, shift) with y = scale * conv_nobias(x) + shift  ==  BN(conv(x)) in eval mode.  A layer without BatchNorm
    entries (an ONNX export folds Conv+BN into the convolution: onnx_reader.py) is scale 1, shift = bias."""
    b = sd.get(conv + ".bias")
    if bn + ".weight" not in sd:
        n = sd[conv + ".weight"].shape[1 if conv.startswith("us.") else 0]
        return torch.ones(n), (b.float().clone() if b is not None and conv_bias_per_channel else torch.zeros(n))
    gamma, beta = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
    mean, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
    scale = gamma / torch.sqrt(var + BN_EPS)
    shift = beta - mean * scale
    if b is not None and conv_bias_per_channel:
        shift = shift + b.double() * scale
    return scale.float(), shift.float()


def folded_tensors(sd: Dict[str, torch.Tensor], cfg: TDFNetConfig) -> Dict[str, torch.Tensor]:
    """state_dict -> the flat table alsep_net_create expects (names in tdfnet.hip build_net)."""
    out: Dict[str, torch.Tensor] = {}

    def conv_bn(dst: str, conv: str, bn: str):
        out[dst + ".weight"] = sd[conv + ".weight"]
        out[dst + ".scale"], out[dst + ".shift"] = fold_batchnorm(sd, conv, bn)

    def block(dst: str, src: str):
        for j in range(cfg.l):
            conv_bn(f"{dst}.tfc.{j}", f"{src}.tfc.H.{j}.0", f"{src}.tfc.H.{j}.1")
        if cfg.bn is not None:
            for j in range(1 if cfg.bn == 0 else 2):
                lin, bn = f"{src}.tdf.{3 * j}", f"{src}.tdf.{3 * j + 1}"
                out[f"{dst}.tdf.{j}.weight"] = sd[lin + ".weight"]
                # the Linear bias is per output FEATURE (f'), BN is per CHANNEL: keep them apart
                s, sh = fold_batchnorm(sd, lin, bn, conv_bias_per_channel=False)
                out[f"{dst}.tdf.{j}.scale"], out[f"{dst}.tdf.{j}.shift"] = s, sh
                if lin + ".bias" in sd:
                    out[f"{dst}.tdf.{j}.bias"] = sd[lin + ".bias"]

    conv_bn("first_conv", "first_conv.0", "first_conv.1")
    for i in range(cfg.n):
        block(f"encoding_blocks.{i}", f"encoding_blocks.{i}")
        conv_bn(f"ds.{i}", f"ds.{i}.0", f"ds.{i}.1")
        block(f"decoding_blocks.{i}", f"decoding_blocks.{i}")
        conv_bn(f"us.{i}", f"us.{i}.0", f"us.{i}.1")
    block("bottleneck_block", "bottleneck_block")
    out["final_conv.weight"] = sd["final_conv.0.weight"]
    fb = sd.get("final_conv.0.bias")
    out["final_conv.bias"] = fb if fb is not None else torch.zeros(cfg.dim_c)
    return out


class TDFNet:
    """One loaded MDX-Net model on one GPU.  ``forward_nhwc`` takes/returns the channels-last
    spectrogram ``[B, dim_t, dim_f, 4]`` in the model dtype; ``run`` offers the ORT-session
    surface of the reference seam (``run(None, {"input": spek[B,4,dim_f,dim_t]}) -> [pred]``)."""

    def __init__(self, cfg: TDFNetConfig, state_dict: Dict[str, torch.Tensor], ctx: Optional[Context] = None,
                 dtype: torch.dtype = torch.float32, max_batch: int = 4, contraction: Optional[str] = None):
        """``contraction`` (float32 networks only): "split" (default) = float32 storage with every contraction as three f16 MFMA products
        of (hi, lo) half pairs, float32 accumulation -- the float32 mode's accuracy (2^-22 per product) at five times its matrix
        throughput, activations limited to the half range (65504; a batch that exceeds it is detected on the device and run again on the
        exact kernels, see forward_nhwc); "exact" =
        v_mfma_f32_16x16x4_f32, bit for bit an fmaf chain."""
        if contraction is None:
            contraction = "split" if dtype == torch.float32 else "native"
        if (dtype == torch.float32) != (contraction in ("split", "exact")) or contraction not in ("split", "exact", "native"):
            raise AlsepError(f"contraction={contraction!r} does not fit dtype {dtype} ('split' / 'exact' are the float32 modes)")
        self.contraction = contraction
        if cfg.k != 3:
            raise AlsepError("only k=3 TFC convolutions are implemented")
        if cfg.bn is None:
            raise AlsepError("bn=None (no TDF) is not implemented")
        if cfg.dim_f % (2 ** cfg.n) or cfg.dim_t % (2 ** cfg.n) or (cfg.bn and (cfg.dim_f >> cfg.n) % cfg.bn):
            raise AlsepError(f"dim_f={cfg.dim_f} / dim_t={cfg.dim_t} must be divisible by 2^{cfg.n} (and by bn at the bottleneck)")
        if cfg.g % 16:
            raise AlsepError("g must be a multiple of 16")
        self.cfg = cfg
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        self.dtype = dtype
        self.max_batch = max_batch
        try:
            table = folded_tensors(state_dict, cfg)
        except KeyError as e:
            raise AlsepError(f"state_dict is missing {e} for this TDFNetConfig") from e
        self._table = table if contraction == "split" else None       # kept (host tensors, a few MB) for the exact twin of a split network
        self.handle = self._create(table, _lib.NET_SPLIT_F16 if contraction == "split" else 0)
        self._exact = None                                             # the same weights on the f32 MFMA kernels, built on first need
        self._ws: Optional[torch.Tensor] = None
        self._ws_batch = 0

    def _create(self, table: Dict[str, torch.Tensor], flags: int) -> C.c_void_p:
        cfg = self.cfg
        keep: List[torch.Tensor] = []
        entries = (_lib.TensorEntry * len(table))()
        for i, (name, t) in enumerate(table.items()):
            d = t.detach().to(device=self.ctx.device, dtype=torch.float32).contiguous()
            keep.append(d)
            entries[i].name = name.encode()
            entries[i].data = d.data_ptr()
            entries[i].numel = d.numel()
        ncfg = _lib.NetConfig(cfg.dim_f, cfg.dim_t, cfg.num_blocks, cfg.l, cfg.g, cfg.bn, _lib.dtype_code(self.dtype), flags)
        self.ctx.synchronize()
        h = C.c_void_p()
        self.ctx.check(self.ctx.lib.alsep_net_create(self.ctx.handle, C.byref(ncfg), entries, len(table), C.byref(h)), "alsep_net_create")
        del keep
        return h

    def __del__(self):
        try:
            for attr in ("handle", "_exact"):
                h = getattr(self, attr, None)
                if h and self.ctx.handle:
                    self.ctx.lib.alsep_net_destroy(h)
                    setattr(self, attr, None)
        except Exception:
            pass

    def range_exceeded(self) -> bool:
        """split-contraction networks: True when a forward since the last call met an activation beyond the half range (its results are
        invalid); reads and clears the network's range word (synchronises the stream).  Always False for the other modes."""
        if self.contraction != "split":
            return False
        flag = C.c_int32(0)
        self.ctx.check(self.ctx.lib.alsep_net_range_flag(self.ctx.handle, self.handle, C.byref(flag)), "alsep_net_range_flag")
        return bool(flag.value)

    def workspace(self, batch: int) -> torch.Tensor:
        if self._ws is None or self._ws_batch < batch:
            nbytes = self.ctx.lib.alsep_net_workspace_bytes(self.handle, batch)
            self._ws = self.ctx.empty((nbytes + 256,), torch.uint8)
            self._ws_batch = batch
        return self._ws

    def _forward(self, spek: torch.Tensor, out: torch.Tensor, in_scale: float = 1.0, alpha: float = 1.0,
                 beta: float = 0.0, handle=None) -> None:
        b = spek.shape[0]
        ws = self.workspace(b)
        base = (ws.data_ptr() + 255) & ~255
        self.ctx.check(self.ctx.lib.alsep_net_forward(self.ctx.handle, handle or self.handle, _lib.ptr(spek), _lib.ptr(out), b,
                                                      C.c_void_p(base), ws.numel() - 256, in_scale, alpha, beta),
                       "alsep_net_forward")

    def forward_nhwc(self, spek: torch.Tensor, denoise: bool = False) -> torch.Tensor:
        cfg = self.cfg
        if tuple(spek.shape[1:]) != (cfg.dim_t, cfg.dim_f, 4) or spek.dtype != self.dtype:
            raise AlsepError(f"forward_nhwc: expected [B,{cfg.dim_t},{cfg.dim_f},4] {self.dtype}, got "
                             f"{tuple(spek.shape)} {spek.dtype}")
        spek = spek.contiguous()
        out = torch.empty_like(spek)

        def run(handle):
            step = self.max_batch if self.max_batch > 0 else spek.shape[0]
            for b0 in range(0, spek.shape[0], step):
                x = spek[b0:b0 + step]
                y = out[b0:b0 + step]
                if denoise:                                   # 0.5*f(x) - 0.5*f(-x), mdxnet.py:168-173
                    self._forward(x, y, 1.0, 0.5, 0.0, handle)
                    self._forward(x, y, -1.0, -0.5, 1.0, handle)
                else:
                    self._forward(x, y, handle=handle)
        run(self.handle)
        # split-half contractions carry activations as IEEE-half pairs (|x| <= 65504).  A forward that met a larger activation raised the
        # network's range word: its result is discarded and the same windows run again on the exact f32 MFMA kernels (same weights, the
        # twin network is built on first need) -- slower, never wrong, never on the CPU.
        if self.contraction == "split" and self.range_exceeded():
            if self._exact is None:
                import logging
                logging.getLogger(__name__).warning("TDFNet: an activation left the half range (|x| > 65504) -- this and every later such batch "
                                                    "runs on the exact float32 kernels (contraction='exact' avoids the double work)")
                self._exact = self._create(self._table, 0)
            run(self._exact)
        return out

    def forward_pcm(self, plan, pcm: torch.Tensor, ch_stride: int, chunk_stride: int, n_chunks: int, pcm_offset: int = 0,
                    zero_low_bins: int = 0) -> Optional[torch.Tensor]:
        """STFT of ``n_chunks`` chunks framed out of the flat float32 buffer ``pcm`` (as StftPlan.stft_strided) + the network, with the STFT
        and the network's first 1x1 convolution fused into one kernel (alsep_net_forward_pcm: no spectrogram in HBM; bit-identical to
        stft_strided + forward_nhwc).  -> [n_chunks, dim_t, dim_f, 4] in the model dtype, or None when this (plan, network) pair has no
        fused kernel (float32 networks, other n_fft / g): the caller then runs the two steps."""
        if self.dtype == torch.float32 or pcm.dtype != torch.float32:
            return None
        cfg = self.cfg
        if (plan.n_fft, plan.hop, plan.dim_f, plan.dim_t) != (cfg.n_fft, cfg.hop, cfg.dim_f, cfg.dim_t):
            raise AlsepError("forward_pcm: the plan's geometry is not the network's")
        need = ch_stride + (n_chunks - 1) * chunk_stride + plan.chunk_size + pcm_offset
        if n_chunks > 0 and pcm.numel() < need:
            raise AlsepError(f"pcm buffer too small: {pcm.numel()} < {need}")
        out = self.ctx.empty((n_chunks, cfg.dim_t, cfg.dim_f, 4), self.dtype)
        step = self.max_batch if self.max_batch > 0 else n_chunks
        for b0 in range(0, n_chunks, step):
            nb = min(step, n_chunks - b0)
            ws = self.workspace(nb)
            base = (ws.data_ptr() + 255) & ~255
            rc = self.ctx.lib.alsep_net_forward_pcm(self.ctx.handle, self.handle, plan.handle,
                                                    C.c_void_p(_lib.ptr(pcm) + 4 * (pcm_offset + b0 * chunk_stride)), ch_stride, chunk_stride,
                                                    _lib.ptr(out[b0:b0 + nb]), nb, C.c_void_p(base), ws.numel() - 256, 1.0, 1.0, 0.0, int(zero_low_bins))
            if rc == -4 and b0 == 0:                       # ALSEP_ERR_STATE: no fused kernel for this pair
                return None
            self.ctx.check(rc, "alsep_net_forward_pcm")
        return out

    def run(self, _names, feed):
        """ORT-session surface (mdxnet.py:170-176, patch_separate.py:52): reference layout in/out."""
        from .mdx import StftPlan
        spek = torch.as_tensor(feed["input"]).to(self.ctx.device)    # ORT sessions are fed host arrays (mdxnet.py:170-176)
        plan = self._plan()
        x = plan.convert(spek.contiguous().to(self.dtype), _lib.LAYOUT_REF)
        y = self.forward_nhwc(x)
        return [plan.convert(y, _lib.LAYOUT_NHWC).float()]

    def _plan(self):
        from .mdx import StftPlan
        if not hasattr(self, "_plan_obj"):
            self._plan_obj = StftPlan(self.ctx, self.cfg.n_fft, self.cfg.hop, self.cfg.dim_f, self.cfg.dim_t)
        return self._plan_obj

    def __call__(self, spek: torch.Tensor) -> torch.Tensor:
        """model_run(spek) of the reference seam (patch_separate.py:52)."""
        return self.run(None, {"input": spek})[0]
