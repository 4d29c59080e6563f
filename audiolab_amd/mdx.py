"""MDX-Net front end / back end / chunker on the GPU, mirroring the reference's in-tree runner
``modules/rvc/infer/modules/uvr5/mdxnet.py`` (ConvTDFNetTrim :15-75, Predictor :90-197) --
same class names, constructor arguments and result shapes -- with every transform executed by
the HIP kernels of libalsep.so.

Two ways in:
  * the *reference-shaped* API (``ConvTDFNetTrim.stft/istft``, ``Predictor.demix`` with a
    ``model.run(None, {"input": spek})`` callable) keeps the reference's tensor layout
    ``[B,4,dim_f,dim_t]`` so parity tests read like the reference;
  * the *fused* path (``Predictor.demix`` with a :class:`audiolab_amd.tdfnet.TDFNet`) frames the
    zero-padded mix in place (no chunk copies), keeps the spectrogram channels-last on the
    device between STFT, network and iSTFT, and lets the iSTFT store straight into the stitched
    track (trim / pad / margins folded into the store).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _lib
from ._lib import AlsepError, Context


class StftPlan:
    """alsep_plan: geometry of ConvTDFNetTrim.__init__ (mdxnet.py:15-39) + device tables."""

    def __init__(self, ctx: Context, n_fft: int, hop: int, dim_f: int, dim_t: int):
        self.ctx = ctx
        self.n_fft, self.hop, self.dim_f, self.dim_t = n_fft, hop, dim_f, dim_t
        self.n_bins = n_fft // 2 + 1
        self.chunk_size = hop * (dim_t - 1)
        self.trim = n_fft // 2
        self.gen_size = self.chunk_size - 2 * self.trim
        h = C.c_void_p()
        ctx.check(ctx.lib.alsep_plan_create(ctx.handle, n_fft, hop, dim_f, dim_t, C.byref(h)), "alsep_plan_create")
        self.handle = h

    def __del__(self):
        try:
            if getattr(self, "handle", None) and self.ctx.handle:
                self.ctx.lib.alsep_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # -- raw calls ---------------------------------------------------------------------------
    def spec_shape(self, n_chunks: int, layout: int) -> Tuple[int, ...]:
        if layout == _lib.LAYOUT_REF:
            return (n_chunks, 4, self.dim_f, self.dim_t)
        return (n_chunks, self.dim_t, self.dim_f, 4)

    def stft_strided(self, pcm: torch.Tensor, ch_stride: int, chunk_stride: int, n_chunks: int,
                     dtype: torch.dtype = torch.float32, layout: int = _lib.LAYOUT_NHWC,
                     out: Optional[torch.Tensor] = None, pcm_offset: int = 0) -> torch.Tensor:
        """Frames ``n_chunks`` chunks out of the flat float32 buffer ``pcm`` (see alsep_stft)."""
        if pcm.dtype != torch.float32:
            raise AlsepError("pcm must be float32")
        need = ch_stride + (n_chunks - 1) * chunk_stride + self.chunk_size + pcm_offset
        if n_chunks > 0 and pcm.numel() < need:
            raise AlsepError(f"pcm buffer too small: {pcm.numel()} < {need}")
        if out is None:
            out = self.ctx.empty(self.spec_shape(n_chunks, layout), dtype)
        elif tuple(out.shape) != self.spec_shape(n_chunks, layout) or out.dtype != dtype:
            raise AlsepError("stft: bad output tensor")
        base = _lib.ptr(pcm) + 4 * pcm_offset
        self.ctx.check(self.ctx.lib.alsep_stft(self.ctx.handle, self.handle, C.c_void_p(base), ch_stride, chunk_stride,
                                               n_chunks, _lib.ptr(out), _lib.dtype_code(dtype), layout), "alsep_stft")
        return out

    def istft_strided(self, spec: torch.Tensor, layout: int, out: torch.Tensor, out_ch_stride: int,
                      out_chunk_stride: int, keep_lo: int, keep_hi: int, out_limit: int, out_offset: int = 0) -> None:
        n_chunks = spec.shape[0]
        if tuple(spec.shape) != self.spec_shape(n_chunks, layout):
            raise AlsepError(f"istft: spec shape {tuple(spec.shape)} != {self.spec_shape(n_chunks, layout)}")
        if out.dtype != torch.float32:
            raise AlsepError("istft output must be float32")
        if out.numel() < out_offset + out_ch_stride + out_limit:
            raise AlsepError("istft: output buffer too small")
        base = _lib.ptr(out) + 4 * out_offset
        self.ctx.check(self.ctx.lib.alsep_istft(self.ctx.handle, self.handle, _lib.ptr(spec), _lib.dtype_code(spec.dtype),
                                                layout, n_chunks, C.c_void_p(base), out_ch_stride, out_chunk_stride,
                                                keep_lo, keep_hi, out_limit), "alsep_istft")

    def convert(self, spec: torch.Tensor, src_layout: int) -> torch.Tensor:
        n = spec.shape[0]
        dst_layout = _lib.LAYOUT_NHWC if src_layout == _lib.LAYOUT_REF else _lib.LAYOUT_REF
        if tuple(spec.shape) != self.spec_shape(n, src_layout):
            raise AlsepError("convert: bad spec shape")
        out = self.ctx.empty(self.spec_shape(n, dst_layout), spec.dtype)
        self.ctx.check(self.ctx.lib.alsep_spec_convert(self.ctx.handle, _lib.ptr(spec), _lib.ptr(out),
                                                       _lib.dtype_code(spec.dtype), src_layout, n, self.dim_f,
                                                       self.dim_t), "alsep_spec_convert")
        return out


class ConvTDFNetTrim:
    """Drop-in for the reference class of the same name (mdxnet.py:15-75): STFT geometry and the
    ``stft`` / ``istft`` pair, tensors in the reference layout, computed by the HIP kernels."""

    def __init__(self, device, model_name, target_name, L, dim_f, dim_t, n_fft, hop=1024, ctx: Optional[Context] = None):
        self.ctx = ctx if ctx is not None else _lib.default_context(device)
        self.dim_f = dim_f
        self.dim_t = 2 ** dim_t                          # mdxnet.py:22
        self.n_fft = n_fft
        self.hop = hop
        self.n_bins = self.n_fft // 2 + 1
        self.chunk_size = hop * (self.dim_t - 1)
        self.target_name = target_name
        self.blender = "blender" in model_name
        self.dim_c = 4
        self.n = L // 2
        if target_name == "*":
            raise AlsepError("multi-target ('*') ConvTDFNetTrim is not supported")
        self.plan = StftPlan(self.ctx, n_fft, hop, dim_f, self.dim_t)

    def stft(self, x: torch.Tensor) -> torch.Tensor:
        """[B,2,chunk] (any shape reshapable to [-1,chunk] pairs) -> [B,4,dim_f,dim_t] float32."""
        x = x.reshape(-1, 2, self.chunk_size).contiguous().float()
        b = x.shape[0]
        return self.plan.stft_strided(x, self.chunk_size, 2 * self.chunk_size, b, torch.float32, _lib.LAYOUT_REF)

    def istft(self, x: torch.Tensor, freq_pad=None) -> torch.Tensor:
        """[B,4,dim_f,dim_t] -> [B,2,chunk]; bins >= dim_f are zero (mdxnet.py:59-64)."""
        if freq_pad is not None:
            raise AlsepError("custom freq_pad is not supported; bins >= dim_f are zero")
        x = x.contiguous()
        if x.dtype not in (torch.float32, torch.bfloat16, torch.float16):
            x = x.float()
        b = x.shape[0]
        out = self.ctx.empty((b, 2, self.chunk_size), torch.float32)
        self.plan.istft_strided(x, _lib.LAYOUT_REF, out, self.chunk_size, 2 * self.chunk_size, 0, self.chunk_size,
                                (b - 1) * 2 * self.chunk_size + self.chunk_size)
        return out


def get_models(device, dim_f, dim_t, n_fft, ctx: Optional[Context] = None):
    """mdxnet.py:78-87."""
    return ConvTDFNetTrim(device=device, model_name="Conv-TDF", target_name="vocals", L=11, dim_f=dim_f,
                          dim_t=dim_t, n_fft=n_fft, ctx=ctx)


class Predictor:
    """Reference ``Predictor`` (mdxnet.py:90-197) on the GPU.

    ``args`` needs ``margin, chunks, denoise, dim_f, dim_t, n_fft`` (as MDXNetDereverb supplies,
    mdxnet.py:241-252).  ``model`` is either
      * a :class:`audiolab_amd.tdfnet.TDFNet` (fused channels-last path), or
      * any object with ``run(None, {"input": tensor}) -> [tensor]`` taking/returning the
        reference layout ``[B,4,dim_f,dim_t]`` on the device (the ORT-session seam, :170-176).
    """

    def __init__(self, args, model, ctx: Optional[Context] = None, hop: int = 1024, max_batch: int = 0,
                 group=None, sharded: bool = False):
        """``sharded=True`` splits every segment's model windows over the ranks of ``group``
        (torch.distributed; default group if None) and all-gathers the stem segments."""
        self.group = group
        self.sharded = sharded
        self.args = args
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        self.model_ = ConvTDFNetTrim(self.ctx.device, "Conv-TDF", "vocals", 11, args.dim_f, args.dim_t, args.n_fft,
                                     hop=hop, ctx=self.ctx)
        self.model = model
        self.max_batch = max_batch                      # chunks per network launch (0 = whole segment)

    # -- outer segmentation, mdxnet.py:109-141 -------------------------------------------------
    def segments(self, samples: int) -> Tuple[List[Tuple[int, int, int]], int]:
        """[(skip, start, end)], margin -- the index arithmetic of demix (:109-133)."""
        margin = self.args.margin
        chunk_size = self.args.chunks * 44100
        assert not margin == 0, "margin cannot be zero!"
        if margin > chunk_size:
            margin = chunk_size
        if self.args.chunks == 0 or samples < chunk_size:
            chunk_size = samples
        segs = []
        counter = -1
        for skip in range(0, samples, chunk_size):
            counter += 1
            s_margin = 0 if counter == 0 else margin
            end = min(skip + chunk_size + margin, samples)
            segs.append((skip, skip - s_margin, end))
            if end == samples:
                break
        return segs, margin

    def demix(self, mix: torch.Tensor, defer: bool = False):
        """mix [2,N] float32 on the device -> sources [1,2,N] (mdxnet.py:109-141).
        ``defer=True`` (sharded runs): returns a callable; the all-gathers of the segments stay in
        flight (RCCL stream) while the caller launches the next model, and the callable assembles."""
        if mix.dim() != 2 or mix.shape[0] != 2:
            raise AlsepError("demix expects a [2,N] stereo tensor")
        mix = mix.contiguous().float()
        samples = mix.shape[-1]
        segs, margin = self.segments(samples)
        pending = [self.demix_segment(mix[:, start:end], async_gather=defer and self.sharded) for (_s, start, end) in segs]

        def assemble() -> torch.Tensor:
            return self._assemble(pending, segs, margin, samples)
        return assemble if defer else assemble()

    def _assemble(self, pending, segs, margin, samples) -> torch.Tensor:
        out = self.ctx.empty((1, 2, samples), torch.float32)
        pos = 0
        for idx, (skip, start, end) in enumerate(segs):
            first, last = idx == 0, idx == len(segs) - 1
            seg = pending[idx].result() if hasattr(pending[idx], "result") else pending[idx]   # [2, end-start]
            lo = 0 if first else margin                            # :185
            hi = seg.shape[-1] if (last or margin == 0) else seg.shape[-1] - margin   # :186-188
            n = hi - lo
            out[0, :, pos:pos + n] = seg[:, lo:hi]
            pos += n
        if pos != samples:
            raise AlsepError(f"demix stitched {pos} samples, expected {samples}")
        return out

    # -- inner framing + inference + stitch of one segment, mdxnet.py:147-183 -------------------
    def demix_segment(self, cmix: torch.Tensor, async_gather: bool = False):
        m = self.model_
        plan = m.plan
        n_sample = cmix.shape[1]
        trim, gen = plan.trim, plan.gen_size
        pad = gen - n_sample % gen                                  # :153
        n_win = (n_sample + pad) // gen                             # windows i = 0, gen, ... < n_sample+pad
        total = trim + n_sample + pad + trim
        mix_p = self.ctx.zeros((2, total), torch.float32)           # :154-156
        mix_p[:, trim:trim + n_sample] = cmix
        w_lo, w_hi, s_lo, s_hi = 0, n_win, 0, n_sample
        if self.sharded:
            import torch.distributed as tdist
            from . import dist as adist
            world, rank = tdist.get_world_size(self.group), tdist.get_rank(self.group)
            w_lo, w_hi = adist.window_range(n_win, world, rank)
            s_lo, s_hi = adist.sample_range(n_win, gen, n_sample, world, rank)
        n_local = s_hi - s_lo
        seg_out = self.ctx.empty((2, max(n_local, 1)), torch.float32)
        if n_local > 0:
            self._run_windows(mix_p, total, seg_out, n_local, w_lo, w_hi, s_lo, n_sample)
        seg_out = seg_out[:, :n_local]
        if self.sharded:
            seg_out = adist.all_gather_segments(seg_out.contiguous(), n_win, gen, n_sample, self.group, async_op=async_gather)
        return seg_out

    def _run_windows(self, mix_p, total, seg_out, n_local, w_lo, w_hi, s_lo, n_sample) -> None:
        """STFT -> network -> iSTFT for windows [w_lo, w_hi); window w lands at sample w*gen - s_lo
        of ``seg_out`` [2, >= n_local]; samples beyond n_sample (the trailing pad, :183) are dropped."""
        m = self.model_
        plan = m.plan
        trim, gen = plan.trim, plan.gen_size
        ld = seg_out.shape[1]
        step = self.max_batch if self.max_batch > 0 else (w_hi - w_lo)
        for w0 in range(w_lo, w_hi, step):
            nb = min(step, w_hi - w0)
            limit = min(n_sample, s_lo + n_local) - w0 * gen
            if hasattr(self.model, "forward_nhwc"):
                net = self.model
                pred = None
                if not self.args.denoise and hasattr(net, "forward_pcm"):     # STFT + first layer as one kernel (half-precision networks)
                    pred = net.forward_pcm(plan, mix_p, total, gen, nb, pcm_offset=w0 * gen)
                if pred is None:
                    spek = plan.stft_strided(mix_p, total, gen, nb, net.dtype, _lib.LAYOUT_NHWC, pcm_offset=w0 * gen)
                    pred = net.forward_nhwc(spek, denoise=bool(self.args.denoise))
                plan.istft_strided(pred, _lib.LAYOUT_NHWC, seg_out, ld, gen, trim, plan.chunk_size - trim,
                                   limit, out_offset=w0 * gen - s_lo)
            else:
                spek = plan.stft_strided(mix_p, total, gen, nb, torch.float32, _lib.LAYOUT_REF, pcm_offset=w0 * gen)
                if self.args.denoise:                               # :168-173
                    pred = -self.model.run(None, {"input": -spek})[0] * 0.5 + self.model.run(None, {"input": spek})[0] * 0.5
                else:
                    pred = self.model.run(None, {"input": spek})[0]
                pred = torch.as_tensor(pred, device=self.ctx.device).contiguous()
                plan.istft_strided(pred, _lib.LAYOUT_REF, seg_out, ld, gen, trim, plan.chunk_size - trim,
                                   limit, out_offset=w0 * gen - s_lo)


class OlaRunner:
    """Hann-window overlap-add runner: the chunker ``Separator.separate`` executes today for MDX models
    (third-party ``MDXSeparator.demix`` / ``run_model``, reached at stem_separator.py:281; PARITY UNPINNED --
    restated in oracle/mdx_oracle.py ``demix_ola``).  Windows start every ``(1-overlap)*chunk`` samples of
    the padded mixture, the lowest ``zero_low_bins`` bins are zeroed before the network, outputs are
    weighted by np.hanning, summed, divided by the summed weights, and scaled by ``compensate``."""

    def __init__(self, net, ctx: Optional[Context] = None, overlap: float = 0.25, zero_low_bins: int = 3,
                 compensate: float = 1.0, denoise: bool = False, max_batch: int = 8, sharded: bool = False, group=None):
        """``sharded=True``: the chunks are split into contiguous ranges over the ranks of ``group`` (torch.distributed); every
        rank accumulates the weighted sums of its chunks, ONE all-reduce adds the parts (chunks of neighbouring ranks overlap at
        the shard seams), the division by the summed window weights happens afterwards on every rank."""
        self.sharded, self.group = sharded, group
        self.net = net
        self.ctx = ctx if ctx is not None else net.ctx
        cfg = net.cfg
        self.plan = StftPlan(self.ctx, cfg.n_fft, cfg.hop, cfg.dim_f, cfg.dim_t)
        self.overlap, self.zero_low_bins, self.compensate, self.denoise = overlap, zero_low_bins, compensate, denoise
        self.max_batch = max_batch

    def demix(self, mix: torch.Tensor, match_mix: bool = False, in_scale: float = 1.0) -> torch.Tensor:
        """mix [2,N] float32 (scaled by ``in_scale`` on the device: the engine's input normalisation) -> [2,N] on the device.  ``match_mix=True`` is the package's ``is_match_mix`` pass: the same framing
        without the network at overlap 0.02 -- the mixture as the model path sees it (bins < 3 and >= dim_f removed), which
        ``invert_using_spec`` subtracts the primary stem from.

        Sharded (``sharded=True``): rank r owns a contiguous range of chunks.  It uploads / frames only the samples those
        chunks cover (``mix`` may live on the host: a 60-minute 8-channel programme need not sit on every GPU), accumulates the
        windowed sums over ITS span, and owns the output samples from its first chunk's start to the next rank's.  Its chunks reach up
        to ``chunk - step`` samples into the following ranks' ranges: those seam sums (a few MB) are exchanged by one small all-gather
        and added locally, each rank divides its own range, and ONE all-gather of the finished stem segments assembles the track."""
        plan, ctx = self.plan, self.ctx
        n = mix.shape[-1]
        chunk, trim, gen = plan.chunk_size, plan.trim, plan.gen_size
        pad = gen + trim - (n % gen)
        total = trim + n + pad
        overlap = 0.02 if match_mix else self.overlap
        step = int((1 - overlap) * chunk)
        n_chunks = (total + step - 1) // step
        use_window = 1 if overlap != 0 else 0
        compensate = 1.0 if match_mix else float(self.compensate)
        c_lo, c_hi, world, rank = 0, n_chunks, 1, 0
        if self.sharded:
            import torch.distributed as tdist
            from . import dist as adist
            world, rank = tdist.get_world_size(self.group), tdist.get_rank(self.group)
            c_lo, c_hi = adist.window_range(n_chunks, world, rank)
        n_local = c_hi - c_lo
        # the padded mixture over this rank's chunks only: padded coordinates [a0, a0 + buf_len); chunks cut by the end see zeros
        a0 = c_lo * step
        buf_len = (max(n_local, 1) - 1) * step + chunk
        mixture = ctx.zeros((2, buf_len), torch.float32)
        s_lo, s_hi = max(a0 - trim, 0), min(a0 + buf_len - trim, n)           # the samples of `mix` under it
        if s_hi > s_lo:
            mixture[:, s_lo + trim - a0: s_hi + trim - a0] = mix[:, s_lo:s_hi].to(ctx.device, dtype=torch.float32)
            if in_scale != 1.0:
                ctx.check(ctx.lib.alsep_axpby(ctx.handle, 0.0, _lib.ptr(mixture), float(in_scale), _lib.ptr(mixture), mixture.numel()), "alsep_axpby")
        waves = ctx.empty((max(n_local, 1), 2, chunk), torch.float32)
        bstep = self.max_batch if self.max_batch > 0 else max(n_local, 1)
        for b0 in range(c_lo, c_hi, bstep):
            nb = min(bstep, c_hi - b0)
            pred = None
            if not match_mix and not self.denoise and hasattr(self.net, "forward_pcm"):   # STFT + zeroed low bins + first layer as one kernel
                pred = self.net.forward_pcm(plan, mixture, buf_len, step, nb, pcm_offset=(b0 - c_lo) * step, zero_low_bins=self.zero_low_bins)
            if pred is None:
                spek = plan.stft_strided(mixture, buf_len, step, nb, self.net.dtype, _lib.LAYOUT_NHWC, pcm_offset=(b0 - c_lo) * step)
                if self.zero_low_bins:
                    ctx.check(ctx.lib.alsep_zero_low_bins(ctx.handle, _lib.ptr(spek), _lib.dtype_code(spek.dtype), _lib.LAYOUT_NHWC,
                                                          nb, plan.dim_f, plan.dim_t, self.zero_low_bins), "alsep_zero_low_bins")
                pred = spek if match_mix else self.net.forward_nhwc(spek, denoise=self.denoise)
            plan.istft_strided(pred, _lib.LAYOUT_NHWC, waves, chunk, 2 * chunk, 0, chunk, (nb - 1) * 2 * chunk + chunk,
                               out_offset=(b0 - c_lo) * 2 * chunk)
        if not self.sharded:
            out = ctx.empty((2, n), torch.float32)
            ctx.check(ctx.lib.alsep_ola_combine(ctx.handle, _lib.ptr(waves), n_chunks, chunk, step, total, use_window,
                                                compensate, _lib.ptr(out), n, trim, n), "alsep_ola_combine")
            return out
        # own output ranges in padded coordinates, clipped to the kept region [trim, trim + n): rank q owns from its first chunk's start
        # (rank 0: from trim) to the next rank's (the last rank: to the end)
        bounds = [adist.window_range(n_chunks, world, q) for q in range(world)]
        starts = [min(max(b[0] * step, trim), trim + n) for b in bounds] + [trim + n]
        starts[0] = trim
        own_lo, own_hi = starts[rank], starts[rank + 1]
        seam = chunk - step                                                   # how far a rank's chunks reach beyond its own range

        def partial(p_lo, length):
            part = ctx.zeros((3, max(length, 1)), torch.float32)
            if n_local > 0 and length > 0:
                ctx.check(ctx.lib.alsep_ola_partial(ctx.handle, _lib.ptr(waves), c_lo, c_hi, chunk, step, total, use_window, _lib.ptr(part),
                                                    p_lo, length), "alsep_ola_partial")
            return part
        own = partial(own_lo, own_hi - own_lo)
        tail_lo = min(c_hi * step if n_local > 0 else own_hi, total)          # this rank's sums beyond its own range: [tail_lo, tail_lo + seam)
        tail_len = max(0, min(seam, total - tail_lo))
        tail = ctx.zeros((3, max(seam, 1)), torch.float32)
        if tail_len > 0 and n_local > 0:
            tail[:, :tail_len] = partial(tail_lo, tail_len)[:, :tail_len]
        tails = adist.all_gather_fixed(tail, self.group)                      # [world, 3, seam]
        for q in range(rank):                                                 # earlier ranks whose chunks reach into this rank's range
            q_lo = min(bounds[q][1] * step, total) if bounds[q][1] > bounds[q][0] else None
            if q_lo is None:
                continue
            lo, hi = max(q_lo, own_lo), min(q_lo + seam, own_hi)
            if hi > lo:
                own[:, lo - own_lo: hi - own_lo] += tails[q][:, lo - q_lo: hi - q_lo]
        seg = ctx.empty((2, max(own_hi - own_lo, 1)), torch.float32)
        if own_hi > own_lo:
            ctx.check(ctx.lib.alsep_ola_finish(ctx.handle, _lib.ptr(own), compensate, _lib.ptr(seg), seg.shape[1], own_hi - own_lo), "alsep_ola_finish")
        ranges = [(starts[q] - trim, starts[q + 1] - trim) for q in range(world)]
        return adist.all_gather_ranges(seg[:, : own_hi - own_lo].contiguous(), ranges, n, self.group)
