"""Ensemble post-ops of the Separate engine on the GPU (HBM-bound element-wise / reductions),
mirroring ``EnsembleDemucsMDXMusicSeparationModel`` helpers of the reference
(modules/separator/stem_separator.py): ``_blend_tracks`` (:241-262), ``_residual_subtract``
(:173-239) and the post-blend de-bleed (:415-456).  Tensors are float32 ``[C, N]`` on the device;
all arithmetic runs in libalsep.so (alsep_axpby / peak_abs / scale_by_device / dot3 /
xcorr_window / shift_subtract); only scalars (a peak, three dot products, 2*529+1 correlation
values) cross to the host to take the reference's branches.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import AlsepError, Context

_DOT_WORDS = 4 + 3 * 1024


def _flat(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise AlsepError("ensemble ops take float32 tensors")
    return t if t.is_contiguous() else t.contiguous()


def peak_abs(ctx: Context, x: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """max|x| as a 1-element device tensor."""
    x = _flat(x)
    out = ctx.empty((1,)) if out is None else out
    ctx.check(ctx.lib.alsep_peak_abs(ctx.handle, _lib.ptr(x), x.numel(), _lib.ptr(out)), "alsep_peak_abs")
    return out


def dot3(ctx: Context, a: torch.Tensor, b: torch.Tensor) -> Tuple[float, float, float]:
    """(<a,b>, <a,a>, <b,b>) accumulated in float64 on the device."""
    a, b = _flat(a), _flat(b)
    if a.numel() != b.numel():
        raise AlsepError("dot3: size mismatch")
    buf = ctx.zeros((_DOT_WORDS,), torch.float64)
    ctx.check(ctx.lib.alsep_dot3(ctx.handle, _lib.ptr(a), _lib.ptr(b), a.numel(), _lib.ptr(buf)), "alsep_dot3")
    r = buf[:3].cpu()
    return float(r[0]), float(r[1]), float(r[2])


def blend_tracks(ctx: Context, tracks: Sequence[torch.Tensor], weights: Sequence[float]) -> torch.Tensor:
    """stem_separator.py:241-262: sum_i w_i x_i / max(sum w, 1e-6) over zero-padded tracks, then
    divide by the global peak (result peak is exactly 1.0 unless all-zero)."""
    max_len = max(t.shape[-1] for t in tracks)
    ch = tracks[0].shape[0]
    acc = ctx.zeros((ch, max_len))
    total_w = max(sum(float(w) for w in weights), 1e-6)     # the reference sums ALL given weights (:252)
    for i, t in enumerate(tracks):
        w = float(weights[i]) if i < len(weights) else 1.0
        t = _flat(t)
        if t.shape[-1] != max_len:                          # zero-pad to the longest (:253-256)
            padded = ctx.zeros((ch, max_len))
            padded[:, : t.shape[-1]] = t
            t = padded
        ctx.check(ctx.lib.alsep_axpby(ctx.handle, w / total_w, _lib.ptr(t), 1.0, _lib.ptr(acc), t.numel()), "alsep_axpby")
    pk = peak_abs(ctx, acc)
    # combined /= peak if peak > 0: floor 0 keeps an all-zero blend at zero (0 * inf guarded in-kernel by floor)
    ctx.check(ctx.lib.alsep_scale_by_device(ctx.handle, _lib.ptr(acc), acc.numel(), 1.0, _lib.ptr(pk), 1e-30),
              "alsep_scale_by_device")
    return acc


def best_lag(ctx: Context, ref: torch.Tensor, sig: torch.Tensor, max_shift: int, probe: int = 44100) -> int:
    """argmax over lag in [-max_shift, +max_shift] of sum_n ref[n+lag]*sig[n] on the first ``probe``
    samples (np.correlate(...,'full') centre window, stem_separator.py:216-224)."""
    n = min(ref.numel(), sig.numel(), probe)
    if n <= max_shift:                                     # probe shorter than the search window
        return 0
    corr = ctx.zeros((2 * max_shift + 1,), torch.float64)
    ctx.check(ctx.lib.alsep_xcorr_window(ctx.handle, _lib.ptr(_flat(ref)), _lib.ptr(_flat(sig)), n, max_shift,
                                         _lib.ptr(corr)), "alsep_xcorr_window")
    return int(torch.argmax(corr.cpu())) - max_shift


def residual_subtract(ctx: Context, base: torch.Tensor, component: torch.Tensor, sr: int,
                      max_shift_ms: float = 12.0, return_params: bool = False):
    """stem_separator.py:173-239 on [C,N] device tensors."""
    if base.dim() == 1:
        base = torch.stack([base, base])
    if component.dim() == 1:
        component = torch.stack([component, component])
    max_shift = max(int((max_shift_ms / 1000.0) * float(sr)), 0)
    n = min(base.shape[-1], component.shape[-1])
    residual = base.clone()
    params: List[Tuple[int, float]] = []
    for ch in range(base.shape[0]):
        ref = base[ch, :n].contiguous()
        sig = component[ch, :n].contiguous()
        lag = best_lag(ctx, ref, sig, max_shift) if (max_shift > 0 and n > 0) else 0
        # gain: <ref, shift(sig)> / (<shift(sig), shift(sig)> + 1e-8), clipped to [0, 1.25]
        if lag >= 0:
            a, b = ref[lag:], sig[: n - lag]
        else:
            a, b = ref[: n + lag], sig[-lag:]
        if a.numel() > 0:
            ab, _, bb = dot3(ctx, a.contiguous(), b.contiguous())
        else:
            ab, bb = 0.0, 0.0
        alpha = min(max(ab / (bb + 1e-8), 0.0), 1.25)
        out = ctx.empty((n,))
        ctx.check(ctx.lib.alsep_shift_subtract(ctx.handle, _lib.ptr(ref), _lib.ptr(sig), n, lag, alpha, _lib.ptr(out)),
                  "alsep_shift_subtract")
        residual[ch, :n] = out
        params.append((lag, alpha))
    return (residual, params) if return_params else residual


def cosine_abs(ctx: Context, a: torch.Tensor, b: torch.Tensor) -> float:
    """stem_separator.py:430-434."""
    ab, aa, bb = dot3(ctx, a.contiguous().reshape(-1), b.contiguous().reshape(-1))
    return abs(ab) / ((aa ** 0.5) * (bb ** 0.5) + 1e-8)


def debleed(ctx: Context, mix: torch.Tensor, vocals: torch.Tensor, instrumental: torch.Tensor, sr: int,
            residual_blend: float = 0.4) -> Tuple[torch.Tensor, bool]:
    """stem_separator.py:415-456: blend (mix - aligned gain-matched vocals) into the instrumental when
    that lowers its correlation with the vocals; silent-instrumental fallback."""
    accepted = False
    resid = residual_subtract(ctx, mix, vocals, sr)
    m = min(resid.shape[-1], instrumental.shape[-1])
    resid_m = resid[:, :m].contiguous()
    inst = instrumental[:, :m].contiguous()
    voc = vocals[:, :m].contiguous()
    out = instrumental
    if cosine_abs(ctx, resid_m, voc) + 1e-6 < cosine_abs(ctx, inst, voc) - 0.01:
        b = min(max(float(residual_blend), 0.0), 1.0)
        ref = inst.clone()
        ctx.check(ctx.lib.alsep_axpby(ctx.handle, b, _lib.ptr(resid_m), 1.0 - b, _lib.ptr(ref), ref.numel()), "alsep_axpby")
        pk = float(peak_abs(ctx, ref).cpu())
        if pk > 0.99:
            z = ctx.zeros(ref.shape)
            ctx.check(ctx.lib.alsep_axpby(ctx.handle, 0.99 / pk, _lib.ptr(ref), 0.0, _lib.ptr(z), ref.numel()), "alsep_axpby")
            ref = z
        out, accepted = ref, True
    if float(peak_abs(ctx, out).cpu()) < 1e-6:
        resid = residual_subtract(ctx, mix, vocals, sr)
        pk = float(peak_abs(ctx, resid).cpu())
        if pk > 1.0:
            z = ctx.zeros(resid.shape)
            ctx.check(ctx.lib.alsep_axpby(ctx.handle, 1.0 / pk, _lib.ptr(resid), 0.0, _lib.ptr(z), resid.numel()), "alsep_axpby")
            resid = z
        out = resid
    return out, accepted


def resample(ctx: Context, x: torch.Tensor, sr_in: int, sr_out: int, zeros: int = 32, rolloff: float = 0.95, beta: float = 12.0) -> torch.Tensor:
    """[C, N] at sr_in -> [C, ceil(N * sr_out / sr_in)] at sr_out on the device (the resampling librosa.load(sr=44100) applies to a
    non-44.1 kHz input, stem_separator.py:865; own Kaiser-windowed sinc, see alsep_resample)."""
    if sr_in == sr_out:
        return x
    x = _flat(x)
    rows, n_in = x.shape
    n_out = -(-n_in * sr_out // sr_in)
    y = ctx.empty((rows, n_out))
    ctx.check(ctx.lib.alsep_resample(ctx.handle, _lib.ptr(x), _lib.ptr(y), rows, n_in, n_out, sr_in, sr_out, zeros, rolloff, beta),
              "alsep_resample")
    return y


_POLY_TAPS: dict = {}


def _poly_filter(up: int, down: int) -> torch.Tensor:
    """scipy.signal.resample_poly's default filter: firwin(2 half_len + 1, 1 / max_rate, window=("kaiser", 5.0)) with half_len =
    10 max_rate -- cutoff sinc(cutoff m) under a symmetric Kaiser window, scaled to unit gain at DC -- cast to float32 (the data's type)
    and multiplied by ``up`` there.  Built in float64 on the host."""
    key = (up, down)
    if key not in _POLY_TAPS:
        max_rate = max(up, down)
        half_len = 10 * max_rate
        m = torch.arange(-half_len, half_len + 1, dtype=torch.float64)
        cutoff = 1.0 / max_rate
        h = cutoff * torch.special.sinc(cutoff * m)
        h = h * torch.kaiser_window(2 * half_len + 1, periodic=False, beta=5.0, dtype=torch.float64)
        h = h / h.sum()
        _POLY_TAPS[key] = (h.to(torch.float32) * float(up)), half_len
    return _POLY_TAPS[key]


def resample_poly(ctx: Context, x: torch.Tensor, sr_in: int, sr_out: int) -> torch.Tensor:
    """librosa.resample(res_type="polyphase") = scipy.signal.resample_poly(x, sr_out // g, sr_in // g) on the device
    (``alsep_resample_poly``): [C, N] -> [C, ceil(N up / down)]."""
    if sr_in == sr_out:
        return x
    g = math.gcd(int(sr_in), int(sr_out))
    up, down = int(sr_out) // g, int(sr_in) // g
    taps, half_len = _poly_filter(up, down)
    x = _flat(x)
    rows, n_in = x.shape
    n_out = -(-n_in * up // down)
    n_pre_pad = down - half_len % down
    n_pre_remove = (half_len + n_pre_pad) // down
    y = ctx.empty((rows, n_out))
    dkey = (up, down, str(ctx.device))
    if dkey not in _POLY_TAPS:                              # the filter on this device, uploaded once
        _POLY_TAPS[dkey] = taps.to(ctx.device)
    t = _POLY_TAPS[dkey]
    ctx.check(ctx.lib.alsep_resample_poly(ctx.handle, _lib.ptr(x), _lib.ptr(y), rows, n_in, n_out, up, down, _lib.ptr(t), t.numel(),
                                          n_pre_pad, n_pre_remove), "alsep_resample_poly")
    return y


def resample_fft(ctx: Context, x: torch.Tensor, sr_in: int, sr_out: int) -> torch.Tensor:
    """librosa.resample(res_type="scipy") = scipy.signal.resample(x, ceil(N sr_out / sr_in)) on the device (``alsep_resample_fft``)."""
    if sr_in == sr_out:
        return x
    x = _flat(x)
    rows, n_in = x.shape
    n_out = -(-n_in * int(sr_out) // int(sr_in))
    nbytes = ctx.lib.alsep_resample_fft_workspace_bytes(n_in, n_out)
    if nbytes < 0:
        raise AlsepError(f"resample_fft: {n_in} -> {n_out} samples is beyond the transform limit (2^27 points)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=ctx.device)
    y = ctx.empty((rows, n_out))
    ctx.check(ctx.lib.alsep_resample_fft(ctx.handle, _lib.ptr(x), n_in, _lib.ptr(y), n_out, rows, n_in, n_out, _lib.ptr(ws), nbytes),
              "alsep_resample_fft")
    return y


def normalize(ctx: Context, x: torch.Tensor, max_peak: float = 0.9) -> torch.Tensor:
    """audio-separator's ``spec_utils.normalize`` (upstream, uncited -- UNPINNED): a signal whose peak exceeds ``max_peak`` is scaled
    down to it, a quieter one is left alone.  In place on the device (x *= max_peak / max(peak, max_peak)); returns x."""
    x = _flat(x)
    pk = peak_abs(ctx, x)
    ctx.check(ctx.lib.alsep_scale_by_device(ctx.handle, _lib.ptr(x), x.numel(), float(max_peak), _lib.ptr(pk), float(max_peak)),
              "alsep_scale_by_device")
    return x


_INVERT_PLANS: dict = {}


def invert_stem(ctx: Context, mixture: torch.Tensor, stem: torch.Tensor, n_fft: int = 2048, hop: int = 1024) -> torch.Tensor:
    """audio-separator's ``spec_utils.invert_stem`` (upstream, uncited -- UNPINNED), the secondary stem of an MDX model when the engine is
    built with ``invert_using_spec=True`` as AudioLab builds it (stem_separator.py:104): both signals go through an n_fft 2048 / hop 1024
    STFT, the spectrograms are subtracted, and the difference is transformed back.  [2, N] float32 device tensors -> [2, N].
    One whole-track frame set per signal (centre-padded by reflection, as librosa): the track is zero-padded to a multiple of the hop for
    the transform and cut back (upstream's inverse transform stops at the last full hop and pads the remainder with zeros)."""
    from .mdx import StftPlan
    a, b = _flat(mixture), _flat(stem)
    if a.shape != b.shape or a.dim() != 2 or a.shape[0] != 2:
        raise AlsepError("invert_stem expects two [2, N] tensors of the same length")
    n = a.shape[1]
    frames = -(-n // hop) + 1
    key = (id(ctx), n_fft, hop, frames)
    if key not in _INVERT_PLANS:
        _INVERT_PLANS.clear()                                # one track length at a time: the tables of a 13 000-frame plan are not small
        _INVERT_PLANS[key] = StftPlan(ctx, n_fft, hop, n_fft // 2 + 1, frames)
    plan = _INVERT_PLANS[key]
    length = plan.chunk_size                                 # hop * (frames - 1) >= n

    def spec_of(x):
        buf = ctx.zeros((2, length))
        buf[:, :n] = x
        return plan.stft_strided(buf, length, 2 * length, 1, torch.float32, _lib.LAYOUT_REF)
    sa, sb = spec_of(a), spec_of(b)
    ctx.check(ctx.lib.alsep_axpby(ctx.handle, -1.0, _lib.ptr(sb), 1.0, _lib.ptr(sa), sa.numel()), "alsep_axpby")      # sa -= sb
    out = ctx.empty((2, length))
    plan.istft_strided(sa, _lib.LAYOUT_REF, out, length, 2 * length, 0, length, length)
    return out[:, :n].contiguous()
