"""Multi-band front / back end of the VR models on the device.

Mirrors the in-tree reference runner /root/reference/modules/rvc/infer/modules/uvr5/vr.py:43-196 (``AudioPre._path_audio_``) and
its helpers modules/rvc/infer/lib/uvr5_pack/lib_v5/spec_utils.py (``wave_to_spectrogram`` :30-56, ``combine_spectrograms`` :95-125,
``cmb_spectrogram_to_wave`` :353-429, ``fft_lp_filter`` / ``fft_hp_filter`` :432-451, ``mirroring`` :453-470):

  wave [2, n] @ 44.1 kHz -- resampled down the band chain -- one STFT per band (librosa ``center=True``, zero padding, periodic Hann)
  -> bins [crop_start, crop_stop) of each band stacked into X [2, bins + 1, l], pre-filter gains on the top bins
  -> network on |X| / max|X| (``vrnet.vr_inference``)  -> y = pred * phase(X), v = X - y
  -> per output: mirrored high end, per band: crop back, low-pass / high-pass ramps, iSTFT, add, resample up the chain.

Everything between the input wave and the two output waves stays in HBM; the STFT / iSTFT are the library's FFT kernels (plans 320, 512,
640, 960); the band chain's resamplers are the two scipy routines librosa runs for the kinds the reference asks for ("polyphase" going
down, "scipy" going up: ``alsep_resample_poly`` / ``alsep_resample_fft``; the other two kinds in the reference's text, "sinc_fastest"
and the top band's "kaiser_fast", only ever see equal rates in the 4-band sets -- DESIGN section 6).  The librosa zero-padded framing is obtained from ``alsep_stft`` (which reflect-pads its chunk) by
placing the wave at an offset inside a zero buffer and reading the frames whose windows stay inside the zeros.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch

from . import _lib
from ._lib import AlsepError, Context
from .mdx import StftPlan
from .vrnet import VRNet, vr_inference

# lib_v5/modelparams/4band_v2.json, 4band_v3.json (hyper-parameters of the published models; both False: mid_side, reverse)
_BANDS_4 = {
    1: dict(sr=7350, hl=80, n_fft=640, crop_start=0, crop_stop=85, lpf_start=25, lpf_stop=53, res_type="polyphase"),
    2: dict(sr=7350, hl=80, n_fft=320, crop_start=4, crop_stop=87, hpf_start=25, hpf_stop=12, lpf_start=31, lpf_stop=62, res_type="polyphase"),
    3: dict(sr=14700, hl=160, n_fft=512, crop_start=17, crop_stop=216, hpf_start=48, hpf_stop=24, lpf_start=139, lpf_stop=210,
            res_type="polyphase"),
    4: dict(sr=44100, hl=480, n_fft=960, crop_start=78, crop_stop=383, hpf_start=130, hpf_stop=86, res_type="kaiser_fast"),
}
# the "_sn" sets: the same bands with the top band's channels converted ("convert_channels": "stereo_n", modelparams/4band_v2_sn.json:48).
# 4band_v3_sn (the BG-vocal model's set) is not in the reference tree: taken as 4band_v3 + the same line, as v2_sn is to v2 (inferred).
_BANDS_4_SN = {**_BANDS_4, 4: dict(_BANDS_4[4], convert_channels="stereo_n")}
MODEL_PARAMS: Dict[str, dict] = {
    "4band_v2": dict(bins=672, unstable_bins=8, reduction_bins=637, band=_BANDS_4, sr=44100, pre_filter_start=668, pre_filter_stop=672),
    "4band_v3": dict(bins=672, unstable_bins=8, reduction_bins=530, band=_BANDS_4, sr=44100, pre_filter_start=668, pre_filter_stop=672),
    "4band_v2_sn": dict(bins=672, unstable_bins=8, reduction_bins=637, band=_BANDS_4_SN, sr=44100, pre_filter_start=668, pre_filter_stop=672),
    "4band_v3_sn": dict(bins=672, unstable_bins=8, reduction_bins=530, band=_BANDS_4_SN, sr=44100, pre_filter_start=668, pre_filter_stop=672),
}
# "stereo_n": L' = (L + R / 4) / 0.9375, R' = (R + L / 4) / 0.9375 on the band's spectrogram; undone on the band's wave after the iSTFT by
# L = L' - R' / 4, R = R' - L' / 4.  The reference tree's spec_utils has no such branch (its wave_to_spectrogram :30-56 knows mid_side,
# mid_side_b2, reverse); the rule is the one of the engine the reference calls for this model (audio-separator's VR spec_utils
# ``convert_channels`` / ``spectrogram_to_wave``: upstream, uncited, restated from memory -- parity for "_sn" sets is unpinned).
_STEREO_N = 0.25
_STEREO_N_NORM = 1.0 - _STEREO_N * _STEREO_N


def _lp_gain(g: torch.Tensor, start: int, stop: int) -> None:
    """fft_lp_filter :432-440 as a per-bin gain (float64 running subtraction as the reference's Python floats, then float32)"""
    v = 1.0
    for b in range(start, stop):
        v -= 1 / (stop - start)
        g[b] *= v
    g[stop:] = 0


def _hp_gain(g: torch.Tensor, start: int, stop: int) -> None:
    """fft_hp_filter :443-451"""
    v = 1.0
    for b in range(start, stop, -1):
        v -= 1 / (start - stop)
        g[b] *= v
    g[: stop + 1] = 0


class VRFrontEnd:
    """Band split, stack, and the way back, for one parameter set; buffers are sized per call (frames depend on the input length)."""

    def __init__(self, params: str = "4band_v2", ctx: Optional[Context] = None):
        if params not in MODEL_PARAMS:
            raise AlsepError(f"VRFrontEnd: unknown parameter set {params!r} (known: {sorted(MODEL_PARAMS)})")
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        self.name = params
        self.mp = mp = MODEL_PARAMS[params]
        self.bands_n = len(mp["band"])
        self.bins = mp["bins"] + 1
        top = mp["band"][self.bands_n]
        self.hh = (top["n_fft"] // 2 - top["crop_stop"]) + (mp["pre_filter_stop"] - mp["pre_filter_start"])     # vr.py:88-93
        dev = self.ctx.device
        # pre-filter gains of the stacked spectrogram (combine_spectrograms :113-123), one table per band's slice of it
        g = torch.ones(self.bins, dtype=torch.float64)
        if mp["pre_filter_start"] > 0:
            gp = 1.0
            for b in range(mp["pre_filter_start"] + 1, mp["pre_filter_stop"]):
                gp = math.pow(10, -(b - mp["pre_filter_start"]) * (3.5 - gp) / 20.0)
                g[b] = gp
        self._stack_gain, self._band_gain, off = {}, {}, 0
        for d in range(1, self.bands_n + 1):
            bp = mp["band"][d]
            h = bp["crop_stop"] - bp["crop_start"]
            self._stack_gain[d] = g[off:off + h].float().to(dev)
            off += h
            # back end: the reference's filter order per band (:383-427)
            bg = torch.ones(bp["n_fft"] // 2 + 1, dtype=torch.float64)
            if d == self.bands_n:
                if bp.get("hpf_start", 0) > 0:
                    _hp_gain(bg, bp["hpf_start"], bp["hpf_stop"] - 1)
            elif d == 1:
                _lp_gain(bg, bp["lpf_start"], bp["lpf_stop"])
            else:
                _hp_gain(bg, bp["hpf_start"], bp["hpf_stop"] - 1)
                _lp_gain(bg, bp["lpf_start"], bp["lpf_stop"])
            self._band_gain[d] = bg.float().to(dev)
        if off > self.bins:
            raise AlsepError("VRFrontEnd: the bands' crops exceed the stacked height")
        for d, bp in mp["band"].items():
            if bp.get("convert_channels") not in (None, "stereo_n"):
                raise AlsepError(f"VRFrontEnd: band {d} of {params} asks for channel conversion {bp['convert_channels']!r}; only 'stereo_n' is built")
        self._plans: Dict[Tuple[int, int, int], StftPlan] = {}

    def _plan(self, n_fft: int, hop: int, dim_t: int) -> StftPlan:
        key = (n_fft, hop, dim_t)
        if key not in self._plans:
            if len(self._plans) > 16:                                               # frames depend on the track length: keep the cache small
                self._plans.clear()
            self._plans[key] = StftPlan(self.ctx, n_fft, hop, n_fft // 2 + 1, dim_t)
        return self._plans[key]

    def _resample(self, x: torch.Tensor, sr_in: int, sr_out: int, res_type: str) -> torch.Tensor:
        """librosa.resample as the reference calls it in the band chain: ``res_type`` "polyphase" on the way down (the parameter
        sets' value, vr.py:74-79) is scipy.signal.resample_poly, "scipy" on the way up (spec_utils.py:427) scipy.signal.resample; equal
        rates return the input (which is why the "sinc_fastest" of the lowest band, :404-414, never runs in the 4-band sets: bands 1 and 2
        share 7350 Hz).  Any other kind is refused: libsamplerate's and resampy's filters are not restated here."""
        if sr_in == sr_out:
            return x
        from . import ensemble
        if res_type == "polyphase":
            return ensemble.resample_poly(self.ctx, x, sr_in, sr_out)
        if res_type == "scipy":
            return ensemble.resample_fft(self.ctx, x, sr_in, sr_out)
        raise AlsepError(f"VRFrontEnd: resampler kind {res_type!r} ({sr_in} -> {sr_out} Hz) is not built (polyphase, scipy)")

    def _band_stft(self, wave: torch.Tensor, n_fft: int, hop: int) -> Tuple[torch.Tensor, int, int]:
        """librosa.stft(center=True, pad_mode="constant") of [2, n]: ([4, Fb, Ty] band spectrogram, first frame, frames)"""
        n = wave.shape[1]
        n_frames = 1 + n // hop
        off = -(-(n_fft // 2) // hop) * hop
        ty = off // hop + n_frames + -(-n_fft // hop) + 1
        buf = torch.zeros(1, 2, hop * (ty - 1), dtype=torch.float32, device=self.ctx.device)
        buf[0, :, off:off + n] = wave
        plan = self._plan(n_fft, hop, ty)
        spec = plan.stft_strided(buf, plan.chunk_size, 2 * plan.chunk_size, 1, torch.float32, _lib.LAYOUT_REF)
        return spec[0], off // hop, n_frames

    def _cross_mix(self, pair: torch.Tensor, self_gain: float, other_gain: float) -> None:
        """in place on a [2, ...] float32 pair: (a, b) <- (self_gain a + other_gain b, self_gain b + other_gain a).  The halves of a view
        need not sit on the 16-byte boundary ``alsep_axpby`` asks for: the arithmetic runs on fresh copies, the results are copied back."""
        ctx = self.ctx
        a, b = pair[0].clone(), pair[1].clone()
        ra, rb = a.clone(), b.clone()
        ctx.check(ctx.lib.alsep_axpby(ctx.handle, other_gain, _lib.ptr(b), self_gain, _lib.ptr(ra), ra.numel()), "alsep_axpby")
        ctx.check(ctx.lib.alsep_axpby(ctx.handle, other_gain, _lib.ptr(a), self_gain, _lib.ptr(rb), rb.numel()), "alsep_axpby")
        pair[0].copy_(ra)
        pair[1].copy_(rb)

    def analyse(self, wave: torch.Tensor):
        """vr.py:55-99: [2, n] at the set's rate -> (X [2, bins + 1, l] complex64, high end [2, hh, l] complex64)"""
        ctx, mp = self.ctx, self.mp
        lib = ctx.lib
        wave = torch.as_tensor(wave, dtype=torch.float32).to(ctx.device).contiguous()
        if wave.dim() != 2 or wave.shape[0] != 2:
            raise AlsepError("VRFrontEnd.analyse: expected a [2, n] stereo wave")
        specs, x = {}, wave
        for d in range(self.bands_n, 0, -1):
            bp = mp["band"][d]
            if d < self.bands_n:
                x = self._resample(x, mp["band"][d + 1]["sr"], bp["sr"], bp.get("res_type", "polyphase"))
            specs[d] = self._band_stft(x, bp["n_fft"], bp["hl"])
            if bp.get("convert_channels") == "stereo_n":                            # planes (L re, L im, R re, R im): a [2, 2 Fb Ty] pair
                band = specs[d][0]
                self._cross_mix(band.view(2, -1), 1.0 / _STEREO_N_NORM, _STEREO_N / _STEREO_N_NORM)
        l = min(s[2] for s in specs.values())
        X = torch.zeros(2, self.bins, l, 2, dtype=torch.float32, device=ctx.device)
        off = 0
        for d in range(1, self.bands_n + 1):
            bp = mp["band"][d]
            band, t0, _ = specs[d]
            h = bp["crop_stop"] - bp["crop_start"]
            ctx.check(lib.alsep_vr_band_crop(ctx.handle, _lib.ptr(band), _lib.ptr(X), _lib.ptr(self._stack_gain[d]), band.shape[1],
                                             band.shape[2], bp["crop_start"], t0, h, l, self.bins, off), "alsep_vr_band_crop")
            off += h
        top = mp["band"][self.bands_n]
        band, t0, _ = specs[self.bands_n]
        he = torch.empty(2, self.hh, l, 2, dtype=torch.float32, device=ctx.device)
        ctx.check(lib.alsep_vr_band_crop(ctx.handle, _lib.ptr(band), _lib.ptr(he), None, band.shape[1], band.shape[2],
                                         top["n_fft"] // 2 - self.hh, t0, self.hh, l, self.hh, 0), "alsep_vr_band_crop")
        return torch.view_as_complex(X), torch.view_as_complex(he)

    def split(self, pred: torch.Tensor, X: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """vr.py:110-111: y = pred * exp(i angle X), v = X - y"""
        ctx = self.ctx
        lib = ctx.lib
        pred = pred.to(torch.float32).contiguous()
        Xr = torch.view_as_real(X.contiguous())
        if tuple(pred.shape) != tuple(X.shape):
            raise AlsepError("VRFrontEnd.split: pred and X differ in shape")
        y, v = torch.empty_like(Xr), torch.empty_like(Xr)
        ctx.check(lib.alsep_vr_split_pred(ctx.handle, _lib.ptr(pred), _lib.ptr(Xr), _lib.ptr(y), _lib.ptr(v), pred.numel()),
                  "alsep_vr_split_pred")
        return torch.view_as_complex(y), torch.view_as_complex(v)

    def synthesise(self, spec_m: torch.Tensor, high_end: Optional[torch.Tensor] = None) -> torch.Tensor:
        """vr.py:161-168 + cmb_spectrogram_to_wave: complex [2, bins + 1, l] (+ the mix's high end for mirroring) -> wave [2, n]"""
        ctx, mp = self.ctx, self.mp
        lib = ctx.lib
        sm = torch.view_as_real(spec_m.contiguous())
        l = sm.shape[2]
        extra = None
        if high_end is not None:
            he = torch.view_as_real(high_end.contiguous())
            if he.shape[1] != self.hh or he.shape[2] != l:
                raise AlsepError("VRFrontEnd.synthesise: high end does not match the spectrogram")
            extra = torch.empty_like(he)
            ctx.check(lib.alsep_vr_mirror(ctx.handle, _lib.ptr(sm), _lib.ptr(he), _lib.ptr(extra), self.bins, self.hh, l,
                                          mp["pre_filter_start"] - 10 - self.hh), "alsep_vr_mirror")
        off, wave = 0, None
        for d in range(1, self.bands_n + 1):
            bp = mp["band"][d]
            fb = bp["n_fft"] // 2 + 1
            h = bp["crop_stop"] - bp["crop_start"]
            band = torch.empty(1, 4, fb, l, dtype=torch.float32, device=ctx.device)
            use_extra = extra is not None and d == self.bands_n
            ctx.check(lib.alsep_vr_band_spec(ctx.handle, _lib.ptr(sm), _lib.ptr(extra) if use_extra else None, _lib.ptr(self._band_gain[d]),
                                             _lib.ptr(band), self.bins, l, fb, bp["crop_start"], h, off,
                                             bp["n_fft"] // 2 - self.hh if use_extra else 0, self.hh if use_extra else 0), "alsep_vr_band_spec")
            off += h
            plan = self._plan(bp["n_fft"], bp["hl"], l)
            w = ctx.empty((2, plan.chunk_size), torch.float32)                      # [2, hl * (l - 1)]
            plan.istft_strided(band, _lib.LAYOUT_REF, w, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size, plan.chunk_size)
            if bp.get("convert_channels") == "stereo_n":
                self._cross_mix(w, 1.0, -_STEREO_N)
            if wave is not None:
                if wave.shape[1] != w.shape[1]:
                    raise AlsepError(f"VRFrontEnd.synthesise: band {d} is {w.shape[1]} samples, the chain below it {wave.shape[1]}")
                w = w + wave
            # spec_utils.py:402-427: the lowest band goes up with "sinc_fastest", the middle ones with "scipy"
            wave = w if d == self.bands_n else self._resample(w, bp["sr"], mp["band"][d + 1]["sr"], "sinc_fastest" if d == 1 else "scipy")
        return wave


class VRSeparator:
    """``AudioPre`` (vr.py:20-196) on the device: a VR network + its parameter set.  ``separate`` returns the two waves the reference
    writes (``instrument_*`` = pred * phase, ``vocal_*`` = the residual spectrogram), both with the mirrored high end (``high_end_process``
    "mirroring", vr.py:32) unless ``high_end_process=False``."""

    def __init__(self, net: VRNet, params: str = "4band_v2", agg: int = 10, window_size: int = 512, tta: bool = False, max_batch: int = 4,
                 high_end_process: bool = True):
        self.net = net
        self.front = VRFrontEnd(params, net.ctx)
        if net.output_bin != self.front.bins:
            raise AlsepError(f"VRSeparator: the network has {net.output_bin} bins, parameter set {params} stacks {self.front.bins}")
        self.agg, self.window_size, self.tta, self.max_batch = agg, window_size, tta, max_batch
        self.high_end_process = high_end_process

    @property
    def sample_rate(self) -> int:
        return self.front.mp["sr"]

    @torch.no_grad()
    def separate(self, wave: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        f = self.front
        X, he = f.analyse(wave)
        aggr = {"value": self.agg / 100.0, "split_bin": f.mp["band"][1]["crop_stop"]}                            # vr.py:97-101
        pred, _, _ = vr_inference(self.net, X, aggr, self.window_size, self.tta, self.max_batch)
        y, v = f.split(pred, X)
        if not self.high_end_process:                                                                            # vr.py:121-125, 163-167
            he = None
        return f.synthesise(y, he), f.synthesise(v, he)
