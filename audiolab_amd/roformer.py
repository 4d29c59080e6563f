"""Roformer separation networks (BS-RoFormer, Mel-Band RoFormer) on the GPU -- the first members of the reference's default
ensemble (modules/separator/stem_separator.py:379-382: ``vocals_mel_band_roformer.ckpt``, ``model_bs_roformer_ep_368_sdr_12.9628.ckpt``,
``melband_roformer_big_beta4.ckpt``) and its de-reverb / de-echo transforms (:796-797), reached there through
``Separator.load_model(<name>.ckpt)`` / ``.separate``.

The network code lives in the un-vendored ``audio-separator[gpu]>=0.32.0`` (setup.sh:96): PARITY UNPINNED -- restated from the published
design; oracle/roformer_oracle.py is the torch-CPU fp32 twin the kernels are checked against.  Parameter names are that code's
(``band_split.to_features.{i}.1.weight``, ``layers.{l}.{0|1}.layers.0.0.to_qkv.weight``, ``mask_estimators.{s}.to_freqs.{i}.0.net.{2j}.weight``),
so a real ``state_dict`` loads as it is; the band layout (contiguous ``freqs_per_bands`` or the bins under each slaney mel filter) is
rebuilt from the hyper-parameters.

Everything runs in libalsep.so on float32 tensors (csrc/nn.hip): tokens are kept ``[T, bands, dim]`` for the whole network -- the time
transformer (attention over frames within a band) and the frequency transformer (attention over bands within a frame) differ only in
the strides given to the batched GEMMs and in the position fed to the rotary embedding, so nothing is ever transposed.  STFT / iSTFT:
the n_fft 2048 / hop 441 kernels of csrc/fft.hip.  Runner: the chunked inference of the training project (chunks every
``chunk_size / num_overlap`` samples of the reflect-padded track, linear edge fades, sum / counter).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import AlsepError, Context

BS_FREQS_PER_BANDS = (2,) * 24 + (4,) * 12 + (12,) * 8 + (24,) * 8 + (48,) * 8 + (128, 129)


@dataclass(frozen=True)
class RoformerConfig:
    kind: str = "bs"                       # "bs" | "mel"
    dim: int = 384
    depth: int = 12
    heads: int = 8
    dim_head: int = 64
    num_stems: int = 1
    n_fft: int = 2048
    hop: int = 441
    num_bands: int = 60
    freqs_per_bands: Tuple[int, ...] = BS_FREQS_PER_BANDS
    sample_rate: int = 44100
    mask_estimator_depth: int = 2
    mlp_expansion_factor: int = 4
    chunk_size: int = 352800
    num_overlap: int = 4

    @property
    def n_freq(self) -> int:
        return self.n_fft // 2 + 1


def _mel_points(sr: int, n_mels: int) -> np.ndarray:
    """slaney mel scale: n_mels + 2 band edges in Hz between 0 and sr / 2 (librosa.mel_frequencies, htk=False)"""
    f_sp, min_log_hz = 200.0 / 3, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def to_mel(f):
        return min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp
    m = np.linspace(to_mel(0.0), to_mel(sr / 2.0), n_mels + 2)
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def band_indices(cfg: RoformerConfig) -> List[np.ndarray]:
    """per band: indices into the merged (frequency, channel) axis, m = 2 f + s"""
    nf = cfg.n_freq
    if cfg.kind == "bs":
        if sum(cfg.freqs_per_bands) != nf:
            raise AlsepError("freqs_per_bands must sum to n_fft / 2 + 1")
        edges = np.cumsum((0,) + tuple(cfg.freqs_per_bands))
        per_f = [np.arange(edges[i], edges[i + 1]) for i in range(len(cfg.freqs_per_bands))]
    elif cfg.kind == "mel":
        # bins with a positive weight under each triangular filter of librosa.filters.mel(sr, n_fft, n_mels); the first filter also takes
        # bin 0 and the last one the Nyquist bin (mel_band_roformer.py sets those two weights to 1)
        pts = _mel_points(cfg.sample_rate, cfg.num_bands)
        freqs = np.linspace(0, cfg.sample_rate / 2.0, nf)
        per_f = []
        for i in range(cfg.num_bands):
            lower = (freqs - pts[i]) / (pts[i + 1] - pts[i])
            upper = (pts[i + 2] - freqs) / (pts[i + 2] - pts[i + 1])
            wt = np.maximum(0.0, np.minimum(lower, upper)) * (2.0 / (pts[i + 2] - pts[i]))
            on = wt.astype(np.float32) > 0
            if i == 0:
                on[0] = True
            if i == cfg.num_bands - 1:
                on[-1] = True
            per_f.append(np.nonzero(on)[0])
    else:
        raise AlsepError("RoformerConfig.kind must be 'bs' or 'mel'")
    return [(f[:, None] * 2 + np.arange(2)[None]).reshape(-1) for f in per_f]


def check_shapes(sd: Dict[str, torch.Tensor], expected: Dict[str, Tuple[Tuple[int, ...], str]], family: str,
                 forbidden_prefixes: Tuple[Tuple[str, str], ...] = ()) -> None:
    """Every tensor of ``expected`` {name: (shape, hyper-parameter that fixes it)} must be in ``sd`` with exactly that shape, and no key
    may start with one of ``forbidden_prefixes`` [(prefix, hyper-parameter)] -- e.g. ``layers.{depth}.`` for a checkpoint deeper than
    the configuration.  The kernels take strides and buffer sizes from the configuration, never from the tensors: a checkpoint whose
    hyper-parameters differ from it would read out of bounds or run a truncated network.  Raises AlsepError naming the hyper-parameter
    (the key of the model's ``.yaml`` that overrides the roster value)."""
    for prefix, hyper in forbidden_prefixes:
        extra = next((k for k in sd if k.startswith(prefix)), None)
        if extra is not None:
            raise AlsepError(f"{family}: the checkpoint holds '{extra}', beyond the configured {hyper} -- the configuration would silently "
                             f"drop it (set {hyper.split('=')[0].strip()} in the model's .yaml beside the weights)")
    for name, (shape, hyper) in expected.items():
        t = sd.get(name)
        if t is None:
            raise AlsepError(f"{family}: state_dict is missing '{name}' (configuration value: {hyper})")
        if tuple(t.shape) != tuple(shape):
            raise AlsepError(f"{family}: '{name}' has shape {tuple(t.shape)}, the configuration expects {tuple(shape)} -- "
                             f"hyper-parameter mismatch: {hyper} (set it in the model's .yaml beside the weights)")


def expected_shapes(cfg: "RoformerConfig") -> Dict[str, Tuple[Tuple[int, ...], str]]:
    """name -> (shape, hyper-parameter) of every tensor a Roformer with this configuration reads"""
    exp: Dict[str, Tuple[Tuple[int, ...], str]] = {}
    bands = band_indices(cfg)
    inner, hidden = cfg.heads * cfg.dim_head, cfg.dim * cfg.mlp_expansion_factor
    layout = f"band layout ({'freqs_per_bands' if cfg.kind == 'bs' else 'num_bands / sample_rate / stft_n_fft'})"
    for i, idx in enumerate(bands):
        w = 2 * len(idx)
        exp[f"band_split.to_features.{i}.0.gamma"] = ((w,), layout)
        exp[f"band_split.to_features.{i}.1.weight"] = ((cfg.dim, w), f"dim={cfg.dim} / {layout}")
        exp[f"band_split.to_features.{i}.1.bias"] = ((cfg.dim,), f"dim={cfg.dim}")
    for li in range(cfg.depth):
        for tr in (0, 1):
            p = f"layers.{li}.{tr}.layers.0"
            exp[p + ".0.norm.gamma"] = ((cfg.dim,), f"dim={cfg.dim} (depth={cfg.depth})")
            exp[p + ".0.to_qkv.weight"] = ((3 * inner, cfg.dim), f"heads={cfg.heads} x dim_head={cfg.dim_head}, dim={cfg.dim}")
            exp[p + ".0.to_gates.weight"] = ((cfg.heads, cfg.dim), f"heads={cfg.heads}")
            exp[p + ".0.to_gates.bias"] = ((cfg.heads,), f"heads={cfg.heads}")
            exp[p + ".0.to_out.0.weight"] = ((cfg.dim, inner), f"heads={cfg.heads} x dim_head={cfg.dim_head}")
            exp[p + ".1.net.0.gamma"] = ((cfg.dim,), f"dim={cfg.dim}")
            exp[p + ".1.net.1.weight"] = ((4 * cfg.dim, cfg.dim), f"dim={cfg.dim} (feed-forward factor 4)")
            exp[p + ".1.net.1.bias"] = ((4 * cfg.dim,), f"dim={cfg.dim}")
            exp[p + ".1.net.4.weight"] = ((cfg.dim, 4 * cfg.dim), f"dim={cfg.dim}")
            exp[p + ".1.net.4.bias"] = ((cfg.dim,), f"dim={cfg.dim}")
    exp["final_norm.gamma"] = ((cfg.dim,), f"dim={cfg.dim}")
    for s_ in range(cfg.num_stems):
        for i, idx in enumerate(bands):
            dims = (cfg.dim,) + (hidden,) * (cfg.mask_estimator_depth - 1) + (4 * len(idx),)
            for j in range(cfg.mask_estimator_depth):
                hyper = f"mask_estimator_depth={cfg.mask_estimator_depth}, mlp_expansion_factor={cfg.mlp_expansion_factor}, {layout}"
                exp[f"mask_estimators.{s_}.to_freqs.{i}.0.net.{2 * j}.weight"] = ((dims[j + 1], dims[j]), hyper)
                exp[f"mask_estimators.{s_}.to_freqs.{i}.0.net.{2 * j}.bias"] = ((dims[j + 1],), hyper)
    return exp


class _Lin:
    def __init__(self, ctx: Context, w: torch.Tensor, b: Optional[torch.Tensor], half: bool = False):
        self.out, self.inp = (int(v) for v in w.shape)
        self.w = w.detach().float().contiguous().to(ctx.device)              # [out, in]: the B operand of alsep_nn_bgemm (rows n, k contiguous)
        self.b = b.detach().float().contiguous().to(ctx.device) if b is not None else None
        # half-precision mode: the weight matrix once more as IEEE half (rounded on the device, round-to-nearest-even) for the f16 MFMA
        # kernel -- only for shapes that kernel takes (whole 8-element k-groups, float4 output columns)
        # (output rows zero-padded to a multiple of 4: the kernel stores whole 4-column groups)
        self.wh, self.out_p, self.bh = None, -(-self.out // 4) * 4, None
        if half:
            if self.inp % 8:
                raise AlsepError(f"Roformer half mode: a Linear with {self.inp} inputs (must be a multiple of 8)")
            wp = torch.zeros((self.out_p, self.inp), device=ctx.device)
            wp[: self.out] = self.w
            self.wh = torch.empty((self.out_p, self.inp), dtype=torch.float16, device=ctx.device)
            ctx.check(ctx.lib.alsep_nn_to_f16(ctx.handle, _lib.ptr(wp), _lib.ptr(self.wh), wp.numel()), "alsep_nn_to_f16")
            if self.b is not None:
                self.bh = torch.zeros((self.out_p,), device=ctx.device)
                self.bh[: self.out] = self.b


class Roformer:
    def __init__(self, cfg: RoformerConfig, state_dict: Dict[str, torch.Tensor], ctx: Optional[Context] = None, precision: str = "f32"):
        """``precision``: "f32" -- every product on the exact-fp32 MFMA (the 1e-4 parity mode); "f16" -- the arithmetic of the reference's
        ``use_autocast=True`` (stem_separator.py:106): the Linear layers of the transformer blocks and of the mask estimators and the
        attention products take IEEE-half operands (csrc/nn_half.hip: f16 MFMA, float32 accumulation, one-pass attention), norms /
        rotary / softmax statistics / residuals / STFT stay float32.  The per-band layers (input projections, mask estimators) run as one
        batched GEMM per layer over all bands (zero-padded to the widest band), the rotary embedding and the head gates inside the
        attention kernel: ~105 launches per chunk instead of ~600."""
        if precision not in ("f32", "f16"):
            raise AlsepError("Roformer precision must be 'f32' or 'f16'")
        self.cfg = cfg
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        self.dtype = torch.float32
        self.precision = precision
        half = precision == "f16"
        if half and cfg.dim_head != 64:
            raise AlsepError("Roformer half mode: the one-pass attention kernel is written for dim_head 64")
        sd = state_dict
        dev = self.ctx.device
        bands = band_indices(cfg)
        check_shapes(sd, expected_shapes(cfg), f"Roformer ({cfg.kind})",
                     ((f"layers.{cfg.depth}.", f"depth={cfg.depth}"), (f"mask_estimators.{cfg.num_stems}.", f"num_stems={cfg.num_stems}"),
                      (f"band_split.to_features.{len(bands)}.", f"number of bands={len(bands)}"),
                      *((f"mask_estimators.{s_}.to_freqs.0.0.net.{2 * cfg.mask_estimator_depth}.", f"mask_estimator_depth={cfg.mask_estimator_depth}")
                        for s_ in range(cfg.num_stems))))
        self.nb = len(bands)
        self._bands = bands
        self.band_len = [len(b) for b in bands]
        self.band_off = np.concatenate([[0], np.cumsum(self.band_len)]).astype(np.int64)      # offsets in merged-index entries
        self.n_idx = int(self.band_off[-1])
        self.midx = torch.from_numpy(np.concatenate(bands).astype(np.int32)).to(dev)
        # occurrences of every merged bin in the concatenated band lists (CSR), with the columns of the mask estimator's output
        H = 4 * self.n_idx                                                  # per band 2 * (2 len): value half | gate half
        occ: List[List[Tuple[int, int]]] = [[] for _ in range(2 * cfg.n_freq)]
        for bi, idx in enumerate(bands):
            cb, n = 4 * int(self.band_off[bi]), 2 * len(idx)
            for i, m in enumerate(idx):
                occ[int(m)].append((cb + 2 * i, cb + n + 2 * i))
        if any(len(o) == 0 for o in occ):
            raise AlsepError("band layout leaves a frequency bin uncovered")
        self.occ_start = torch.tensor(np.concatenate([[0], np.cumsum([len(o) for o in occ])]), dtype=torch.int32, device=dev)
        self.col_a = torch.tensor([a for o in occ for a, _ in o], dtype=torch.int32, device=dev)
        self.col_g = torch.tensor([g for o in occ for _, g in o], dtype=torch.int32, device=dev)
        self.H = H
        try:
            v = lambda k: sd[k].detach().float().contiguous().to(dev)
            self.split = [(v(f"band_split.to_features.{i}.0.gamma"),
                           _Lin(self.ctx, sd[f"band_split.to_features.{i}.1.weight"], sd[f"band_split.to_features.{i}.1.bias"]))
                          for i in range(self.nb)]
            self.layers = []
            for li in range(cfg.depth):
                pair = []
                for tr in (0, 1):
                    p = f"layers.{li}.{tr}.layers.0"
                    pair.append(dict(norm=v(p + ".0.norm.gamma"), qkv=_Lin(self.ctx, sd[p + ".0.to_qkv.weight"], None, half),
                                     gates=_Lin(self.ctx, sd[p + ".0.to_gates.weight"], sd[p + ".0.to_gates.bias"], half),
                                     out=_Lin(self.ctx, sd[p + ".0.to_out.0.weight"], None, half), ffn=v(p + ".1.net.0.gamma"),
                                     l1=_Lin(self.ctx, sd[p + ".1.net.1.weight"], sd[p + ".1.net.1.bias"], half),
                                     l2=_Lin(self.ctx, sd[p + ".1.net.4.weight"], sd[p + ".1.net.4.bias"], half)))
                self.layers.append(pair)
            self.final_norm = v("final_norm.gamma")
            self.masks = [[[_Lin(self.ctx, sd[f"mask_estimators.{s}.to_freqs.{i}.0.net.{2 * j}.weight"],
                                 sd[f"mask_estimators.{s}.to_freqs.{i}.0.net.{2 * j}.bias"], half) for j in range(cfg.mask_estimator_depth)]
                           for i in range(self.nb)] for s in range(cfg.num_stems)]
        except KeyError as e:
            raise AlsepError(f"state_dict is missing {e} for this RoformerConfig") from e
        self._plans: Dict[int, object] = {}
        self._rot: Dict[int, torch.Tensor] = {}                # rotary tables per sequence length (half mode)
        self._mask_cols: Dict[int, tuple] = {}                 # mask-kernel column tables of the padded estimator output, per frame count
        if half:
            self._build_half_images(sd, bands)

    def _build_half_images(self, sd, bands) -> None:
        """Half mode runs the per-band layers (input projections, mask estimators) as ONE batched f16 GEMM per layer over all bands:
        the ragged widths are zero-padded to the widest band (K of the input projections, N of the estimators' last layers; the
        kernel skips the column tiles beyond a band's true N)."""
        ctx, cfg, dev = self.ctx, self.cfg, self.ctx.device
        nb, dim = self.nb, cfg.dim

        def f16(t: torch.Tensor) -> torch.Tensor:
            src = t.detach().float().contiguous().to(dev)
            dst = torch.empty(src.shape, dtype=torch.float16, device=dev)
            ctx.check(ctx.lib.alsep_nn_to_f16(ctx.handle, _lib.ptr(src), _lib.ptr(dst), src.numel()), "alsep_nn_to_f16")
            return dst
        widths = [2 * len(b) for b in bands]
        kmax = -(-max(widths) // 8) * 8
        if kmax > 640:
            raise AlsepError("Roformer half mode: a band wider than 320 bins is not supported")
        pidx = torch.full((nb, kmax // 2), -1, dtype=torch.int32)
        gam = torch.zeros((nb, kmax))
        w0 = torch.zeros((nb, dim, kmax))
        b0 = torch.zeros((nb, dim))
        for i, idx in enumerate(bands):
            pidx[i, : len(idx)] = torch.from_numpy(np.asarray(idx, dtype=np.int32))
            gam[i, : widths[i]] = sd[f"band_split.to_features.{i}.0.gamma"].float()
            w0[i, :, : widths[i]] = sd[f"band_split.to_features.{i}.1.weight"].float()
            b0[i] = sd[f"band_split.to_features.{i}.1.bias"].float()
        self.h_kmax = kmax
        self.h_pidx, self.h_gamma = pidx.to(dev), gam.contiguous().to(dev)
        self.h_width = torch.tensor(widths, dtype=torch.int32, device=dev)
        self.h_split_w, self.h_split_b = f16(w0), b0.contiguous().to(dev)
        # mask estimators: per stem a list of layers (stacked weights f16 [nb][out][in], bias [nb][out], per-band out or None)
        nouts = [4 * len(b) for b in bands]
        self.h_nmax = -(-max(nouts) // 4) * 4
        self.h_nout = torch.tensor(nouts, dtype=torch.int32, device=dev)
        self.h_masks = []
        for s_ in range(cfg.num_stems):
            layers = []
            for j in range(cfg.mask_estimator_depth):
                last = j + 1 == cfg.mask_estimator_depth
                ws = [sd[f"mask_estimators.{s_}.to_freqs.{i}.0.net.{2 * j}.weight"].float() for i in range(nb)]
                bs = [sd[f"mask_estimators.{s_}.to_freqs.{i}.0.net.{2 * j}.bias"].float() for i in range(nb)]
                k_in = ws[0].shape[1]
                n_out = self.h_nmax if last else ws[0].shape[0]
                wst, bst = torch.zeros((nb, n_out, k_in)), torch.zeros((nb, n_out))
                for i in range(nb):
                    wst[i, : ws[i].shape[0]] = ws[i]
                    bst[i, : bs[i].shape[0]] = bs[i]
                layers.append((f16(wst), bst.contiguous().to(dev), n_out, k_in, last))
            self.h_masks.append(layers)

    def _rot_table(self, L: int) -> torch.Tensor:
        if L not in self._rot:
            t = self.ctx.empty((L, self.cfg.dim_head // 2, 2))
            self.ctx.check(self.ctx.lib.alsep_nn_rotary_table(self.ctx.handle, _lib.ptr(t), L, self.cfg.dim_head), "alsep_nn_rotary_table")
            self._rot[L] = t
        return self._rot[L]

    def _padded_mask_cols(self, T: int):
        """col_a / col_g of roformer_mask for the PADDED estimator output [bands][T][nmax]: entry = band * T * nmax + column in the band"""
        if T not in self._mask_cols:
            nmax = self.h_nmax
            a, g = [], []
            occ: List[List[Tuple[int, int]]] = [[] for _ in range(2 * self.cfg.n_freq)]
            for bi, idx in enumerate(self._bands):
                base, n = bi * T * nmax, 2 * len(idx)
                for i, m in enumerate(idx):
                    occ[int(m)].append((base + 2 * i, base + n + 2 * i))
            for o in occ:
                for ca, cg in o:
                    a.append(ca)
                    g.append(cg)
            dev = self.ctx.device
            self._mask_cols[T] = (torch.tensor(a, dtype=torch.int32, device=dev), torch.tensor(g, dtype=torch.int32, device=dev))
        return self._mask_cols[T]

    def _bgemm_h(self, a16: torch.Tensor, lda: int, sa_b: int, w: torch.Tensor, bias: torch.Tensor, c: torch.Tensor, c_off: int, ldc: int, sc_b: int,
                 M: int, N: int, K: int, act: int = 0, nvec: Optional[torch.Tensor] = None) -> None:
        """one batched f16 GEMM over the bands: C[b] = act(A[b] W[b]^T + bias[b]); C half or float32 by its dtype, written from element c_off"""
        ctx = self.ctx
        half_out = c.dtype == torch.float16
        ctx.check(ctx.lib.alsep_nn_gemm_f16(ctx.handle, _lib.ptr(a16), lda, sa_b, _lib.ptr(w), K, N * K,
                                            C.c_void_p(c.data_ptr() + c_off * (2 if half_out else 4)), 1 if half_out else 0, ldc, sc_b, _lib.ptr(bias), N,
                                            None, 0, 0, self.nb, M, N, K, 1.0, act, _lib.ptr(nvec) if nvec is not None else None), "alsep_nn_gemm_f16")

    # -- helpers --------------------------------------------------------------------------------------------
    def _gemm(self, a_ptr: int, sa, lin: _Lin, c_ptr: int, sc, M: int, act: int = 0, nb1: int = 1, nb2: int = 1, use_bias: bool = True) -> None:
        """float32 mode: C = act(A W^T + b) with A / C given by (pointer, strides (b1, b2, row, k))"""
        ctx = self.ctx
        arr = C.c_int64 * 4
        bias = _lib.ptr(lin.b) if (lin.b is not None and use_bias) else None
        ctx.check(ctx.lib.alsep_nn_bgemm_bias(ctx.handle, C.c_void_p(a_ptr), _lib.ptr(lin.w), C.c_void_p(c_ptr), nb1, nb2, M, lin.out, lin.inp,
                                              arr(*sa), arr(0, 0, lin.inp, 1), arr(*sc), 1.0, bias, act), "alsep_nn_bgemm_bias")

    def _dense(self, x: torch.Tensor, rows: int, lin: _Lin, act: int = 0, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """float32 mode: act(x W^T + b) [+ residual]"""
        y = self.ctx.empty((rows, lin.out))
        self._gemm(x.data_ptr(), (0, 0, lin.inp, 1), lin, y.data_ptr(), (0, 0, lin.out, 1), rows, act)
        if residual is not None:
            out = self.ctx.empty((rows, lin.out))
            self.ctx.check(self.ctx.lib.alsep_nn_scale_add(self.ctx.handle, _lib.ptr(residual), _lib.ptr(y), None, _lib.ptr(out), rows, lin.out),
                           "alsep_nn_scale_add")
            return out
        return y

    # -- half mode: IEEE-half activations between the kernels, float32 residual stream -------------------------------------------------
    def _dense_h(self, x16: torch.Tensor, rows: int, lin: _Lin, act: int = 0, out_half: bool = True, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        """act(x W^T + b) [+ residual] on the f16 MFMA: x16 [rows, in] half -> [rows, out padded to 4] half or float32"""
        ctx = self.ctx
        y = ctx.empty((rows, lin.out_p), torch.float16 if out_half else torch.float32)
        ctx.check(ctx.lib.alsep_nn_gemm_f16(ctx.handle, _lib.ptr(x16), lin.inp, 0, _lib.ptr(lin.wh), lin.inp, 0, _lib.ptr(y), 1 if out_half else 0,
                                            lin.out_p, 0, _lib.ptr(lin.bh) if lin.bh is not None else None, 0,
                                            _lib.ptr(residual) if residual is not None else None, lin.out_p, 0, 1, rows, lin.out_p, lin.inp, 1.0, act,
                                            None), "alsep_nn_gemm_f16")
        return y

    def _rmsnorm_h(self, x: torch.Tensor, rows: int, Cn: int, gamma: torch.Tensor) -> torch.Tensor:
        ctx = self.ctx
        y = ctx.empty((rows, Cn), torch.float16)
        ctx.check(ctx.lib.alsep_nn_rmsnorm_f16(ctx.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), rows, Cn, Cn, Cn), "alsep_nn_rmsnorm_f16")
        return y

    def _transformer_h(self, x: torch.Tensor, T: int, P, over_time: bool) -> torch.Tensor:
        """one transformer block in the half mode: 8 launches (RMSNorm, q | k | v, gates, attention with the rotary embedding and the head
        gates inside, output projection + residual, RMSNorm, GELU Linear, Linear + residual)"""
        ctx, cfg = self.ctx, self.cfg
        nb, dim, Hh, d = self.nb, cfg.dim, cfg.heads, cfg.dim_head
        inner, rows = Hh * d, T * nb
        xn = self._rmsnorm_h(x, rows, dim, P["norm"])
        qkv = self._dense_h(xn, rows, P["qkv"])                               # half [rows, 3 inner], columns (qkv, head, d)
        gates = self._dense_h(xn, rows, P["gates"], out_half=False)           # float32 [rows, heads padded to 4]
        ld, gp = 3 * inner, P["gates"].out_p
        if over_time:                                                         # batch (band, head); a sequence's rows are bands * ld apart
            n_seq, L, seq_stride, row_stride, o_seq, o_row, g_seq, g_row = nb, T, ld, nb * ld, inner, nb * inner, gp, nb * gp
        else:                                                                 # batch (frame, head); rows are ld apart
            n_seq, L, seq_stride, row_stride, o_seq, o_row, g_seq, g_row = T, nb, nb * ld, ld, nb * inner, inner, nb * gp, gp
        att = ctx.empty((rows, inner), torch.float16)
        ctx.check(ctx.lib.alsep_nn_attention_f16(ctx.handle, _lib.ptr(qkv), _lib.ptr(att), n_seq, L, Hh, d, seq_stride, row_stride, o_seq, o_row,
                                                 d ** -0.5, _lib.ptr(self._rot_table(L)), _lib.ptr(gates), g_seq, g_row), "alsep_nn_attention_f16")
        x1 = self._dense_h(att, rows, P["out"], out_half=False, residual=x)
        f = self._dense_h(self._rmsnorm_h(x1, rows, dim, P["ffn"]), rows, P["l1"], act=3)
        return self._dense_h(f, rows, P["l2"], out_half=False, residual=x1)

    def _rmsnorm(self, x: torch.Tensor, rows: int, Cn: int, gamma: torch.Tensor) -> torch.Tensor:
        ctx = self.ctx
        y = ctx.empty((rows, Cn))
        ctx.check(ctx.lib.alsep_nn_rmsnorm(ctx.handle, _lib.ptr(x), _lib.ptr(y), _lib.ptr(gamma), rows, Cn, Cn, Cn), "alsep_nn_rmsnorm")
        return y

    def _transformer(self, x: torch.Tensor, T: int, P, over_time: bool) -> torch.Tensor:
        """x [T, bands, dim] -> same.  over_time: sequences are the frames of one band; else the bands of one frame."""
        if self.precision == "f16":
            return self._transformer_h(x, T, P, over_time)
        ctx, cfg = self.ctx, self.cfg
        lib, h = ctx.lib, ctx.handle
        nb, dim, Hh, d = self.nb, cfg.dim, cfg.heads, cfg.dim_head
        inner = Hh * d
        rows = T * nb
        xn = self._rmsnorm(x, rows, dim, P["norm"])
        qkv = self._dense(xn, rows, P["qkv"])                                 # [rows, 3 inner], columns (qkv, head, d)
        pos = (nb, T) if over_time else (1, nb)                               # row r = t * bands + f: position t, or f
        for off in (0, inner):
            ctx.check(lib.alsep_nn_rotary(h, _lib.ptr(qkv), rows, 3 * inner, off, Hh, d, pos[0], pos[1]), "alsep_nn_rotary")
        arr = C.c_int64 * 4
        ld = 3 * inner
        if over_time:                                                         # batch (band, head); a sequence's rows are bands * ld apart
            n_seq, L, seq_stride, row_stride = nb, T, ld, nb * ld
        else:                                                                 # batch (frame, head); rows are ld apart
            n_seq, L, seq_stride, row_stride = T, nb, nb * ld, ld
        att = ctx.empty((rows, inner))
        o_seq, o_row = (inner, nb * inner) if over_time else (nb * inner, inner)
        Lp = -(-L // 4) * 4                                                    # score rows padded to 16 bytes: the tiled GEMM's float4 loads
        scores = ctx.empty((n_seq, Hh, L, Lp))
        base = qkv.data_ptr()
        ctx.check(lib.alsep_nn_bgemm(h, C.c_void_p(base), C.c_void_p(base + 4 * inner), _lib.ptr(scores), n_seq, Hh, L, L, d,
                                     arr(seq_stride, d, row_stride, 1), arr(seq_stride, d, row_stride, 1), arr(Hh * L * Lp, L * Lp, Lp, 1),
                                     d ** -0.5), "alsep_nn_bgemm")
        ctx.check(lib.alsep_nn_softmax_rows_ld(h, _lib.ptr(scores), n_seq * Hh * L, L, Lp), "alsep_nn_softmax_rows_ld")
        ctx.check(lib.alsep_nn_bgemm(h, _lib.ptr(scores), C.c_void_p(base + 8 * inner), _lib.ptr(att), n_seq, Hh, L, d, L,
                                     arr(Hh * L * Lp, L * Lp, Lp, 1), arr(seq_stride, d, 1, row_stride), arr(o_seq, d, o_row, 1), 1.0), "alsep_nn_bgemm")
        gates = self._dense(xn, rows, P["gates"])
        ctx.check(lib.alsep_nn_gate(h, _lib.ptr(att), _lib.ptr(gates), rows, Hh, d), "alsep_nn_gate")
        x1 = self._dense(att, rows, P["out"], residual=x)
        f = self._dense(self._rmsnorm(x1, rows, dim, P["ffn"]), rows, P["l1"], act=3)
        return self._dense(f, rows, P["l2"], residual=x1)

    def _plan(self, dim_t: int):
        from .mdx import StftPlan
        if dim_t not in self._plans:
            self._plans[dim_t] = StftPlan(self.ctx, self.cfg.n_fft, self.cfg.hop, self.cfg.n_freq, dim_t)
        return self._plans[dim_t]

    # -- forward --------------------------------------------------------------------------------------------
    def forward(self, audio: torch.Tensor) -> torch.Tensor:
        """audio [2, L] float32 on the device, L a multiple of hop -> [num_stems, 2, L]"""
        ctx, cfg = self.ctx, self.cfg
        lib, h = ctx.lib, ctx.handle
        if audio.dim() != 2 or audio.shape[0] != 2 or audio.dtype != torch.float32:
            raise AlsepError("Roformer.forward expects a float32 [2, L] tensor")
        L = audio.shape[-1]
        if L % cfg.hop:
            raise AlsepError(f"Roformer.forward: length {L} is not a multiple of hop {cfg.hop}")
        audio = audio.contiguous()
        T, Fq, nb, dim = L // cfg.hop + 1, cfg.n_freq, self.nb, cfg.dim
        plan = self._plan(T)
        spec = plan.stft_strided(audio, L, 2 * L, 1, torch.float32, _lib.LAYOUT_REF)        # [1, 4, Fq, T]
        x = ctx.empty((T, nb, dim))
        half = self.precision == "f16"
        if half:                                                              # all bands: one gather + RMSNorm launch, one batched f16 GEMM
            kmax = self.h_kmax
            featp = ctx.empty((nb, T, kmax), torch.float16)
            ctx.check(lib.alsep_roformer_bandsplit_in(h, _lib.ptr(spec), _lib.ptr(self.h_pidx), _lib.ptr(self.h_gamma), _lib.ptr(self.h_width),
                                                      _lib.ptr(featp), nb, Fq, T, kmax), "alsep_roformer_bandsplit_in")
            self._bgemm_h(featp, kmax, T * kmax, self.h_split_w, self.h_split_b, x, 0, nb * dim, dim, T, dim, kmax)
        else:
            feat = ctx.empty((T, 2 * self.n_idx))
            ctx.check(lib.alsep_roformer_gather(h, _lib.ptr(spec), _lib.ptr(self.midx), _lib.ptr(feat), self.n_idx, Fq, T), "alsep_roformer_gather")
            FW = 2 * self.n_idx
            for i, (gamma, lin) in enumerate(self.split):                     # band split: RMSNorm + Linear per band, on column slices
                col = 2 * int(self.band_off[i])
                p = feat.data_ptr() + 4 * col
                ctx.check(lib.alsep_nn_rmsnorm(h, C.c_void_p(p), C.c_void_p(p), _lib.ptr(gamma), T, lin.inp, FW, FW), "alsep_nn_rmsnorm")
                self._gemm(p, (0, 0, FW, 1), lin, x.data_ptr() + 4 * i * dim, (0, 0, nb * dim, 1), T)
        for pair in self.layers:
            x = self._transformer(x, T, pair[0], over_time=True)
            x = self._transformer(x, T, pair[1], over_time=False)
        x = (self._rmsnorm_h if half else self._rmsnorm)(x, T * nb, dim, self.final_norm)
        out = ctx.empty((cfg.num_stems, 2, L))
        hidden = cfg.dim * cfg.mlp_expansion_factor
        for s in range(cfg.num_stems):
            if half:                                                          # every estimator layer: one batched f16 GEMM over the bands
                cur, lda, sa_b = x, nb * dim, dim                             # layer input [band][T][k]: band i of x sits at column offset i * dim
                for wst, bst, n_out, k_in, last in self.h_masks[s]:
                    y = ctx.empty((nb, T, n_out), torch.float32 if last else torch.float16)
                    self._bgemm_h(cur, lda, sa_b, wst, bst, y, 0, n_out, T * n_out, T, n_out, k_in, act=0 if last else 5,
                                  nvec=self.h_nout if last else None)
                    cur, lda, sa_b = y, n_out, T * n_out
                col_a, col_g = self._padded_mask_cols(T)
                masked = ctx.empty((1, 4, Fq, T))
                ctx.check(lib.alsep_roformer_mask(h, _lib.ptr(spec), _lib.ptr(cur), _lib.ptr(self.occ_start), _lib.ptr(col_a), _lib.ptr(col_g),
                                                  _lib.ptr(masked), Fq, T, self.h_nmax), "alsep_roformer_mask")
                plan.istft_strided(masked, _lib.LAYOUT_REF, out[s], L, 2 * L, 0, L, L)
                continue
            hm = ctx.empty((T, self.H))
            for i in range(nb):
                cur_ptr, cur_stride = x.data_ptr() + 4 * i * dim, nb * dim
                nl = len(self.masks[s][i])
                for j, lin in enumerate(self.masks[s][i]):
                    if j + 1 < nl:
                        tmp = ctx.empty((T, lin.out))
                        self._gemm(cur_ptr, (0, 0, cur_stride, 1), lin, tmp.data_ptr(), (0, 0, lin.out, 1), T, act=5)
                        cur_ptr, cur_stride, keep = tmp.data_ptr(), lin.out, tmp
                    else:
                        self._gemm(cur_ptr, (0, 0, cur_stride, 1), lin, hm.data_ptr() + 4 * 4 * int(self.band_off[i]), (0, 0, self.H, 1), T)
            masked = ctx.empty((1, 4, Fq, T))
            ctx.check(lib.alsep_roformer_mask(h, _lib.ptr(spec), _lib.ptr(hm), _lib.ptr(self.occ_start), _lib.ptr(self.col_a), _lib.ptr(self.col_g),
                                              _lib.ptr(masked), Fq, T, self.H), "alsep_roformer_mask")
            plan.istft_strided(masked, _lib.LAYOUT_REF, out[s], L, 2 * L, 0, L, L)
        _ = hidden
        return out

    __call__ = forward


def view_on_stream(net, ctx: Context):
    """A view of a network object that launches on another context (= another HIP stream of the same device): shared read-only
    weights, its own per-call caches and scratch (``_plans``, ``_pos``, ``_ws``, ``_cws``, and the lazily built device tables ``_rot`` /
    ``_mask_cols``, which are written by a kernel on the stream that first needs them: a view sharing them could read a table another
    lane's stream has not finished writing, and two lanes sharing a split-K workspace would overwrite each other's partial sums)."""
    import copy
    if ctx.device != net.ctx.device:
        raise AlsepError("view_on_stream: the other context must be on the same device")
    v = copy.copy(net)
    v.ctx = ctx
    for name, fresh in (("_plans", {}), ("_pos", {}), ("_ws", None), ("_cws", None), ("_rot", {}), ("_mask_cols", {})):
        if hasattr(v, name):
            setattr(v, name, fresh)
    return v


class RoformerRunner:
    """chunked inference (``demix_track`` of the training project the checkpoints come from): mix [2, L] -> {label: [2, L]}"""

    def __init__(self, net, labels: Tuple[str, ...], lanes: Optional[int] = None, graphs: Optional[bool] = None, sharded: bool = False,
                 group=None, contraction: str = "exact"):
        """``sharded=True``: the chunks of a track are split into contiguous ranges over the ranks of ``group`` (torch.distributed; one
        process per GPU); see ``demix``.  ``net``: a Roformer, or any network of the same training project with ``cfg.chunk_size / num_overlap / num_stems`` and
        ``forward([2, chunk]) -> [num_stems, 2, chunk]`` (MDX23C).  ``lanes``: chunks in flight at once, each on a HIP stream of its
        own (default ``ALSEP_RUNNER_LANES`` or 4 on a GPU, 1 elsewhere); per-lane weighted sums are added at the end.
        ``graphs`` (default ``ALSEP_RUNNER_GRAPH`` or on, GPU only): a chunk is ~100-600 small launches issued from Python, which the host
        cannot issue as fast as the GPU retires them; every lane therefore captures ONE chunk forward into a HIP graph (its launches go
        to the lane's stream, which is the capturing stream) and replays it per chunk: static input / output buffers, one graph launch."""
        self.net, self.ctx, self.labels = net, net.ctx, labels
        self.sharded, self.group = bool(sharded), group
        # ``contraction="split"`` (float32 networks, opt-in): their convolutions / GEMMs as split-half products on the f16 matrix pipe for
        # the duration of a track (csrc/nn_f32s.h), on ONE lane; a track during which an operand left the half range is run again on the
        # exact kernels.  With four lanes the split kernels (f16 MFMA) corrupt the other lanes' FFT launches: Mel-Band float32 120 s
        # 2.72 -> 2.33 s but 7.6e-3 off at peak 0.10 and not reproducible (profiles/r04_contraction_ab.txt)
        if contraction not in ("split", "exact"):
            raise AlsepError("contraction must be 'split' or 'exact'")
        self.contraction = contraction
        if contraction == "split":
            lanes = 1          # f16 MFMA kernels on every lane's stream: the one-lane rule of the half-precision networks applies (below)
        if len(labels) != net.cfg.num_stems:
            raise AlsepError("one label per stem")
        import os
        gpu = self.ctx.device.type == "cuda"
        if graphs is None:
            graphs = os.environ.get("ALSEP_RUNNER_GRAPH", "1") != "0"
        self.graphs = bool(graphs) and gpu
        # Half-precision networks run on ONE lane, with no override.  (1) With graph replay one lane is as fast as four (Mel-Band 120 s: 576 vs
        # 564 ms, BS 1 101 vs 1 100, MDX23C 1 175 vs 1 247).  (2) FFT launches must not share the GPU with 16-bit MFMA kernels of another
        # stream: beside ANY kernel that issues f16 / bf16 MFMA and leaves room on its SIMDs -- a neutral 60-line GEMM does it 20 times out
        # of 20 -- the packed-f32 FFT kernels return aligned 16-lane groups of slightly wrong values; f32 MFMA and VALU kernels do not do
        # it, and the same FFT source compiled without packed arithmetic is immune (profiles/r04_cross_stream_discrimination.txt,
        # scripts/dbg/discriminate.py).  The float32 modes (f32 MFMA only) keep their lanes.
        half = getattr(net, "precision", "f32") == "f16" or bool(getattr(net, "half", False))
        if lanes is None:
            lanes = (1 if half else int(os.environ.get("ALSEP_RUNNER_LANES", "4"))) if gpu else 1
        elif half and gpu and int(lanes) > 1:
            import logging
            logging.getLogger(__name__).warning("RoformerRunner: %d lanes asked for a half-precision network; using 1 (FFT launches beside the "
                                                "16-bit MFMA kernels of another stream come back corrupted on this stack)", int(lanes))
            lanes = 1
        self.lanes = max(1, int(lanes)) if gpu else 1
        self._lane_nets: List[tuple] = []
        self._graphs: Dict[int, list] = {}                     # lane index -> [(graph, static input, static output)] x graphs_per_lane
        self._graph_turn: Dict[int, int] = {}
        self.graphs_per_lane = max(1, int(os.environ.get("ALSEP_RUNNER_GRAPHS_PER_LANE", "2")))

    def _lanes(self):
        if not self._lane_nets:
            if self.graphs:                                    # capture needs a non-default stream: every lane gets its own
                for _ in range(self.lanes):
                    st = torch.cuda.Stream(device=self.ctx.device)
                    self._lane_nets.append((view_on_stream(self.net, Context(self.ctx.device, stream=st.cuda_stream)), st))
            else:
                self._lane_nets = [(self.net, None)]
                for _ in range(1, self.lanes):
                    st = torch.cuda.Stream(device=self.ctx.device)
                    self._lane_nets.append((view_on_stream(self.net, Context(self.ctx.device, stream=st.cuda_stream)), st))
        return self._lane_nets

    def _forward_chunk(self, k: int, lane_net, st, chunk: torch.Tensor) -> torch.Tensor:
        """the network on one [2, Cn] chunk on lane k's stream: eagerly, or -- from the lane's second chunk on -- as a replay of the HIP
        graph captured on that stream"""
        if not self.graphs:
            return lane_net.forward(chunk)
        # TWO captures per lane, replayed in turn: a graph launched again while its previous launch is still running makes the host wait
        # for that launch first (ROCm 7.2), so with one capture the GPU idles for the host's per-chunk work (input copy, ~110 packets of
        # the launch itself: 3.8 of 13.2 ms per Mel-Band chunk, profiles/r04_mel_half_trace_gaps.txt); with two, chunk n + 1 is queued
        # while chunk n runs.  Same stream: the replays still execute one after the other.
        slots = self._graphs.setdefault(k, [])
        if len(slots) < self.graphs_per_lane:
            y = lane_net.forward(chunk)                         # eager first: builds plans, tables and kernel attributes outside a capture
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

    def demix(self, mix: torch.Tensor) -> torch.Tensor:
        half = getattr(self.net, "precision", "f32") == "f16" or bool(getattr(self.net, "half", False))
        if self.contraction != "split" or half:                  # (the f16 networks keep their few float32 contractions exact)
            return self._demix(mix)
        ctxs = [self.ctx] + [ln.ctx for ln, _ in self._lanes() if ln.ctx is not self.ctx]
        try:
            for c in ctxs:
                c.set_nn_contraction(True)
            out = self._demix(mix)
            exceeded = [c.nn_range_exceeded() for c in ctxs]
        finally:
            for c in ctxs:
                c.set_nn_contraction(False)
        if any(exceeded):
            import logging
            logging.getLogger(__name__).warning("RoformerRunner: an operand left the half range (|x| > 65504) during this track -- running it "
                                                "again on the exact float32 kernels")
            self._graphs.clear()                               # captured with the split kernels
            out = self._demix(mix)
        return out

    def _demix(self, mix: torch.Tensor) -> torch.Tensor:
        """Sharded: rank r runs a contiguous range of the chunks and owns the output samples from its first chunk's start to
        the next rank's.  Its chunks reach up to ``chunk - step`` samples beyond that span: those seam sums of all ranks travel in one
        small all-gather and every rank adds the pieces that fall into its span; it then divides its span by the summed window weights,
        and ONE all-gather of the finished stem segments gives every rank the track -- no full-length all-reduce."""
        ctx, net = self.ctx, self.net
        cfg = net.cfg
        lib, h = ctx.lib, ctx.handle
        mix = mix.contiguous().float()
        L0 = mix.shape[-1]
        Cn = cfg.chunk_size
        step = Cn // cfg.num_overlap
        border = Cn - step
        padded = L0 > 2 * border and border > 0
        if padded:
            buf = ctx.empty((2, L0 + 2 * border))
            ctx.check(lib.alsep_nn_reflect_pad(h, _lib.ptr(mix), _lib.ptr(buf), 2, L0, border, border), "alsep_nn_reflect_pad")
            mix = buf
        total = mix.shape[-1]
        fade = Cn // 10
        fin, fout = torch.linspace(0, 1, fade), torch.linspace(1, 0, fade)
        w_start, w_mid, w_fin = torch.ones(Cn), torch.ones(Cn), torch.ones(Cn)
        w_start[-fade:] *= fout
        w_fin[:fade] *= fin
        w_mid[-fade:] *= fout
        w_mid[:fade] *= fin
        wins = [w.to(ctx.device) for w in (w_start, w_mid, w_fin)]
        S = cfg.num_stems
        starts = list(range(0, total, step))
        rank, world = 0, 1
        if self.sharded:
            import torch.distributed as tdist
            rank, world = tdist.get_rank(self.group), tdist.get_world_size(self.group)
        from . import dist as adist
        c_lo, c_hi = adist.window_range(len(starts), world, rank)
        mine = starts[c_lo:c_hi]
        lanes = self._lanes()[: max(1, min(self.lanes, len(mine)))]
        results = [ctx.zeros((S * 2, total)) for _ in lanes]     # one weighted sum per lane
        counter = torch.zeros(total)                             # every rank needs the whole counter (host arithmetic, no model)
        for i in starts:
            length = min(Cn, total - i)
            kind = 0 if i == 0 else (2 if i + step >= total else 1)
            counter[i:i + length] += (w_start, w_mid, w_fin)[kind][:length]
        main = torch.cuda.current_stream(ctx.device) if (len(lanes) > 1 or self.graphs) else None
        for _, st in lanes:
            if st is not None:
                st.wait_stream(main)

        def run_chunk(k, lane_net, st, res, i):
            lctx = lane_net.ctx
            length = min(Cn, total - i)
            part = mix[:, i:i + length]
            if length < Cn:
                chunk = lctx.zeros((2, Cn))
                if length > Cn // 2 + 1:                                   # F.pad(mode="reflect") on the right
                    piece = part.contiguous()
                    lctx.check(lctx.lib.alsep_nn_reflect_pad(lctx.handle, _lib.ptr(piece), _lib.ptr(chunk), 2, length, 0, Cn - length),
                               "alsep_nn_reflect_pad")
                else:
                    chunk[:, :length] = part
            else:
                chunk = part.contiguous()
            y = self._forward_chunk(k, lane_net, st, chunk)                  # [S, 2, Cn]
            kind = 0 if i == 0 else (2 if i + step >= total else 1)
            lctx.check(lctx.lib.alsep_nn_vec_fma(lctx.handle, C.c_void_p(res.data_ptr() + 4 * i), _lib.ptr(y), _lib.ptr(wins[kind]), S * 2,
                                                 length, total, Cn), "alsep_nn_vec_fma")

        for n, i in enumerate(mine):
            k = n % len(lanes)
            lane_net, st = lanes[k]
            if st is None:
                run_chunk(k, lane_net, st, results[0], i)
            else:
                with torch.cuda.stream(st):
                    run_chunk(k, lane_net, st, results[k], i)
        result = results[0]
        for k, (_, st) in enumerate(lanes):
            if st is not None:
                main.wait_stream(st)
            if k > 0:
                ctx.check(lib.alsep_axpby(h, 1.0, _lib.ptr(results[k]), 1.0, _lib.ptr(result), result.numel()), "alsep_axpby")
        cnt = counter.to(ctx.device)
        if world > 1:
            # own spans: from a rank's first chunk start to the next rank's (the last rank with chunks: to the end).  window_range deals
            # the chunks so that ranks WITHOUT any (tracks of fewer chunks than ranks) are the last ones: they own the empty span at the end
            bounds = [adist.window_range(len(starts), world, q) for q in range(world)]
            m = sum(1 for b in bounds if b[1] > b[0])
            ranges = []
            for q in range(world):
                if q >= m:
                    ranges.append((total, total))
                else:
                    ranges.append((starts[bounds[q][0]], starts[bounds[q + 1][0]] if q + 1 < m else total))
            own_lo, own_hi = ranges[rank]
            # seam: what this rank's chunks wrote beyond its own span (at most `border` samples), for every rank in one all-gather
            seam = max(border, 1)
            tail = ctx.zeros((S * 2, seam))
            t_lo = own_hi
            t_hi = min(total, (mine[-1] + Cn) if mine else own_hi)
            if t_hi > t_lo:
                tail[:, : t_hi - t_lo] = result[:, t_lo:t_hi]
            tails = adist.all_gather_fixed(tail, self.group)                       # [world, S * 2, seam]
            for q in range(world):
                if q == rank or bounds[q][1] <= bounds[q][0]:
                    continue
                q_lo = ranges[q][1]                                                # rank q's tail starts where its span ends
                q_hi = min(total, starts[bounds[q][1] - 1] + Cn)
                a, b = max(q_lo, own_lo), min(q_hi, own_hi)
                if b > a:
                    piece = tails[q][:, a - q_lo: b - q_lo].contiguous()
                    dst = result[:, a:b]
                    result[:, a:b] = dst + piece
            seg = result[:, own_lo:own_hi].contiguous()
            if own_hi > own_lo:
                ctx.check(lib.alsep_nn_vec_div(h, _lib.ptr(seg), _lib.ptr(cnt[own_lo:own_hi].contiguous()), S * 2, own_hi - own_lo), "alsep_nn_vec_div")
            result = adist.all_gather_ranges(seg, ranges, total, self.group)       # ONE all-gather of the finished stem segments
        else:
            ctx.check(lib.alsep_nn_vec_div(h, _lib.ptr(result), _lib.ptr(cnt), S * 2, total), "alsep_nn_vec_div")
        out = result.view(S, 2, total)
        return out[..., border:border + L0].contiguous() if padded else out

    def separate(self, mix: torch.Tensor) -> Dict[str, torch.Tensor]:
        out = self.demix(mix)
        return {label: out[i] for i, label in enumerate(self.labels)}


# ---- synthetic weights (data only; bench / tests, allow_synthetic=True) -----------------------------------------------------------
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

    bands = band_indices(cfg)
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
