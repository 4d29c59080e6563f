#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in this container.

Run from any scratch directory:  ``python /root/repo/oracle/make_golden.py``
Needs /root/reference (read-only).  The reference never travels: only the input seeds /
small inputs and the reference's OUTPUTS are stored.  Nothing here is imported by tests.

What is imported from the reference (third-party imports stubbed, as SURVEY.md 8c):
  * modules/rvc/infer/modules/uvr5/mdxnet.py   ConvTDFNetTrim.stft/istft (:41-75),
                                                Predictor.demix/demix_base (:109-197)
  * modules/separator/stem_separator.py        _blend_tracks (:241-262),
                                                _residual_subtract (:173-239)
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))

from oracle.toy import toy_net, toy_net_affine, synth_mix, resid_case  # noqa: E402


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # allow submodule imports
    sys.modules[name] = m
    return m


def load_ref_mdxnet():
    _stub("librosa")
    _stub("soundfile")
    spec = importlib.util.spec_from_file_location(
        "ref_mdxnet", os.path.join(REF, "modules/rvc/infer/modules/uvr5/mdxnet.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_ref_stem_separator():
    class _Dummy:  # stand-in classes for names the reference imports at module scope
        def __init__(self, *a, **k):
            pass
    _stub("librosa")
    _stub("soundfile")
    _stub("audio_separator")
    _stub("audio_separator.separator", Separator=_Dummy)
    _stub("audio_separator.separator.architectures")
    _stub("audio_separator.separator.architectures.mdx_separator", MDXSeparator=_Dummy)
    _stub("onnx")
    _stub("onnx2torch")
    _stub("onnxruntime")
    _stub("pydub", AudioSegment=_Dummy)
    _stub("gradio")
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir("/tmp")
    try:
        import modules.separator.stem_separator as ss  # noqa
    finally:
        os.chdir(cwd)
    return ss


class _FakeOrt:
    """Object with the ORT ``run(None, {"input": ndarray}) -> [ndarray]`` surface
    (mdxnet.py:170-176) wrapping a numpy callable."""

    def __init__(self, fn):
        self.fn = fn

    def run(self, _names, feed):
        return [self.fn(feed["input"])]


def make_predictor(mod, net, fn, margin, chunks, denoise):
    pred = object.__new__(mod.Predictor)                 # bypass onnxruntime import (:92)
    pred.args = types.SimpleNamespace(margin=margin, chunks=chunks, denoise=denoise)
    pred.model_ = net
    pred.model = _FakeOrt(fn)
    return pred


def probes(arr: np.ndarray, n: int, seed: int):
    flat = arr.reshape(-1)
    idx = np.random.default_rng(seed).integers(0, flat.size, size=n)
    return idx.astype(np.int64), flat[idx].copy()


def gen_mdx_small(mod):
    out = {}
    cpu = torch.device("cpu")
    # (name, n_fft, hop, dim_t_arg, dim_f): hop | n_fft, 2^a*3, 2^a*15 (n_fft/hop = 7.5)
    geoms = [("p2", 256, 64, 4, 96), ("p3", 384, 64, 4, 160), ("p15", 480, 64, 4, 192),
             ("full", 256, 64, 5, 129)]
    for name, n_fft, hop, dta, dim_f in geoms:
        net = mod.ConvTDFNetTrim(cpu, "Conv-TDF", "vocals", 11, dim_f, dta, n_fft, hop=hop)
        rng = np.random.default_rng(1000 + n_fft + dta)
        x = rng.standard_normal((3, 2, net.chunk_size)).astype(np.float32)
        spec = net.stft(torch.from_numpy(x)).numpy()
        y = net.istft(torch.from_numpy(spec)).numpy()
        s2 = rng.standard_normal(spec.shape).astype(np.float32)  # arbitrary (non-consistent) spec
        y2 = net.istft(torch.from_numpy(s2)).numpy()
        out[f"{name}_geom"] = np.array([n_fft, hop, dta, dim_f], dtype=np.int64)
        out[f"{name}_x"] = x
        out[f"{name}_spec"] = spec
        out[f"{name}_y"] = y
        out[f"{name}_s2"] = s2
        out[f"{name}_y2"] = y2
    np.savez_compressed(os.path.join(OUT, "mdx_small.npz"), **out)


def gen_mdx_real(mod):
    out = {}
    cpu = torch.device("cpu")
    for name, n_fft, dim_f in [("n6144", 6144, 3072), ("n7680", 7680, 3072), ("n4096", 4096, 2048)]:
        net = mod.ConvTDFNetTrim(cpu, "Conv-TDF", "vocals", 11, dim_f, 8, n_fft)
        seed = 7000 + n_fft
        x = np.random.default_rng(seed).standard_normal((2, 2, net.chunk_size)).astype(np.float32)
        spec = net.stft(torch.from_numpy(x)).numpy()
        y = net.istft(torch.from_numpy(spec)).numpy()
        out[f"{name}_geom"] = np.array([n_fft, 1024, 8, dim_f], dtype=np.int64)
        out[f"{name}_seed"] = np.array(seed)
        out[f"{name}_spec_lo"] = spec[:, :, :8, :4].copy()
        out[f"{name}_spec_hi"] = spec[:, :, -8:, -4:].copy()
        out[f"{name}_spec_idx"], out[f"{name}_spec_val"] = probes(spec, 8192, seed + 1)
        out[f"{name}_spec_l2"] = np.array(np.sqrt((spec.astype(np.float64) ** 2).sum()))
        out[f"{name}_spec_sum"] = np.array(spec.astype(np.float64).sum())
        out[f"{name}_y_idx"], out[f"{name}_y_val"] = probes(y, 8192, seed + 2)
        out[f"{name}_y_head"] = y[:, :, :64].copy()
        out[f"{name}_y_tail"] = y[:, :, -64:].copy()
        out[f"{name}_y_l2"] = np.array(np.sqrt((y.astype(np.float64) ** 2).sum()))
    np.savez_compressed(os.path.join(OUT, "mdx_real.npz"), **out)


def gen_demix(mod):
    out = {}
    cpu = torch.device("cpu")
    nets = {"lin": toy_net, "aff": toy_net_affine}
    # small geometry: full outputs.  chunk_size (outer) = chunks*44100 samples (mdxnet.py:111)
    net_s = mod.ConvTDFNetTrim(cpu, "Conv-TDF", "vocals", 11, 96, 5, 256, hop=64)
    cases = [("a", 100000, 1, 4410, False, "lin"), ("b", 60000, 1, 4410, True, "aff"),
             ("c", 30000, 0, 44100, False, "aff"), ("d", 50000, 1, 44100, True, "lin"),
             ("e", net_s.chunk_size - 256, 0, 100, False, "lin"),   # n % gen == 0 -> extra gen of pad
             ("f", 1, 0, 44100, False, "aff")]                       # 1-sample track
    for tag, n, chunks, margin, denoise, which in cases:
        mix = synth_mix(n, seed=300 + n + chunks)
        pred = make_predictor(mod, net_s, nets[which], margin, chunks, denoise)
        res = np.asarray(pred.demix(mix))
        out[f"small_{tag}_cfg"] = np.array([n, chunks, margin, int(denoise)], dtype=np.int64)
        out[f"small_{tag}_net"] = np.array(which)
        out[f"small_{tag}_out"] = res.astype(np.float32)
    out["small_geom"] = np.array([256, 64, 5, 96], dtype=np.int64)
    # real geometry, 30 s (config 1 size): probes only
    net_r = mod.ConvTDFNetTrim(cpu, "Conv-TDF", "vocals", 11, 3072, 8, 6144)
    n = 1323000
    mix = synth_mix(n)
    for tag, chunks, denoise, which in [("r0", 0, False, "lin"), ("r15", 15, True, "aff")]:
        pred = make_predictor(mod, net_r, nets[which], 44100, chunks, denoise)
        res = np.asarray(pred.demix(mix)).astype(np.float32)
        assert res.shape == (1, 2, n)
        out[f"real_{tag}_cfg"] = np.array([n, chunks, 44100, int(denoise)], dtype=np.int64)
        out[f"real_{tag}_net"] = np.array(which)
        out[f"real_{tag}_idx"], out[f"real_{tag}_val"] = probes(res, 16384, 99)
        out[f"real_{tag}_strided"] = res[0, :, ::2003].copy()
        gen = net_r.chunk_size - net_r.n_fft
        out[f"real_{tag}_seam"] = res[0, :, gen - 64: gen + 64].copy()     # inner window seam
        out[f"real_{tag}_seg"] = res[0, :, 15 * 44100 - 64: 15 * 44100 + 64].copy()  # outer seam
        out[f"real_{tag}_l2"] = np.array(np.sqrt((res.astype(np.float64) ** 2).sum()))
    out["real_geom"] = np.array([6144, 1024, 8, 3072], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "demix.npz"), **out)


def gen_ensemble(ss):
    out = {}
    eng = object.__new__(ss.EnsembleDemucsMDXMusicSeparationModel)
    rng = np.random.default_rng(4242)
    # blend: 3 tracks of unequal length
    tr = [rng.standard_normal((2, n)).astype(np.float32) * 0.3 for n in (5000, 4800, 5100)]
    wts = [8.6, 8.4, 8.5]
    out["blend_t0"], out["blend_t1"], out["blend_t2"] = tr
    out["blend_w"] = np.array(wts)
    out["blend_out"] = eng._blend_tracks(tr, wts)
    # residual subtract: known lag / gain
    for tag, lag, gain in [("p", 37, 0.8), ("m", -37, 0.8), ("z", 0, 1.6), ("big", 400, 0.5)]:
        base, comp = resid_case(lag, gain)
        res = eng._residual_subtract(base, comp, 44100).astype(np.float32)
        out[f"resid_{tag}_cfg"] = np.array([lag, gain])
        out[f"resid_{tag}_out"] = res if tag == "p" else res[:, ::5].copy()
    np.savez_compressed(os.path.join(OUT, "ensemble.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    mod = load_ref_mdxnet()
    gen_mdx_small(mod)
    gen_mdx_real(mod)
    gen_demix(mod)
    ss = load_ref_stem_separator()
    gen_ensemble(ss)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
