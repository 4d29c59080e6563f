#!/usr/bin/env python3
"""Golden vectors for the VR-architecture network (TEST INFRASTRUCTURE).

Imports the REFERENCE modules modules/rvc/infer/lib/uvr5_pack/lib_v5/nets.py / nets_61968KB.py (+ layers*.py) in this
container with ``librosa`` / ``soundfile`` stubbed (spec_utils.py imports them at module scope; the network classes do
not use them), loads the seeded random state_dict of audiolab_amd.vrnet.random_state_dict into ``CascadedASPPNet`` (eval
mode) and records its output on a seeded random magnitude spectrogram.  Weights are regenerated from the seed by the
tests, so the fixture holds only inputs / outputs: tests/golden/vrnet.npz.

    python oracle/make_golden_vr.py       (only where /root/reference exists)
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from audiolab_amd.vrnet import WIDTHS, random_state_dict, random_state_dict_new  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def load_ref_nets(variant: str):
    for name in ("librosa", "soundfile"):
        sys.modules.setdefault(name, types.ModuleType(name))
    pkg_dir = os.path.join(REF, "modules/rvc/infer/lib/uvr5_pack/lib_v5")
    pkg = types.ModuleType("ref_lib_v5")
    pkg.__path__ = [pkg_dir]
    sys.modules["ref_lib_v5"] = pkg
    if variant == "nets":                                     # nets.py does a top-level "import layers"
        sys.modules["layers"] = importlib.import_module("ref_lib_v5.layers")
    return importlib.import_module(f"ref_lib_v5.{variant}")


def main():
    out = {}
    cases = [("nets", 128, 48, 11, None), ("nets", 128, 32, 12, {"split_bin": 24, "value": 0.3}),
             ("nets_61968KB", 64, 32, 13, None)]
    for k, (variant, n_fft, frames, seed, aggr) in enumerate(cases):
        mod = load_ref_nets(variant)
        net = mod.CascadedASPPNet(n_fft)
        sd = random_state_dict(WIDTHS[variant], seed=seed)
        missing, unexpected = net.load_state_dict(sd, strict=False)
        assert not unexpected, unexpected
        assert all(m.endswith("num_batches_tracked") for m in missing), missing
        net.eval()
        g = torch.Generator().manual_seed(100 + seed)
        x = torch.rand((2, 2, n_fft // 2 + 1, frames), generator=g) * 3.0
        with torch.no_grad():
            y = net.forward(x, aggr)
        out[f"c{k}_cfg"] = np.array([n_fft, frames, seed, -1 if aggr is None else aggr["split_bin"]], dtype=np.int64)
        out[f"c{k}_aggr"] = np.array([-1.0 if aggr is None else aggr["value"]], dtype=np.float64)
        out[f"c{k}_variant"] = np.array(variant)
        out[f"c{k}_x"] = x.numpy()
        out[f"c{k}_y"] = y.numpy()
        print(variant, n_fft, frames, "out", tuple(y.shape), "peak", float(y.abs().max()))
    # the VR runner (utils.py:25-100 ``inference``: normalise, pad, window, predict, concat, optional TTA) with the same net;
    # ``offset`` is an attribute of the module (128 in the reference): 8 here so that the windows stay small
    spec = importlib.util.spec_from_file_location("ref_uvr5_utils", os.path.join(REF, "modules/rvc/infer/lib/uvr5_pack/utils.py"))
    utils = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(utils)
    utils.tqdm = lambda it: it
    mod = load_ref_nets("nets")
    net = mod.CascadedASPPNet(64)
    net.load_state_dict(random_state_dict(WIDTHS["nets"], seed=21), strict=False)
    net.eval()
    net.offset = 8
    g = torch.Generator().manual_seed(77)
    xs = (torch.randn((2, 33, 70), generator=g) + 1j * torch.randn((2, 33, 70), generator=g)).numpy().astype(np.complex64)
    for tag, tta, aggr in (("plain", False, None), ("tta", True, {"split_bin": 12, "value": 0.2})):
        pred, x_mag, phase = utils.inference(xs, "cpu", net, aggr, {"window_size": 48, "tta": tta})
        out[f"inf_{tag}_pred"] = pred.astype(np.float32)
        print("inference", tag, pred.shape, float(np.abs(pred).max()))
    # nets_new.CascadedNet (LSTM branch): predict = x * mask before the offset crop; offset 0 for a small fixture
    modn = load_ref_nets("nets_new")
    for k, (n_fft, nout, nout_lstm, frames, seed) in enumerate([(128, 16, 64, 32, 31), (64, 32, 128, 48, 32)]):
        netn = modn.CascadedNet(n_fft, nout=nout, nout_lstm=nout_lstm)
        missing, unexpected = netn.load_state_dict(random_state_dict_new(n_fft, nout, nout_lstm, seed=seed), strict=False)
        assert not unexpected and all(m.endswith("num_batches_tracked") for m in missing), (missing, unexpected)
        netn.eval()
        netn.offset = 0
        g = torch.Generator().manual_seed(200 + seed)
        x = torch.rand((2, 2, n_fft // 2 + 1, frames), generator=g) * 3.0
        with torch.no_grad():
            y = netn.predict(x)
        out[f"n{k}_cfg"] = np.array([n_fft, nout, nout_lstm, frames, seed], dtype=np.int64)
        out[f"n{k}_x"], out[f"n{k}_y"] = x.numpy(), y.numpy()
        print("nets_new", n_fft, nout, nout_lstm, tuple(y.shape), float(y.abs().max()))
    out["inf_x"] = xs
    out["inf_mag"], out["inf_phase"] = x_mag.astype(np.float32), phase.astype(np.complex64)
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "vrnet.npz"), **out)
    print(os.path.getsize(os.path.join(OUT, "vrnet.npz")), "bytes")


if __name__ == "__main__":
    main()
