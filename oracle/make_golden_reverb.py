#!/usr/bin/env python3
"""Generate tests/golden/reverb.npz by RUNNING THE REFERENCE's ``handlers/reverb.py`` in this container.

Run from any scratch directory:  ``python /root/repo/oracle/make_golden_reverb.py``
Needs /root/reference (read-only).  The reference never travels: only seeds / small inputs and the reference's OUTPUTS are stored.

What is imported from the reference (third-party imports it does not need here are stubbed: pydub, soundfile, audio_separator):
  * handlers/reverb.py   fft_xcorr (:55-66), estimate_rt60 (:69-91), wiener_deconvolution (:94-106), extract_reverb (:112-172)
``extract_reverb`` reads its two inputs with ``read_audio`` (pydub decode -> float32 [N, C] in -1..1); here that one function is
replaced by a reader that hands over the arrays of the case (the same float32 [N, C] layout), everything after it is the reference's code.
"""
from __future__ import annotations

import importlib.util
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))

from oracle.reverb_cases import CASES, make_case  # noqa: E402


def _stub(name: str, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []
    sys.modules[name] = m
    return m


def load_ref_reverb():
    class _Dummy:
        def __init__(self, *a, **k):
            pass
    _stub("soundfile")
    _stub("audio_separator")
    _stub("audio_separator.separator", Separator=_Dummy)
    _stub("pydub", AudioSegment=_Dummy)
    _stub("handlers")
    _stub("handlers.config", output_path="/tmp")
    spec = importlib.util.spec_from_file_location("ref_reverb", os.path.join(REF, "handlers/reverb.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_ref_reverb()
    out = {}
    for name in CASES:
        dry, wet, sr = make_case(name)                       # float32 [N, C]
        feed = {"dry": (dry, sr), "wet": (wet, sr)}
        ref.read_audio = lambda path: feed[path]
        # With numpy >= 2.0 -- the reference pins numpy==2.0.2 (setup.sh:93, requirements.txt:56) -- np.fft keeps float32 inputs in single
        # precision, so ``early_reflection_ratio`` etc. are np.float32 and the reference's own json.dump (reverb.py:39-41) raises
        # "Object of type float32 is not JSON serializable" (the call site logs it, stem_separator.py:828-829, and a truncated file is
        # left behind).  The values the reference COMPUTED are captured here by replacing that one writer; nothing else is touched.
        captured = {}
        ref.save_params_to_file = lambda params, path: captured.update(params)
        ref.extract_reverb("dry", "wet", "unused")
        p = {k: (v if isinstance(v, (list, int)) else float(v)) for k, v in captured.items()}
        out[f"{name}_dtypes"] = np.array([str(np.asarray(captured[k]).dtype) for k in ("early_reflection_ratio", "diffusion", "spectral_centroid")])
        assert list(p) == ["sample_rate", "pre_delay", "decay_time", "early_reflection_ratio", "late_reverb_ratio", "diffusion",
                           "spectral_centroid", "impulse_response"]
        ir = np.asarray(p["impulse_response"], dtype=np.float64)
        out[f"{name}_scalars"] = np.array([p["sample_rate"], p["pre_delay"], p["decay_time"], p["early_reflection_ratio"],
                                           p["late_reverb_ratio"], p["diffusion"], p["spectral_centroid"]], dtype=np.float64)
        out[f"{name}_ir_len"] = np.array([len(ir)])
        out[f"{name}_ir_head"] = ir[:4096]
        idx = np.random.default_rng(5).integers(0, len(ir), size=min(2048, len(ir)))
        out[f"{name}_ir_idx"] = idx
        out[f"{name}_ir_val"] = ir[idx]
        out[f"{name}_ir_l2"] = np.array([np.sqrt(np.sum(ir ** 2))])
        # the two helper outputs on the mono signals, for the stage-wise checks
        dm, wm = ref.to_mono(dry), ref.to_mono(wet)
        corr = ref.fft_xcorr(wm, dm)
        out[f"{name}_corr_argmax"] = np.array([int(np.argmax(corr))])
        out[f"{name}_corr_probe"] = corr[np.linspace(0, len(corr) - 1, 64).astype(np.int64)]
        print(name, {k: v for k, v in p.items() if k != "impulse_response"}, "ir", len(ir))
    np.savez_compressed(os.path.join(OUT, "reverb.npz"), **out)
    print("wrote", os.path.join(OUT, "reverb.npz"))


if __name__ == "__main__":
    main()
