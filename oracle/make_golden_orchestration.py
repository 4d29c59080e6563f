#!/usr/bin/env python3
"""Golden vectors for the orchestration stages a12 / a13 of SURVEY.md section 8 (TEST INFRASTRUCTURE).

Runs the REFERENCE's own stage methods -- modules/separator/stem_separator.py ``_apply_transform_chain`` (:777-840),
``_apply_bg_vocal_splitting`` (:737-775), ``_advanced_drum_separation_all`` (:534-587), ``_woodwinds_separation_all``
(:589-623), ``_multistem_separation_all`` (:459-503), ``_should_apply_transform`` (:680-699), ``_rename_file`` (:702-735) -- imported in this container with the absent
third-party packages stubbed (as oracle/make_golden.py), against a FAKE separator: ``load_model`` / ``separate`` with
deterministic toy "models" (gain + shift per output label, oracle/toy.py ``toy_model_outputs``) and an in-memory file
system behind the stubbed ``soundfile.write`` / ``librosa.load`` (PCM_16 temp files are quantised like libsndfile does).
What is pinned is the orchestration: which model runs on which stem, which output is picked by label, the residual
subtraction bookkeeping, the fallbacks.  Output: tests/golden/orchestration.npz + orchestration.json.

    python oracle/make_golden_orchestration.py       (only where /root/reference exists)
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.make_golden import load_ref_stem_separator  # noqa: E402
from oracle.toy import synth_mix, toy_model_outputs  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FS = {}                                                   # path -> (array [C,N] float32, sr)


def sf_write(path, data, sr, format=None, subtype=None):
    x = np.asarray(data, dtype=np.float32).T.copy()      # soundfile takes [N,C]
    if subtype == "PCM_16":
        x = (np.clip(np.rint(x.astype(np.float64) * 32768.0), -32768, 32767) / 32768.0).astype(np.float32)
    FS[path] = (x, sr)


def librosa_load(path, sr=None, mono=False):
    x, fsr = FS[path]
    return x.copy(), fsr


class FakeSeparator:
    """The six members of audio_separator's Separator that the orchestrator touches."""

    def __init__(self):
        self.output_dir = None
        self.model_instance = types.SimpleNamespace(output_dir=None)
        self.model = None
        self.calls = []

    def load_model(self, name):
        self.model = name

    def separate(self, path):
        x, sr = FS[path]
        self.calls.append((self.model, os.path.basename(path)[:4]))
        base = os.path.splitext(os.path.basename(path))[0]
        tag = os.path.splitext(self.model)[0]
        names = []
        for label, y in toy_model_outputs(self.model, x):
            name = f"{base}_({label})_{tag}.wav"
            FS[os.path.join(self.output_dir, name)] = (y.astype(np.float32), sr)
            names.append(name)
        return names


def make_engine(ss, **opts):
    eng = object.__new__(ss.EnsembleDemucsMDXMusicSeparationModel)      # bypass downloads / device setup (:96-158)
    eng.separator = FakeSeparator()
    eng.reverb_removal = opts.get("reverb_removal", "Nothing")
    eng.echo_removal = opts.get("echo_removal", "Nothing")
    eng.crowd_removal = opts.get("crowd_removal", "Nothing")
    eng.noise_removal = opts.get("noise_removal", "Nothing")
    eng.delay_removal_model = "dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt"
    eng.noise_removal_model = "UVR-DeNoise.pth"
    eng.crowd_removal_model = "UVR-MDX-NET_Crowd_HQ_1.onnx"
    eng.store_reverb_ir = False
    eng.callback = None
    eng.global_step = 0
    eng.total_steps = 0
    return eng


def main():
    ss = load_ref_stem_separator()
    ss.sf.write = sf_write
    ss.librosa.load = librosa_load
    os.path.exists_orig = os.path.exists
    ss.os.path.exists = lambda p: p in FS or os.path.exists_orig(p)
    ss.os.remove = lambda p: FS.pop(p, None)
    ss.os.rename = lambda a, b: FS.__setitem__(b, FS.pop(a))
    out, meta = {}, {}
    sr, n = 44100, 20000
    folder = "/mem"
    vocals = synth_mix(n, seed=301) * np.float32(0.6)
    inst = synth_mix(n, seed=302) * np.float32(0.7)

    # pure string logic
    E = ss.EnsembleDemucsMDXMusicSeparationModel
    meta["should_apply"] = [[stem, setting, bool(E._should_apply_transform(stem, setting))]
                            for stem in ("(vocals)", "(Vocals)", "(bg_vocals)", "(BG_Vocals)", "(instrumental)", "(Main Vocals)")
                            for setting in ("Nothing", "All", "All Vocals", "Main Vocals", "bogus")]
    names = ["/o/tmp_ab_(No Reverb)_dereverb_mel_band_roformer_anvuew_sdr_19.1729.wav", "/o/tmp_(Vocals)_(No Crowd)_UVR-MDX-NET_Crowd_HQ_1.wav",
             "/o/x_(dry)_dereverb-echo_mel_band_roformer_sdr_13.4843_v2.wav", "/o/x_(No Noise)_UVR-DeNoise.wav",
             "/o/song_(Instrumental)_model_bs_roformer.wav", "/o/a (b) (c)_(Kick)_MDX23C-DrumSep.wav"]
    meta["rename"] = [[base, f, E._rename_file(base, f)] for base in ("song.wav", "/in/My Track.flac") for f in names]

    # transform chain under several settings (a13)
    cases = {"rev_main": dict(reverb_removal="Main Vocals"),
             "all_four": dict(reverb_removal="All Vocals", crowd_removal="All", noise_removal="All Vocals", echo_removal="All"),
             "noise_main": dict(noise_removal="Main Vocals")}
    for tag, opts in cases.items():
        for label, arr in (("vocals", vocals), ("instrumental", inst), ("bg_vocals", vocals)):
            for skip in (None, ["No Reverb"]):
                eng = make_engine(ss, **opts)
                FS.clear()
                got = eng._apply_transform_chain(arr.copy(), sr, "song", label, folder, skip_transforms=skip)
                key = f"chain_{tag}_{label}_{'skip' if skip else 'full'}"
                out[key] = np.asarray(got, dtype=np.float32)
                meta[key] = {"opts": opts, "calls": [c[0] for c in eng.separator.calls], "steps": eng.global_step}
    # BG vocal split
    eng = make_engine(ss)
    FS.clear()
    main_v, bg_v = eng._apply_bg_vocal_splitting(vocals.copy(), sr, "song", folder)
    out["bg_main"], out["bg_bg"] = np.asarray(main_v, np.float32), np.asarray(bg_v, np.float32)
    eng = make_engine(ss)
    FS.clear()
    main_v, bg_v = eng._apply_bg_vocal_splitting(np.zeros_like(vocals), sr, "song", folder)       # empty background -> fallback
    meta["bg_silent_fallback"] = bool(bg_v is None and float(np.abs(main_v).max()) == 0.0)

    # drum kit and woodwinds (a12)
    drums = synth_mix(n, seed=303) * np.float32(0.5)
    other = synth_mix(n, seed=304) * np.float32(0.5)
    eng = make_engine(ss)
    FS.clear()
    results = {"song": {"sr": sr, "instrumental": inst.copy(), "drums": drums.copy(), "other": other.copy(), "bass": None,
                        "output_folder": folder}}
    eng._advanced_drum_separation_all(results)
    r = results["song"]
    for k in ("drums_kick", "drums_snare", "drums_toms", "drums_hh", "drums_ride", "drums_crash", "drums_other", "bass", "guitar"):
        out[f"drum_{k}"] = np.asarray(r[k], np.float32)
    eng._woodwinds_separation_all(results)
    out["ww_woodwinds"], out["ww_other"] = np.asarray(r["woodwinds"], np.float32), np.asarray(r["other"], np.float32)
    meta["drum_ww_steps"] = eng.global_step
    # 6-stem stage on the full mix (a10 orchestration: label mapping only; the model is a toy)
    eng = make_engine(ss)
    FS.clear()
    mix = (vocals + inst).astype(np.float32)
    results = {"song": {"sr": sr, "mix_np": mix, "instrumental": inst.copy(), "output_folder": folder}}
    eng._multistem_separation_all(results)
    for k in ("drums", "bass", "guitar", "piano", "other"):
        out[f"multi_{k}"] = np.asarray(results["song"][k], np.float32)
    meta["inputs"] = {"n": n, "sr": sr, "vocals": [301, 0.6], "inst": [302, 0.7], "drums": [303, 0.5], "other": [304, 0.5],
                      "note": "synth_mix(n, seed) * gain; arrays in the npz are decimated [:, ::8]"}
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "orchestration.npz"), **{k: v[:, ::8].copy() for k, v in out.items()})
    json.dump(meta, open(os.path.join(OUT, "orchestration.json"), "w"), indent=1, sort_keys=True)
    print("wrote", len(out), "arrays;", os.path.getsize(os.path.join(OUT, "orchestration.npz")), "bytes")


if __name__ == "__main__":
    main()
