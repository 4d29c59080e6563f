"""Separation engine behind the Separate wrapper: same entry points, option names, stage order,
stem labels and progress protocol as the reference's ``modules/separator/stem_separator.py``
(``separate_music`` :949-1001, ``predict_with_model`` :847-946,
``EnsembleDemucsMDXMusicSeparationModel`` :82-840), with the arithmetic on the GPU and tensors
resident in HBM between stages (no temp PCM16 WAV per model, :57-75 / :278).

Roster (the build owns its roster; weights are synthetic offline unless the files are present): every stage of the
reference has kernels behind it -- the vocal ensemble in the reference's member order (Mel-Band / BS Roformer, MDX23C, MDX-Net; :379-387),
htdemucs_6s for the multistem stage (:466), the reverb / echo / crowd / noise transform chain (:777-840), the drum-kit split (:534-587), the
woodwinds split (:589-623) and the reverb impulse-response extraction (:822-829, audiolab_amd/reverb.py).  A stage whose model file is not in
the engine's roster (a caller-supplied roster may be narrower) is skipped with a log line instead of failing the whole job.
"""
from __future__ import annotations

import logging
import os
import shutil
import subprocess
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from audiolab_amd import ensemble, wavio
from audiolab_amd.engine import FOUR_STEM_SET, MODEL_ROSTER, Separator
from audiolab_amd.handlers import config

logger = logging.getLogger(__name__)


def ensure_wav(input_path: str, sr: int = 44100) -> str:
    """Path of a WAV rendition of ``input_path`` (:31-54): a ``.wav`` is returned as it is; anything else is transcoded ONCE to
    ``<name>_converted.wav`` (16-bit stereo at ``sr``) by ffmpeg, and that file is reused on later calls."""
    if not os.path.exists(input_path):
        raise FileNotFoundError(f"Missing file: {input_path}")
    stem, suffix = os.path.splitext(input_path)
    if suffix.lower() == ".wav":
        return input_path
    target = f"{stem}_converted.wav"
    if os.path.isfile(target):
        return target
    ffmpeg = shutil.which("ffmpeg")
    if ffmpeg is None:
        raise RuntimeError(f"{input_path}: only WAV input is supported without ffmpeg")
    subprocess.run([ffmpeg, "-y", "-i", input_path, "-acodec", "pcm_s16le", "-ac", "2", "-ar", str(sr), target],
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return target


def _call_progress(callback, frac: float, desc: str, total: int) -> None:
    """The engine calls ``callback(frac, desc, total)`` (:170); Gradio's Progress and the chain API's
    2-argument ``APIProgress`` (layouts/process.py:837-839) must both work."""
    if callback is None:
        return
    try:
        callback(frac, desc, total)
    except TypeError:
        callback(frac, desc)


class EnsembleDemucsMDXMusicSeparationModel:
    """:82-840.  Holds one ``Separator`` (weights stay resident across files and stages)."""

    # models_with_weights of the reference, in its order (:379-387): (file, vocal weight, instrumental weight)
    ENSEMBLE = [("vocals_mel_band_roformer.ckpt", 8.6, 16.0), ("model_bs_roformer_ep_368_sdr_12.9628.ckpt", 8.4, 16.0),
                ("melband_roformer_big_beta4.ckpt", 8.5, 16.0), ("MDX23C-8KFFT-InstVoc_HQ.ckpt", 7.2, 14.9),
                ("UVR-MDX-NET-Voc_FT.onnx", 6.9, 14.9), ("Kim_Vocal_2.onnx", 6.9, 14.9), ("Kim_Vocal_1.onnx", 6.8, 14.9)]

    def __init__(self, options: Dict, callback: Callable = None, separator: Optional[Separator] = None):
        self.options = options
        # The engine the reference builds at :102-107 is audio-separator's: its MDX runner normalises to 0.9, overlap-adds Hann-windowed
        # chunks (mdx_params overlap 0.25) and makes the secondary stem by spectral inversion (invert_using_spec=True, :104) -- that runner
        # is ``chunker="ola"`` here and it is the default; ``chunker="margin"`` selects the in-tree runner (mdxnet.py) instead.  The
        # knobs arrive from the wrapper's hidden inputs (wrappers/separate.py: precision / chunker / overlap / num_gpus).
        num_gpus = int(options.get("num_gpus", 1) or 1)
        if separator is None and num_gpus > 1:
            import torch.distributed as dist
            if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() == num_gpus):
                raise RuntimeError(f"num_gpus={num_gpus}: this build runs one process per GPU -- start {num_gpus} ranks with "
                                   f"torch.distributed initialised (python -m torch.distributed.run --nproc-per-node {num_gpus} ...)")
        self.separator = separator if separator is not None else Separator(
            log_level=logging.ERROR, model_file_dir=os.path.join(config.app_path, "models", "audio_separator"),
            invert_using_spec=True, use_autocast=not options.get("cpu", False) and options.get("precision", "fp16") != "fp32",
            dtype={"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}.get(options.get("precision", "fp16")),
            chunker=options.get("chunker", "ola"), overlap=float(options.get("overlap", 0.25)), normalization_threshold=0.9,
            sharded=num_gpus > 1)
        self.ctx = self.separator.ctx
        self.vocals_only = bool(options.get("vocals_only", False))
        self.separate_drums = bool(options.get("separate_drums", False))
        self.separate_woodwinds = bool(options.get("separate_woodwinds", False))
        self.alt_bass_model = bool(options.get("alt_bass_model", False))
        self.reverb_removal = options.get("reverb_removal", "Nothing")
        self.echo_removal = options.get("echo_removal", "Nothing")
        self.crowd_removal = options.get("crowd_removal", "Nothing")
        self.noise_removal = options.get("noise_removal", "Nothing")
        self.crowd_removal_model = options.get("crowd_removal_model", "UVR-MDX-NET_Crowd_HQ_1.onnx")
        self.delay_removal_model = options.get("delay_removal_model", "dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt")
        self.noise_removal_model = options.get("noise_removal_model", "UVR-DeNoise.pth")
        self.separate_bg_vocals = options.get("separate_bg_vocals", True)
        self.bg_vocal_layers = options.get("bg_vocal_layers", 1)
        self.store_reverb_ir = options.get("store_reverb_ir", False)
        self.ensemble_strength = options.get("ensemble_strength", 1)
        self.global_step = 0
        self.total_steps = 0
        self.callback = options.get("callback", None)          # overrides the ctor argument, as :158

    def _advance_progress(self, desc: str, weight: int = 1) -> None:
        self.global_step += weight
        if self.callback is not None and self.total_steps > 0:
            _call_progress(self.callback, self.global_step / self.total_steps, desc, self.total_steps)
        logger.info(f"[{self.global_step}/{self.total_steps}] {desc}")

    # -- ensemble (:357-457) --------------------------------------------------------------------
    def _separate_as_arrays_current(self, mix: torch.Tensor) -> Dict[str, torch.Tensor]:
        """:264-355 without the disk round trip: {"vocals", "instrumental"} device tensors."""
        stems = self.separator.separate_array(mix)
        out = {}
        for label, t in stems.items():
            low = f"({label})".lower()
            if "(vocals)" in low:
                out["vocals"] = t
            elif "(instrumental)" in low:
                out["instrumental"] = t
        return out

    def _ensemble_separate_all(self, files_data: List[Dict]) -> Dict[str, Dict]:
        results = {f["base_name"]: {"mix": f["mix"], "sr": f["sr"], "vocals_list": [], "instrumental_list": [],
                                    "v_weights": [], "i_weights": [], "output_folder": f["output_folder"]} for f in files_data}
        if self.ensemble_strength <= 2:                          # :389-390
            self.options["residual_blend"] = min(float(self.options.get("residual_blend", 0.4)), 0.2)
        # the first ``ensemble_strength`` members (:391) -- of those whose architecture the engine's roster knows: with the default roster
        # that is the reference's list minus the architectures still without kernels (named in the log), with a caller's roster its own models
        known = [m for m in self.ENSEMBLE if m[0] in self.separator.roster]
        for m in self.ENSEMBLE:
            if m[0] not in self.separator.roster:
                logger.warning("ensemble member '%s' is not in this engine's roster (architecture without kernels yet); skipped", m[0])
        members = known[: max(1, int(self.ensemble_strength))]
        if not members:
            raise RuntimeError("no ensemble member of the reference's list is available in this engine's roster")
        for model_name, v_wt, i_wt in members:                   # model-major: weights resident per model (:393-395)
            self.separator.load_model(model_name)
            for f in files_data:
                res = results[f["base_name"]]
                sep = self._separate_as_arrays_current(f["mix"])
                res["vocals_list"].append(sep.get("vocals", torch.zeros_like(f["mix"])))
                res["instrumental_list"].append(sep.get("instrumental", torch.zeros_like(f["mix"])))
                res["v_weights"].append(v_wt)
                res["i_weights"].append(i_wt)
                self._advance_progress(f"Ensemble model '{model_name}' processed for {f['base_name']}.")
        for res in results.values():
            res["vocals"] = ensemble.blend_tracks(self.ctx, res["vocals_list"], res["v_weights"])           # :412
            res["instrumental"] = ensemble.blend_tracks(self.ctx, res["instrumental_list"], res["i_weights"])  # :413
            res["instrumental"], _ = ensemble.debleed(self.ctx, res["mix"], res["vocals"], res["instrumental"], res["sr"],
                                                      float(self.options.get("residual_blend", 0.4)))      # :415-456
            del res["vocals_list"], res["instrumental_list"]
        return results

    # -- transform chain (:777-840), BG-vocal split (:737-775) ----------------------------------------
    REVERB_MODEL = "dereverb_mel_band_roformer_anvuew_sdr_19.1729.ckpt"     # :796
    BG_VOCAL_MODEL = "UVR-BVE-4B_SN-44100-1.pth"                            # :752
    DRUM_MODEL = "MDX23C-DrumSep-aufr33-jarredou.ckpt"                      # :541
    WOODWIND_MODEL = "17_HP-Wind_Inst-UVR.pth"                              # :596

    # which stems a transform setting covers (:680-699): "All Vocals" matches any "...vocals)" tag case-insensitively, "Main Vocals" the
    # lead only -- case-SENSITIVELY on "vocals)" as the reference tests it, so the "(Vocals)" file tag does not match but the lower-case
    # stem labels this engine passes do
    _SCOPE_TESTS = {
        "All": lambda tag: True,
        "All Vocals": lambda tag: "vocals)" in tag.lower(),
        "Main Vocals": lambda tag: "vocals)" in tag and "(bg_vocals" not in tag.lower(),
    }

    @classmethod
    def _should_apply_transform(cls, stem_name: str, setting: str) -> bool:
        test = cls._SCOPE_TESTS.get(setting)                 # "Nothing" and unknown settings apply to no stem
        return bool(test and test(stem_name))

    # model references that the reference strips from the parenthesised tags of a transformed stem's file name (:713-727)
    _MODEL_TAGS = tuple(t.lower() for t in (
        "deverb_bs_roformer", "UVR-DeEcho-DeReverb", "UVR-De-Echo-Normal", "UVR-DeNoise", "UVR-DeNoise-Lite", "mel_band_roformer", "MDX23C",
        "UVR-MDX-NET", "drumsep", "roformer", "viperx", "crowd", "karaoke", "instrumental", "_InstVoc", "_VOCFT", "NoReverb", "NoEcho",
        "NoDelay", "NoCrowd", "NoNoise", "_mel_band_roformer_karaoke_aufr33_viperx_sdr_10"))

    @classmethod
    def _rename_file(cls, base_in: str, filepath: str) -> str:
        """:702-735 -- the name a transformed stem gets: ``<input base>_`` + the parenthesised tags of ``filepath`` that name no model.
        The in-memory path writes no intermediate files, so nothing is renamed on disk; kept for callers that build names."""
        import re
        folder, name = os.path.split(filepath)
        tags = [t for t in re.findall(r"\([^)]*\)", name) if not any(m in t.lower() for m in cls._MODEL_TAGS)]
        stem = os.path.splitext(os.path.basename(base_in))[0]
        new_name = f"{stem}_{''.join(tags)}{os.path.splitext(name)[1]}".replace(") (", ")(").replace("__", "_")
        return os.path.join(folder, new_name)

    def _have(self, model_file: str, what: str) -> bool:
        if model_file in self.separator.roster:
            return True
        logger.warning("%s needs '%s', which is not in this engine's roster (architecture without kernels yet); skipped",
                       what, model_file)
        return False

    def _run_model(self, model_file: str, x: torch.Tensor):
        """load + separate on a device tensor; returns [(simulated output file name, tensor)] in the engine's output order
        (the reference matches labels against the FILE NAMES ``<base>_(<Label>)_<model>.wav``, :808-833)."""
        self.separator.load_model(model_file)
        outs = self.separator.separate_array(x)
        tag = os.path.splitext(model_file)[0]
        return [(f"tmp_(%s)_%s.wav" % (label, tag), t) for label, t in outs.items()]

    def _apply_transform_chain(self, stem: torch.Tensor, base_name: str, stem_label: str, skip_transforms=None,
                               output_folder: Optional[str] = None, sr: int = 44100) -> torch.Tensor:
        """:777-840.  Output selection as the reference: with two outputs the one whose name contains the wanted label
        (spaces removed, lower case) -- else the SECOND one; otherwise the first match, or the input unchanged.  After a de-reverb /
        de-echo transform on the vocals with ``store_reverb_ir``, the impulse response between the chosen (dry) and the other (wet)
        output is extracted into ``<output_folder>/impulse_response.ir`` (:822-829); a failure there is logged, not raised, as :828-829."""
        skip_transforms = skip_transforms or []
        chain = [(self.REVERB_MODEL, "No Reverb", self.reverb_removal), (self.delay_removal_model, "dry", self.echo_removal),
                 (self.crowd_removal_model, "No Crowd", self.crowd_removal), (self.noise_removal_model, "No Noise", self.noise_removal)]
        cur = stem
        for model_file, out_label, flag in chain:
            if out_label in skip_transforms or not self._should_apply_transform(f"({stem_label})", flag):
                continue
            if self._have(model_file, f"transform '{out_label}'"):
                outs = self._run_model(model_file, cur)
                want = out_label.replace(" ", "").lower()
                chosen = None
                if len(outs) == 2:
                    first = want in outs[0][0].replace(" ", "").lower()
                    chosen, alt = (outs[0][1], outs[1][1]) if first else (outs[1][1], outs[0][1])
                    if out_label in ("No Echo", "No Reverb") and stem_label.lower() == "vocals" and self.store_reverb_ir and output_folder:
                        try:
                            from audiolab_amd.reverb import extract_reverb
                            out_ir = os.path.join(output_folder, "impulse_response.ir")
                            logger.info(f"Extracting reverb IR for {base_name}")
                            extract_reverb(chosen, alt, out_ir, sr=sr, ctx=self.ctx)
                        except Exception as e:                       # the reference logs and carries on (:828-829)
                            logger.error(f"Error extracting IR: {e}")
                else:
                    for name, t in outs:
                        if want in name.replace(" ", "").lower():
                            chosen = t
                            break
                if chosen is not None:
                    cur = chosen
            self._advance_progress(f"TRANSFORM: {out_label} on {stem_label} for {base_name}")
        return cur

    def _apply_bg_vocal_splitting(self, vocals: torch.Tensor, base_name: str):
        """:737-775 -- the BVE model's "(Vocals)" output is the BACKGROUND, its "(Instrumental)" the main vocal."""
        if not self._have(self.BG_VOCAL_MODEL, "background-vocal split"):
            self._advance_progress("Background vocal splitting skipped.")
            return vocals, None
        outs = self._run_model(self.BG_VOCAL_MODEL, vocals)
        self._advance_progress("Background vocal splitting executed.")
        bg = main = None
        for name, t in outs:
            if "(Vocals)" in name:
                bg = t
            elif "(Instrumental)" in name:
                main = t
        if bg is not None and main is not None:
            if float(ensemble.peak_abs(self.ctx, bg.contiguous()).cpu()) > 0.0:
                return main, bg
            logger.info("Background vocals are empty after splitting.")
        return vocals, None

    # -- drum kit (:534-587), woodwinds (:589-623) -------------------------------------------------
    DRUM_PARTS = (("(kick)", "drums_kick"), ("(snare)", "drums_snare"), ("(toms)", "drums_toms"), ("(hh)", "drums_hh"),
                  ("(ride)", "drums_ride"), ("(crash)", "drums_crash"))

    def _advanced_drum_separation_all(self, results: Dict[str, Dict]) -> None:
        if not self._have(self.DRUM_MODEL, "drum-kit split"):
            for base_name in results:
                self._advance_progress(f"Advanced drum separation skipped for {base_name}.")
            return
        for base_name, res in results.items():
            drums = res.get("drums")
            if drums is None:
                drums = torch.zeros_like(res["instrumental"])
            parts = self._run_model(self.DRUM_MODEL, drums)
            drums_other = drums.clone()
            for _, key in self.DRUM_PARTS:
                res[key] = None
            for name, arrp in parts:                             # every part is subtracted, named or not (:557-561)
                low = name.lower()
                n = arrp.shape[-1]
                if n <= drums_other.shape[-1]:
                    drums_other[:, :n] = ensemble.residual_subtract(self.ctx, drums_other[:, :n].contiguous(), arrp, res["sr"])
                for tag, key in self.DRUM_PARTS:
                    if tag in low:
                        res[key] = arrp
                        break
            for key in ("drums", "bass", "guitar", "piano", "other"):
                if res.get(key) is None:
                    res[key] = torch.zeros_like(drums)
            res["drums_other"] = drums_other
            self._advance_progress(f"Advanced drum separation done for {base_name}.")

    def _woodwinds_separation_all(self, results: Dict[str, Dict]) -> None:
        if not self._have(self.WOODWIND_MODEL, "woodwinds split"):
            for base_name in results:
                self._advance_progress(f"Woodwinds separation skipped for {base_name}.")
            return
        for base_name, res in results.items():
            other = res.get("other")
            if other is None:
                other = torch.zeros_like(res["instrumental"])
            new_ww = torch.zeros_like(other)
            for name, arrw in self._run_model(self.WOODWIND_MODEL, other):
                if "(woodwinds)" in name.lower():
                    new_ww = arrw
            leftover = other.clone()
            n = new_ww.shape[-1]
            if n <= leftover.shape[-1]:
                leftover[:, :n] = ensemble.residual_subtract(self.ctx, leftover[:, :n].contiguous(), new_ww, res["sr"])
            res["woodwinds"] = new_ww
            res["other"] = leftover
            self._advance_progress(f"Woodwinds separated for {base_name}.")

    # -- multistem (:459-503), alt bass (:505-532) ---------------------------------------------------
    MULTISTEM_MODEL = "htdemucs_6s.yaml"                                     # :466

    def _multistem_separation_all(self, results: Dict[str, Dict]) -> None:
        """:459-503.  When the roster knows ``htdemucs_6s.yaml`` (a multi-stem entry) the stage runs as the reference
        does: that model on the full mix, outputs mapped to drums / bass / guitar / piano / other by substring of their
        file names (the vocals output is ignored).  Otherwise (default roster: no Demucs kernels yet) the 4-stem MDX-Net
        set stands in and yields drums / bass / other."""
        for key in ("drums", "bass", "guitar", "piano", "other"):
            for res in results.values():
                res[key] = None
        if self.MULTISTEM_MODEL in self.separator.roster:
            for base_name, res in results.items():
                for name, arr in self._run_model(self.MULTISTEM_MODEL, res["mix"]):
                    low = name.lower()
                    if "(drums)" in low or "drums" in low or "drum" in low:
                        res["drums"] = arr
                    elif "(bass)" in low or "bass" in low:
                        res["bass"] = arr
                    elif "(guitar)" in low or "guitar" in low:
                        res["guitar"] = arr
                    elif "(piano)" in low or "piano" in low:
                        res["piano"] = arr
                    elif "(other)" in low or "other" in low or "accompaniment" in low or "rest" in low:
                        res["other"] = arr
                self._advance_progress(f"6-stem separation completed for {base_name}.")
            return
        for model_name in FOUR_STEM_SET[1:]:                     # vocals output is ignored, as :491-500
            self.separator.load_model(model_name)
            label = self.separator.roster[model_name][0]
            for res in results.values():
                res[label.lower()] = self.separator.separate_array(res["mix"])[label]
        for base_name in results:
            self._advance_progress(f"Multi-stem separation completed for {base_name}.")

    def _alt_bass_separation_all(self, results: Dict[str, Dict]) -> None:
        self.separator.load_model("kuielab_a_bass.onnx")          # :512, on the INSTRUMENTAL
        for base_name, res in results.items():
            res["bass"] = self.separator.separate_array(res["instrumental"])["Bass"]
            self._advance_progress(f"Alternate bass separation done for {base_name}.")

    # -- output (:625-677) -----------------------------------------------------------------------------
    STEM_NAMES = {"vocals": "(Vocals)", "bg_vocals": "(BG_Vocals)", "instrumental": "(Instrumental)", "drums": "(Drums)",
                  "bass": "(Bass)", "guitar": "(Guitar)", "piano": "(Piano)", "woodwinds": "(Woodwinds)", "other": "(Other)",
                  "drums_kick": "(Drums_Kick)", "drums_snare": "(Drums_Snare)", "drums_toms": "(Drums_Toms)",
                  "drums_hh": "(Drums_HH)", "drums_ride": "(Drums_Ride)", "drums_crash": "(Drums_Crash)",
                  "drums_other": "(Drums_Other)"}

    def _save_all_stems(self, results: Dict[str, Dict]) -> List[str]:
        self._advance_progress("Saving all stems...")
        output_files = []
        for base_name, res in results.items():
            for stem_key, label in self.STEM_NAMES.items():
                arr = res.get(stem_key)
                if arr is None:
                    continue
                if arr.numel() == 0 or float(ensemble.peak_abs(self.ctx, arr.contiguous()).cpu()) < 1e-6:
                    continue                                     # silent stems are not written (:660-663)
                if stem_key == "bg_vocals" and "bg_vocals_" in base_name:
                    label = f"(BG_Vocals_{int(base_name.split('bg_vocals_')[-1])})"
                path = os.path.join(res["output_folder"], f"{base_name}__{label}.wav")
                wavio.write_wav(path, arr.cpu().numpy(), res["sr"], subtype="FLOAT")
                output_files.append(path)
            self._advance_progress(f"Stems saved for {base_name}.")
        for res in results.values():
            for tmp in os.listdir(res["output_folder"]):
                if tmp.startswith("tmp_"):
                    os.remove(os.path.join(res["output_folder"], tmp))
        return output_files


def predict_with_model(options: Dict, callback: Callable = None, separator: Optional[Separator] = None) -> List[str]:
    """:847-946."""
    files_data = []
    model = EnsembleDemucsMDXMusicSeparationModel(options, callback, separator=separator)
    for out_folder, input_files in options["input_dict"].items():
        for ip in input_files:
            if not os.path.isfile(ip):
                continue
            audio, sr = wavio.read_wav(ensure_wav(ip))
            if audio.shape[0] == 1:
                audio = np.concatenate([audio, audio])
            mix = torch.from_numpy(audio[:2].copy()).to(model.ctx.device)
            if sr != 44100:                                      # librosa.load(wav_path, sr=44100, mono=False), :865
                mix = ensemble.resample(model.ctx, mix, sr, 44100)
                sr = 44100
            files_data.append({"base_name": os.path.splitext(os.path.basename(ip))[0], "mix": mix, "sr": sr,
                               "output_folder": out_folder})
    if not files_data:
        return []
    n = len(files_data)
    trans_opts = [model.reverb_removal, model.crowd_removal, model.noise_removal]
    count_v = sum(1 for o in trans_opts if o in {"All", "All Vocals", "Main Vocals"})
    count_i = sum(1 for o in trans_opts if o == "All")
    multi = not model.vocals_only
    model.total_steps = (min(max(1, model.ensemble_strength), len([m for m in model.ENSEMBLE if m[0] in model.separator.roster])) * n   # ensemble (:884-886)
                         + (n if model.separate_bg_vocals else 0) + (count_v + count_i) * n      # bg split, transforms
                         + (n if multi else 0) + (n if (model.alt_bass_model and multi) else 0)
                         + (n if (model.separate_drums and multi) else 0) + (n if (model.separate_woodwinds and multi) else 0)
                         + 1 + n)                                                                 # saving (:897)
    if model.callback is not None:
        _call_progress(model.callback, 0, "Starting ensemble separation...", model.total_steps)
    results = model._ensemble_separate_all(files_data)
    # stage order of :903-945: reverb removal on the vocals BEFORE the background split (the first call runs the WHOLE
    # chain, so with reverb removal on, the vocals meet the crowd / noise transforms twice -- reference behaviour)
    if model.reverb_removal != "Nothing":
        for base_name, res in results.items():
            if res.get("vocals") is not None:
                res["vocals"] = model._apply_transform_chain(res["vocals"], base_name, "vocals", output_folder=res["output_folder"], sr=res["sr"])
    if model.separate_bg_vocals:
        for base_name, res in results.items():
            if res.get("vocals") is not None:
                main_v, bg_v = model._apply_bg_vocal_splitting(res["vocals"], base_name)
                res["vocals"] = main_v
                if bg_v is not None:
                    res["bg_vocals"] = bg_v
    if any(o != "Nothing" for o in (model.crowd_removal, model.noise_removal)):
        for base_name, res in results.items():
            if res.get("vocals") is not None:
                res["vocals"] = model._apply_transform_chain(res["vocals"], base_name, "vocals", skip_transforms=["No Reverb"],
                                                             output_folder=res["output_folder"], sr=res["sr"])
            if res.get("instrumental") is not None:
                res["instrumental"] = model._apply_transform_chain(res["instrumental"], base_name, "instrumental",
                                                                   output_folder=res["output_folder"], sr=res["sr"])
    if not model.vocals_only:
        model._multistem_separation_all(results)
        if model.alt_bass_model:
            model._alt_bass_separation_all(results)
        if model.separate_drums:
            model._advanced_drum_separation_all(results)
        if model.separate_woodwinds:
            model._woodwinds_separation_all(results)
    return model._save_all_stems(results)


def separate_music(input_dict: Dict[str, List[str]], callback: Callable = None, **kwargs) -> List[str]:
    """:949-1001 -- same option names and defaults; extra keys of this build: ``precision`` ("fp16"|"bf16"|"fp32"), ``chunker``
    ("ola": audio-separator's runner, what :281 runs -- the default; "margin": the in-tree mdxnet.py runner), ``overlap`` (of the
    overlap-add runner), ``num_gpus`` (ranks of an initialised torch.distributed job that shard every model's chunks) and ``separator``
    (a pre-built engine, which then carries its own settings)."""
    options = {
        "input_dict": input_dict,
        "cpu": kwargs.get("cpu", False),
        "vocals_only": kwargs.get("vocals_only", True),
        "use_VOCFT": kwargs.get("use_VOCFT", False),
        "separate_drums": kwargs.get("separate_drums", False),
        "separate_woodwinds": kwargs.get("separate_woodwinds", False),
        "alt_bass_model": kwargs.get("alt_bass_model", False),
        "weight_InstVoc": kwargs.get("weight_InstVoc", 8.0),
        "weight_VOCFT": kwargs.get("weight_VOCFT", 1.0),
        "weight_VitLarge": kwargs.get("weight_VitLarge", 5.0),
        "reverb_removal": kwargs.get("reverb_removal", "Nothing"),
        "echo_removal": kwargs.get("echo_removal", "Nothing"),
        "delay_removal": kwargs.get("delay_removal", "Nothing"),
        "crowd_removal": kwargs.get("crowd_removal", "Nothing"),
        "noise_removal": kwargs.get("noise_removal", "Nothing"),
        "delay_removal_model": kwargs.get("delay_removal_model", "dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt"),
        "noise_removal_model": kwargs.get("noise_removal_model", "UVR-DeNoise.pth"),
        "crowd_removal_model": kwargs.get("crowd_removal_model", "UVR-MDX-NET_Crowd_HQ_1.onnx"),
        "separate_bg_vocals": kwargs.get("separate_bg_vocals", True),
        "bg_vocal_layers": kwargs.get("bg_vocal_layers", 1),
        "store_reverb_ir": kwargs.get("store_reverb_ir", False),
        "callback": callback,
        "ensemble_strength": kwargs.get("ensemble_strength", 2),
        "residual_blend": kwargs.get("residual_blend", 0.4),
        "precision": kwargs.get("precision", "fp16"),
        "chunker": kwargs.get("chunker", "ola"),
        "overlap": kwargs.get("overlap", 0.25),
        "num_gpus": kwargs.get("num_gpus", 1),
    }
    return predict_with_model(options, callback, separator=kwargs.get("separator"))
