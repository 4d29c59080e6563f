"""``Separate`` -- drop-in for the reference's Process->Separate plugin (wrappers/separate.py):
same class attributes (:22-30), the same 15 ``allowed_kwargs`` with the same defaults / choices
(:32-138), the same ``process_audio(inputs, callback=None, **kwargs)`` behaviour (:233-388: unknown
kwargs dropped, special TTS/ZONOS inputs copied to ``stems/<name>(Vocals).ext``, cache hit rule on
``stems/separation_info.json`` + SHA-256 of every stem, one batched ``separate_music`` call, results
mapped back by project folder, cache rewrite, pruning of extra stems under a lock)."""
import hashlib
import json
import logging
import os
import shutil
import threading
from typing import Any, Dict, List

from audiolab_amd.handlers import config
from audiolab_amd.separator.stem_separator import separate_music
from audiolab_amd.util.data_classes import ProjectFiles
from audiolab_amd.wrappers.base_wrapper import BaseWrapper, TypedInput

logger = logging.getLogger(__name__)

_SCOPE = ["Nothing", "Main Vocals", "All Vocals", "All"]


class Separate(BaseWrapper):
    title = "Separate"
    priority = 1
    default = True
    required = False
    description = ("Separate audio into distinct stems with optional background vocal splitting "
                   "and audio transformations (reverb, echo, delay, crowd, noise removal).")
    file_operation_lock = threading.Lock()

    allowed_kwargs = {
        "delete_extra_stems": TypedInput(default=True, type=bool, gradio_type="Checkbox",
                                         description="Automatically delete intermediate stem files after processing."),
        "separate_bg_vocals": TypedInput(default=False, type=bool, gradio_type="Checkbox",
                                         description="Separate background vocals from main vocals."),
        "bg_vocal_layers": TypedInput(default=1, le=10, ge=1, type=int, gradio_type="Slider", render=False,
                                      description="Number of background vocal layers to separate."),
        "vocals_only": TypedInput(default=True, type=bool, gradio_type="Checkbox",
                                  description="Enable to separate only the main vocals and instrumental, disable for additional stems."),
        "store_reverb_ir": TypedInput(default=False, type=bool, gradio_type="Checkbox",
                                      description="Store the impulse response for reverb removal. Will be used to re-apply reverb later."),
        "separate_drums": TypedInput(default=False, type=bool, gradio_type="Checkbox", description="Separate the drum track."),
        "separate_woodwinds": TypedInput(default=False, type=bool, gradio_type="Checkbox",
                                         description="Separate the woodwind instruments."),
        "alt_bass_model": TypedInput(default=False, type=bool, gradio_type="Checkbox", description="Use an alternative bass model."),
        "reverb_removal": TypedInput(default="Nothing", type=str, choices=list(_SCOPE), gradio_type="Dropdown",
                                     description="Apply reverb removal."),
        "echo_removal": TypedInput(default="Nothing", type=str, choices=list(_SCOPE), gradio_type="Dropdown",
                                   description="Apply echo/delay removal."),
        "crowd_removal": TypedInput(default="Nothing", type=str, choices=list(_SCOPE), gradio_type="Dropdown",
                                    description="Apply crowd noise removal."),
        "noise_removal": TypedInput(default="Nothing", type=str, choices=list(_SCOPE), gradio_type="Dropdown",
                                    description="Apply general noise removal."),
        "noise_removal_model": TypedInput(default="UVR-DeNoise.pth", type=str, gradio_type="Dropdown",
                                          choices=["UVR-DeNoise.pth", "UVR-DeNoise-Lite.pth"],
                                          description="Choose the model used for noise removal."),
        "delay_removal_model": TypedInput(default="dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt", type=str,
                                          gradio_type="Dropdown",
                                          choices=["dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt",
                                                   "dereverb-echo_mel_band_roformer_sdr_10.0169.ckpt", "UVR-DeEcho-DeReverb.pth"],
                                          description="Select the model for echo/delay removal."),
        "crowd_removal_model": TypedInput(default="UVR-MDX-NET_Crowd_HQ_1.onnx", type=str, gradio_type="Dropdown",
                                          choices=["UVR-MDX-NET_Crowd_HQ_1.onnx", "mel_band_roformer_crowd_aufr33_viperx_sdr_8.7144.ckpt"],
                                          description="Select the model for crowd noise removal."),
        # --- this build's engine knobs, after the reference's 15 keys and never rendered (base_wrapper.py:376-425 turns every entry of
        # this table into a validated settings field; render=False keeps it out of the UI as bg_vocal_layers above) ---
        "precision": TypedInput(default="fp16", type=str, choices=["fp16", "bf16", "fp32"], gradio_type="Dropdown", render=False,
                                description="Arithmetic of the separation networks: fp16 (the reference's autocast), bf16, or fp32."),
        "chunker": TypedInput(default="ola", type=str, choices=["ola", "margin"], gradio_type="Dropdown", render=False,
                              description="MDX-Net runner: 'ola' = audio-separator's overlap-add (what the reference runs), "
                                          "'margin' = the in-tree mdxnet.py margin chunker."),
        "overlap": TypedInput(default=0.25, ge=0.0, le=0.99, type=float, gradio_type="Slider", render=False,
                              description="Chunk overlap of the overlap-add runner."),
        "num_gpus": TypedInput(default=1, ge=1, le=8, type=int, gradio_type="Slider", render=False,
                               description="Ranks (one process per GPU, torch.distributed) that shard every model's chunks."),
    }
    ENGINE_KNOBS = {"precision": "fp16", "chunker": "ola", "overlap": 0.25, "num_gpus": 1}

    # further engine-level options (a pre-built engine under "separator", ensemble_strength, ...) reach separate_music when set here
    engine_options: Dict[str, Any] = {}

    def process_audio(self, inputs: List[ProjectFiles], callback=None, **kwargs: Dict[str, Any]) -> List[ProjectFiles]:
        filtered_kwargs = {k: v for k, v in kwargs.items() if k in self.allowed_kwargs}          # :234
        final_projects, to_separate = [], []
        for project in inputs:                                                                    # pass 1 (:239-315)
            project.base_name = os.path.splitext(os.path.basename(project.src_file))[0]
            out_dir = os.path.join(project.project_dir, "stems")
            os.makedirs(out_dir, exist_ok=True)
            cache_file = os.path.join(out_dir, "separation_info.json")
            file_basename, file_dir = os.path.basename(project.src_file), os.path.dirname(project.src_file)
            if (file_basename.startswith("TTS_") or file_basename.startswith("ZONOS_") or
                    any(d in file_dir for d in ("tts", "zonos", "stable_audio"))):                # :247-272
                base_name, ext = os.path.splitext(file_basename)
                new_path = os.path.join(out_dir, f"{base_name}(Vocals){ext}")
                if not os.path.exists(new_path):
                    shutil.copyfile(project.src_file, new_path)
                project.add_output("stems", [new_path])
                final_projects.append(project)
                logger.info(f"Skipping separation for special file {project.src_file}")
                continue
            g = filtered_kwargs.get
            current_config = {                                                                    # :274-291
                "file": project.src_file, "vocals_only": g("vocals_only", True), "separate_drums": g("separate_drums", False),
                "separate_woodwinds": g("separate_woodwinds", False), "alt_bass_model": g("alt_bass_model", False),
                "separate_bg_vocals": g("separate_bg_vocals", True), "bg_vocal_layers": g("bg_vocal_layers", 1),
                "reverb_removal": g("reverb_removal", "Nothing"), "echo_removal": g("echo_removal", "Nothing"),
                "delay_removal": g("delay_removal", "Nothing"), "crowd_removal": g("crowd_removal", "Nothing"),
                "noise_removal": g("noise_removal", "Nothing"),
                "delay_removal_model": g("delay_removal_model", "dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt"),
                "noise_removal_model": g("noise_removal_model", "UVR-DeNoise.pth"),
                "crowd_removal_model": g("crowd_removal_model", "UVR-MDX-NET_Crowd_HQ_1.onnx"),
                "store_reverb_ir": g("store_reverb_ir", True),
            }
            # an engine knob away from its default changes the stems: it joins the key then (at the defaults the key is the reference's)
            for knob, dflt in self.ENGINE_KNOBS.items():
                val = filtered_kwargs.get(knob, self.engine_options.get(knob, dflt))
                if knob != "num_gpus" and val != dflt:
                    current_config[knob] = val
            eng = self.engine_options.get("separator")
            if eng is not None and getattr(eng, "allow_synthetic", False):
                # stems made from random-init weights (bench / tests) must never satisfy a later run with real models;
                # with real weights the key is exactly the reference's (:274-291)
                current_config["weights"] = "synthetic-allowed"
            valid_cache = False
            if os.path.exists(cache_file):                                                        # :293-313
                try:
                    with open(cache_file, "r") as f:
                        cached = json.load(f)
                    if cached.get("config") == current_config:
                        stems, good = [], True
                        for info in cached.get("stems", []):
                            path, digest = info.get("path"), info.get("hash")
                            if not os.path.exists(path) or self._hash_file(path) != digest:
                                good = False
                                break
                            stems.append(path)
                        if good:
                            project.add_output("stems", stems)
                            final_projects.append(project)
                            valid_cache = True
                except Exception as e:
                    logger.warning(f"Error reading cache file {cache_file}: {e}")
            if not valid_cache:
                to_separate.append((project, current_config))

        if to_separate:                                                                           # pass 2 (:318-373)
            input_dict, project_map = {}, {}
            for proj, cfg in to_separate:
                stem_dir = os.path.join(proj.project_dir, "stems")
                os.makedirs(stem_dir, exist_ok=True)
                input_dict.setdefault(stem_dir, []).append(proj.src_file)
                project_map[os.path.basename(proj.project_dir)] = (proj, cfg)
            combined = separate_music(input_dict=input_dict, callback=callback, **{**self.engine_options, **filtered_kwargs})
            results: Dict[str, List[str]] = {}
            skip_parts = os.path.join(config.output_path, "process").split(os.path.sep)
            for stem in combined:                                                                 # :343-351
                parts = [p for p in os.path.dirname(stem).split(os.path.sep) if p not in skip_parts]
                results.setdefault(parts[0], []).append(stem)
            for base, (proj, cfg) in project_map.items():
                if base not in results:
                    logger.warning(f"No separation results found for project {proj.src_file}")
                    continue
                stems = results[base]
                proj.add_output("stems", stems)
                final_projects.append(proj)
                cache_info = {"config": cfg, "stems": [{"path": p, "hash": self._hash_file(p)} for p in stems]}
                try:
                    with open(os.path.join(proj.project_dir, "stems", "separation_info.json"), "w") as f:
                        json.dump(cache_info, f, indent=2)
                except Exception as e:
                    logger.warning(f"Error writing cache file: {e}")

        if filtered_kwargs.get("delete_extra_stems", True):                                       # :376-386
            for project in final_projects:
                out_dir = os.path.join(project.project_dir, "stems")
                final_stems = project.file_dict.get("stems", [])
                for fname in os.listdir(out_dir):
                    full = os.path.join(out_dir, fname)
                    if fname in ("separation_info.json", "impulse_response.ir"):
                        continue
                    if full not in final_stems:
                        self.del_stem(full)
        return final_projects

    def del_stem(self, path: str) -> bool:
        try:
            with self.file_operation_lock:
                if os.path.exists(path):
                    os.remove(path)
                    return True
        except Exception as e:
            print(f"Error deleting {path}: {e}")
        return False

    def _hash_file(self, filepath: str) -> str:
        h = hashlib.sha256()
        try:
            with open(filepath, "rb") as f:
                for chunk in iter(lambda: f.read(65536), b""):
                    h.update(chunk)
        except Exception as e:
            logger.warning(f"Error hashing file {filepath}: {e}")
        return h.hexdigest()
