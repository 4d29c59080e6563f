"""``Separator`` -- the engine object the Separate orchestrator drives, with exactly the members the
reference touches on ``audio_separator.separator.Separator`` (modules/separator/stem_separator.py:
102-107 ctor, :124 download_model_files, :394 load_model, :399-400 output_dir /
model_instance.output_dir, :281 separate(path) -> [basenames]) plus an in-memory
``separate_array`` used by the fast path (no temp WAV, no PCM16 round trip: SURVEY 8(b) b2).

Model roster.  The reference downloads its models at run time (stem_separator.py:109-124); none is reachable offline.  ``MODEL_ROSTER``
maps every file name the reference loads to its architecture and hyper-parameters: MDX-Net ``.onnx`` (TFC-TDF U-Net, csrc/tdfnet.hip),
Mel-Band / BS Roformer and MDX23C ``.ckpt`` (roformer.py / mdx23c.py over csrc/nn.hip), htdemucs ``.yaml`` (htdemucs.py), VR ``.pth``
(vrnet.py + vr_frontend.py).  Weights come from ``<model_file_dir>/<name>`` -- an ``.onnx`` graph (audiolab_amd.onnx_reader: network
hyper-parameters from the graph, n_fft and labels from the roster), a ``.ckpt`` / ``.pth`` / ``.pt`` state_dict, a demucs ``.th`` package
(restricted unpickler, audiolab_amd/th_reader.py), with a training-project ``.yaml`` beside a ``.ckpt`` overriding the roster's
hyper-parameters -- or, only with ``allow_synthetic=True`` (bench / tests), from a seeded random init.  Every loader checks the tensors'
shapes against the configuration and names the mismatching hyper-parameter.  Geometry per file name follows the public UVR / KUIELab /
training-project tables (PARITY UNPINNED: upstream, uncited).

Multi-stem entries.  A roster value ``("multi", [(label, cfg), ...])`` describes an MDX-Net style file set that yields several stems: one
network per label.

MDX runner.  ``chunker="margin"`` (default): margin chunker + trim stitching exactly as the in-tree
runner (mdxnet.py:109-197, pinned); the secondary stem is ``mix - primary`` in the time domain (mdxnet.py:211).
``chunker="ola"``: the third-party MDXSeparator's sequence (unpinned): normalisation to 0.9, Hann-window overlap-add with
``overlap`` and ``compensate``, and -- with ``invert_using_spec`` -- the secondary stem by spectral inversion against the
``is_match_mix`` rendition of the mix.
"""
from __future__ import annotations

import hashlib
import logging
import os
import types
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib, wavio
from ._lib import AlsepError, Context
from .mdx import Predictor
from .synth import synthetic_state_dict
from .htdemucs import DemucsRunner, HTDemucs, HTDemucsConfig
from .mdx23c import MDX23C, MDX23CConfig
from .roformer import Roformer, RoformerConfig, RoformerRunner
from .tdfnet import TDFNet, TDFNetConfig

logger = logging.getLogger(__name__)

# name -> (primary stem label, secondary stem label or None, TDFNetConfig[, {"compensate": c}])
#
# PROVENANCE of the geometry below: the model files are downloaded at run time by the reference
# (stem_separator.py:109-124) and are not in /root/reference; n_fft / dim_f / dim_t / compensate per file are the
# values of the public UVR ``model_data.json`` / KUIELab mdx-net-submission tables as recalled by the builder
# (upstream, uncited -- PARITY UNPINNED).  What a real ``.onnx`` file itself fixes (dim_f, dim_t, L, g, l, bn) is
# read from its graph and OVERRIDES this table (onnx_reader.py); only n_fft / hop / compensate / the stem labels
# cannot be recovered from the file and come from here (or from the caller: ``load_model(..., n_fft=)``,
# ``patched_load_model`` passes MDXSeparator's own ``n_fft`` / ``hop_length``).
def _cfg(n_fft: int, dim_f: int, dim_t: int) -> TDFNetConfig:
    return TDFNetConfig(dim_f=dim_f, dim_t=dim_t, n_fft=n_fft, g=48)


MODEL_ROSTER: Dict[str, tuple] = {
    # UVR MDX-Net vocal models (ensemble slots 5-7, stem_separator.py:384-386): n_fft 7680, dim_f 3072, dim_t 2**8
    "UVR-MDX-NET-Voc_FT.onnx": ("Vocals", "Instrumental", _cfg(7680, 3072, 256), {"compensate": 1.021}),
    "Kim_Vocal_2.onnx": ("Vocals", "Instrumental", _cfg(7680, 3072, 256), {"compensate": 1.009}),
    "Kim_Vocal_1.onnx": ("Vocals", "Instrumental", _cfg(7680, 3072, 256), {"compensate": 1.043}),
    # crowd removal (wrappers/separate.py:131-137, stem_separator.py:798): n_fft 5120, dim_f 2560
    "UVR-MDX-NET_Crowd_HQ_1.onnx": ("No Crowd", "Crowd", _cfg(5120, 2560, 256), {"compensate": 1.035}),
    # KUIELab MDX-Net "a" set (stem_separator.py:512 uses the bass one): dim_f 2048 everywhere, n_fft per target
    "kuielab_a_vocals.onnx": ("Vocals", "Instrumental", _cfg(6144, 2048, 256), {"compensate": 1.035}),
    "kuielab_a_drums.onnx": ("Drums", "No Drums", _cfg(4096, 2048, 128), {"compensate": 1.035}),
    "kuielab_a_bass.onnx": ("Bass", "No Bass", _cfg(16384, 2048, 512), {"compensate": 1.035}),
    "kuielab_a_other.onnx": ("Other", "No Other", _cfg(8192, 2048, 512), {"compensate": 1.035}),
    # Roformer members of the ensemble (stem_separator.py:380-382) and the de-reverb / de-echo transforms (:796-797, wrappers/separate.py:
    # 123-130).  Hyper-parameters as published with the checkpoints (upstream, uncited); a ``<name>.yaml`` beside the weights overrides them.
    "vocals_mel_band_roformer.ckpt": ("roformer", RoformerConfig(kind="mel", dim=384, depth=6), {"labels": ("Vocals",), "secondary": "Instrumental"}),
    "model_bs_roformer_ep_368_sdr_12.9628.ckpt": ("roformer", RoformerConfig(kind="bs", dim=384, depth=12),
                                                  {"labels": ("Vocals",), "secondary": "Instrumental"}),
    "melband_roformer_big_beta4.ckpt": ("roformer", RoformerConfig(kind="mel", dim=384, depth=12), {"labels": ("Vocals",), "secondary": "Instrumental"}),
    "dereverb_mel_band_roformer_anvuew_sdr_19.1729.ckpt": ("roformer", RoformerConfig(kind="mel", dim=384, depth=6),
                                                           {"labels": ("No Reverb",), "secondary": "Reverb"}),
    "dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt": ("roformer", RoformerConfig(kind="mel", dim=384, depth=6),
                                                            {"labels": ("dry",), "secondary": "No dry"}),
    "dereverb-echo_mel_band_roformer_sdr_10.0169.ckpt": ("roformer", RoformerConfig(kind="mel", dim=384, depth=6),
                                                         {"labels": ("dry",), "secondary": "No dry"}),
    "mel_band_roformer_crowd_aufr33_viperx_sdr_8.7144.ckpt": ("roformer", RoformerConfig(kind="mel", dim=384, depth=6),
                                                              {"labels": ("No Crowd",), "secondary": "Crowd"}),
    # MDX23C (TFC-TDF v3): ensemble slot 4 (:383) and the drum-kit splitter (:541; its hyper-parameters are not known offline -- the
    # InstVoc shape with six instruments stands in until a yaml sits beside the weights)
    "MDX23C-8KFFT-InstVoc_HQ.ckpt": ("mdx23c", MDX23CConfig(instruments=("vocals", "other")), {"labels": ("Vocals", "Instrumental")}),
    "MDX23C-DrumSep-aufr33-jarredou.ckpt": ("mdx23c", MDX23CConfig(instruments=("kick", "snare", "toms", "hh", "ride", "crash")),
                                            {"labels": ("Kick", "Snare", "Toms", "HH", "Ride", "Crash")}),
    # VR-architecture models: woodwinds split (stem_separator.py:596), noise removal (:148, wrappers/separate.py:114-117), the standalone
    # de-echo / de-reverb list (:1048-1050).  (architecture, parameter set, nout, nout_lstm) and which stem the network predicts are the
    # values published with the models (upstream, uncited); labels as the orchestrator matches them ("(woodwinds)" :615, "No Noise" :800).
    # UVR-BVE-4B_SN-44100-1.pth (:752, the BG-vocal split): its parameter set converts the top band's channels ("stereo_n", as modelparams/
    # 4band_v2_sn.json:48 does for the v2 bands); the set itself (4band_v3_sn) and the conversion rule are upstream's, not in the reference
    # tree (vr_frontend.MODEL_PARAMS note) -- this entry is unpinned.  Labels as :763-766 matches them: "(Vocals)" = background.
    "17_HP-Wind_Inst-UVR.pth": ("vr", dict(arch="nets_123821KB", params="4band_v2"), {"labels": ("No Woodwinds", "Woodwinds")}),
    "UVR-BVE-4B_SN-44100-1.pth": ("vr", dict(arch="new", params="4band_v3_sn", nout=64, nout_lstm=128), {"labels": ("Vocals", "Instrumental")}),
    "UVR-DeNoise.pth": ("vr", dict(arch="new", params="4band_v3", nout=48, nout_lstm=128), {"labels": ("Noise", "No Noise")}),
    "UVR-DeNoise-Lite.pth": ("vr", dict(arch="new", params="4band_v3", nout=16, nout_lstm=128), {"labels": ("Noise", "No Noise")}),
    "UVR-DeEcho-DeReverb.pth": ("vr", dict(arch="new", params="4band_v3", nout=64, nout_lstm=128), {"labels": ("No Reverb", "Reverb")}),
    "UVR-De-Echo-Normal.pth": ("vr", dict(arch="new", params="4band_v3", nout=48, nout_lstm=128), {"labels": ("No Echo", "Echo")}),
    "UVR-De-Echo-Aggressive.pth": ("vr", dict(arch="new", params="4band_v3", nout=48, nout_lstm=128), {"labels": ("No Echo", "Echo")}),
    # the multi-stem stage (stem_separator.py:466): HTDemucs 6 sources on the full mix; DemucsSeparator defaults shifts 2, overlap 0.25
    "htdemucs_6s.yaml": ("demucs", HTDemucsConfig(), {"shifts": 2, "overlap": 0.25}),
}
# BASELINE configs[1] "MDX-Net UVR 4-stem": four single-target networks of the in-tree geometry (mdxnet.py:247-251:
# dim_f 3072, n_fft 6144) -- the bench workload; never reached by a file name of the reference
_BENCH = _cfg(6144, 3072, 256)
BENCH_ROSTER: Dict[str, tuple] = {f"bench_4stem_{t.lower()}.onnx": (t, f"No {t}", _BENCH) for t in ("Vocals", "Drums", "Bass", "Other")}
FOUR_STEM_SET = ("kuielab_a_vocals.onnx", "kuielab_a_drums.onnx", "kuielab_a_bass.onnx", "kuielab_a_other.onnx")


# The .yaml that audio-separator downloads beside a .ckpt does not always carry the checkpoint's name.  Names as published in that
# package's model list, as recalled by the builder (upstream, uncited -- the package is not in /root/reference): tried after
# ``<checkpoint stem>.yaml``; a miss falls back to the roster's hyper-parameters with a WARNING.
YAML_NAMES: Dict[str, tuple] = {
    "MDX23C-8KFFT-InstVoc_HQ.ckpt": ("model_2_stem_full_band_8k.yaml",),
    "MDX23C-DrumSep-aufr33-jarredou.ckpt": ("aufr33-jarredou_DrumSep_model_mdx23c_ep_141_sdr_10.8059.yaml",),
    "melband_roformer_big_beta4.ckpt": ("config_melbandroformer_big_beta4.yaml",),
    "dereverb_mel_band_roformer_anvuew_sdr_19.1729.ckpt": ("dereverb_mel_band_roformer_anvuew.yaml",),
    "dereverb-echo_mel_band_roformer_sdr_13.4843_v2.ckpt": ("config_dereverb-echo_mel_band_roformer.yaml",),
    "dereverb-echo_mel_band_roformer_sdr_10.0169.ckpt": ("config_dereverb-echo_mel_band_roformer.yaml",),
    "mel_band_roformer_crowd_aufr33_viperx_sdr_8.7144.ckpt": ("model_mel_band_roformer_crowd.yaml",),
}


def _yaml_loader():
    """yaml.SafeLoader that also reads ``!!python/tuple`` (the training project's BS-RoFormer configs write ``freqs_per_bands`` with
    that tag; plain safe_load raises ConstructorError on it).  Nothing else beyond the safe schema is constructed."""
    import yaml

    class Loader(yaml.SafeLoader):
        pass
    Loader.add_constructor("tag:yaml.org,2002:python/tuple", lambda ld, node: tuple(ld.construct_sequence(node, deep=True)))
    return Loader


def load_model_yaml(model_file_dir: str, model_filename: str) -> Optional[dict]:
    """The hyper-parameter file of a .ckpt model: ``<stem>.yaml``, then the names of ``YAML_NAMES``; None (and a WARNING) if absent."""
    import yaml
    cands = (os.path.splitext(model_filename)[0] + ".yaml",) + tuple(YAML_NAMES.get(model_filename, ()))
    for name in cands:
        path = os.path.join(model_file_dir, name)
        if os.path.isfile(path):
            with open(path) as f:
                return yaml.load(f, Loader=_yaml_loader()) or {}
    logger.warning("%s: no hyper-parameter file (%s) under %s -- using the roster's values; a checkpoint trained with other values is "
                   "rejected by the shape check", model_filename, " / ".join(cands), model_file_dir)
    return None


class _ModelInstance:
    """What ``separator.model_instance`` exposes to the orchestrator (only ``output_dir`` is touched)."""

    def __init__(self, name: str, net: TDFNet, predictor: Predictor, primary: str, secondary: Optional[str]):
        self.model_name = name
        self.net = net
        self.predictor = predictor
        self.primary_stem_name = primary
        self.secondary_stem_name = secondary
        self.output_dir = None
        self.model_run = net                                  # callable(spek) -> pred, the patch_separate seam
        self.extra: List[tuple] = []                          # multi-stem models: further (label, net, predictor)
        self.demucs: Optional[DemucsRunner] = None            # Demucs-family model: one network, all sources at once
        self.roformer: Optional[RoformerRunner] = None        # Roformer-family model: its own chunked runner
        self.vr = None                                        # VR-architecture model: vr_frontend.VRSeparator


class Separator:
    def __init__(self, log_level=logging.INFO, model_file_dir: str = "models/audio_separator", output_dir: Optional[str] = None,
                 invert_using_spec: bool = True, use_autocast: bool = True, ctx: Optional[Context] = None,
                 dtype: Optional[torch.dtype] = None, sample_rate: int = 44100, chunks: int = 0, margin: int = 44100,
                 denoise: bool = False, max_batch: int = 32, sharded: bool = False, roster: Optional[Dict[str, tuple]] = None,
                 chunker: str = "margin", overlap: float = 0.25, compensate: Optional[float] = None,
                 allow_synthetic: bool = False, normalization_threshold: float = 0.9, f32_contraction: str = "split", nn_contraction: str = "exact", **_ignored):
        """``allow_synthetic=True`` (bench, tests): a roster name without a weight file gets seeded random-init weights.
        The default refuses to: a missing model file is an error, never plausible-looking noise."""
        self.log_level = log_level
        self.model_file_dir = model_file_dir
        self.output_dir = output_dir
        # honoured by the audio-separator style runner (chunker="ola"): the secondary stem by spectral inversion; the in-tree runner
        # (chunker="margin", mdxnet.py:211) has no such option and yields mix - primary
        self.invert_using_spec = bool(invert_using_spec)
        self.normalization = float(normalization_threshold or 0.0)
        self.use_autocast = use_autocast
        self.ctx = ctx if ctx is not None else _lib.default_context(None)
        # use_autocast=True is the reference's GPU setting (stem_separator.py:106): torch autocast on CUDA = IEEE half.  Measured at
        # the bench geometry against the fp32 oracle (tests/test_gpu_parity.py): f16 3.5 % relative L2 (SDR 29 dB), bf16 20 % (14 dB),
        # at the same speed -- so half precision here means float16; bfloat16 stays available through ``dtype=``.
        self.dtype = dtype if dtype is not None else (torch.float16 if use_autocast else torch.float32)
        self.sample_rate = sample_rate
        self.chunks, self.margin, self.denoise = chunks, margin, denoise
        self.max_batch = max_batch
        self.sharded = sharded
        self.roster = dict(MODEL_ROSTER if roster is None else roster)
        if chunker not in ("margin", "ola"):
            raise AlsepError("chunker must be 'margin' (in-tree runner, pinned) or 'ola' (audio-separator style, unpinned)")
        self.chunker, self.overlap, self.compensate = chunker, overlap, compensate
        self.allow_synthetic = bool(allow_synthetic)
        # float32 TFC-TDF networks: "split" = float32 storage with the contractions as (hi, lo) half products on the 16-bit matrix pipe
        # (float32 accuracy at ~3 x the speed; activations limited to the half range, checked per batch: an out-of-range batch is redone on the exact kernels), "exact" = f32 MFMA fmaf chains
        if f32_contraction not in ("split", "exact"):
            raise AlsepError("f32_contraction must be 'split' or 'exact'")
        self.f32_contraction = f32_contraction
        # the float32 modes of the OTHER families (HTDemucs, Roformer, MDX23C): "exact" (default: f32 MFMA, four runner lanes) or "split" (the
        # generic kernels of csrc/nn_f32s.h on ONE lane: faster per kernel, but 16-bit MFMA kernels cannot share the GPU with other lanes' FFTs)
        if nn_contraction not in ("split", "exact"):
            raise AlsepError("nn_contraction must be 'split' or 'exact'")
        self.nn_contraction = nn_contraction
        self.model_instance: Optional[_ModelInstance] = None
        self._cache: Dict[str, _ModelInstance] = {}

    # -- roster ---------------------------------------------------------------------------------
    def download_model_files(self, model_filename: str) -> None:
        """No network here: just validate that the name is known (or present on disk)."""
        if model_filename not in self.roster and not self._onnx_path(model_filename):
            logger.debug("model %s is not an MDX-Net model of this build's roster; load_model would fail", model_filename)

    def _onnx_path(self, model_filename: str) -> Optional[str]:
        p = os.path.join(self.model_file_dir, model_filename)
        return p if model_filename.lower().endswith(".onnx") and os.path.isfile(p) else None

    def weights_provenance(self) -> str:
        """"real" / "synthetic" for the loaded model (goes into the Separate cache key)."""
        return getattr(self.model_instance, "weights", "none") if self.model_instance else "none"

    def load_model(self, model_filename: str, n_fft: Optional[int] = None, hop: Optional[int] = None) -> None:
        """Weights stay resident per model name: the reference reloads per ensemble member
        (stem_separator.py:394); here a second load_model of the same name is a dictionary hit.
        ``n_fft`` / ``hop`` override the roster's values (they are not in an .onnx file)."""
        if model_filename in self._cache:
            self.model_instance = self._cache[model_filename]
            self.model_instance.output_dir = self.output_dir
            return
        onnx_path = self._onnx_path(model_filename)
        if model_filename not in self.roster and onnx_path:    # a real MDX-Net file outside the roster: vocal model defaults
            self.roster[model_filename] = ("Vocals", "Instrumental", None)
        if model_filename not in self.roster:
            raise AlsepError(f"model '{model_filename}' is not available in this build (MDX-Net roster: {sorted(self.roster)})")
        entry = self.roster[model_filename]
        if entry[0] == "demucs":
            self._load_demucs(model_filename, entry)
            return
        if entry[0] == "roformer":
            self._load_roformer(model_filename, entry)
            return
        if entry[0] == "mdx23c":
            self._load_mdx23c(model_filename, entry)
            return
        if entry[0] == "vr":
            self._load_vr(model_filename, entry)
            return
        meta = entry[3] if len(entry) > 3 and entry[0] != "multi" else {}
        entry = entry[:3]
        provenance = []
        if entry[0] == "multi":                                 # ("multi", [(label, cfg), ...]): one network per stem
            stems = [(label, None, cfg) for label, cfg in entry[1]]
        else:
            stems = [entry]

        def build(tag: str, cfg: Optional[TDFNetConfig]):
            pt = os.path.join(self.model_file_dir, tag + ".pt")
            if onnx_path and tag == model_filename:
                from .onnx_reader import load_mdx_onnx
                m = load_mdx_onnx(onnx_path, n_fft=n_fft or (cfg.n_fft if cfg else None), hop=hop or (cfg.hop if cfg else 1024))
                if cfg is not None and (m.config.dim_f, m.config.dim_t) != (cfg.dim_f, cfg.dim_t):
                    logger.warning("%s: the file holds dim_f=%d dim_t=%d, the roster says %d / %d; using the file's", model_filename,
                                   m.config.dim_f, m.config.dim_t, cfg.dim_f, cfg.dim_t)
                cfg, sd = m.config, m.state_dict
                provenance.append("real")
            elif cfg is None:
                raise AlsepError(f"model '{model_filename}': no geometry in the roster and no model file")
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu", weights_only=True)
                provenance.append("real")
            elif self.allow_synthetic:
                seed = int.from_bytes(hashlib.sha256(tag.encode()).digest()[:4], "little")
                sd = synthetic_state_dict(cfg, seed=seed)
                logger.warning("%s: no weight file under %s -- SYNTHETIC random-init weights (allow_synthetic=True)", tag,
                               self.model_file_dir)
                provenance.append("synthetic")
            else:
                raise AlsepError(f"model '{model_filename}': no weight file ({os.path.join(self.model_file_dir, model_filename)} "
                                 f"or {pt}); random-init weights are only used with Separator(allow_synthetic=True)")
            net = TDFNet(cfg, sd, ctx=self.ctx, dtype=self.dtype, max_batch=self.max_batch,
                         contraction=self.f32_contraction if self.dtype == torch.float32 else None)
            dim_t_arg = int(cfg.dim_t).bit_length() - 1
            args = types.SimpleNamespace(margin=self.margin, chunks=self.chunks, denoise=self.denoise, dim_f=cfg.dim_f,
                                         dim_t=dim_t_arg, n_fft=cfg.n_fft)
            if self.chunker == "ola":
                from .mdx import OlaRunner
                comp = self.compensate if self.compensate is not None else float(meta.get("compensate", 1.0))
                pred = OlaRunner(net, ctx=self.ctx, overlap=self.overlap, compensate=comp, denoise=self.denoise,
                                 max_batch=self.max_batch, sharded=self.sharded)
            else:
                pred = Predictor(args, net, ctx=self.ctx, hop=cfg.hop, sharded=self.sharded)
            return net, pred

        primary, secondary, cfg = stems[0]
        net, pred = build(model_filename if len(stems) == 1 else f"{model_filename}#{primary}", cfg)
        inst = _ModelInstance(model_filename, net, pred, primary, secondary)
        for label, _, cfg_i in stems[1:]:
            net_i, pred_i = build(f"{model_filename}#{label}", cfg_i)
            inst.extra.append((label, net_i, pred_i))
        inst.output_dir = self.output_dir
        inst.weights = "synthetic" if "synthetic" in provenance else "real"
        self._cache[model_filename] = inst
        self.model_instance = inst

    def _load_demucs(self, model_filename: str, entry: tuple) -> None:
        """("demucs", HTDemucsConfig, {shifts, overlap}).  Weights, in this order: the files the reference has -- ``<dir>/<name>.yaml``
        (demucs' bag-of-models list) pointing at ``<signature>-<checksum>.th`` (a pickled package: read by audiolab_amd.th_reader's
        allow-list unpickler, hyper-parameters from its ``kwargs``, ``state`` as the weights); ``<dir>/<name>.pt`` (a plain state_dict with
        demucs' parameter names); with allow_synthetic, seeded random-init ones.  float32 (the kernels of this family are fp32)."""
        from . import th_reader
        cfg = entry[1]
        opts = entry[2] if len(entry) > 2 else {}
        pt = os.path.join(self.model_file_dir, model_filename + ".pt")
        th_path = th_reader.resolve_demucs_yaml(self.model_file_dir, model_filename) if model_filename.endswith(".yaml") else None
        if th_path is None and model_filename.endswith(".th") and os.path.isfile(os.path.join(self.model_file_dir, model_filename)):
            th_path = os.path.join(self.model_file_dir, model_filename)
        sd = None
        if th_path is not None:
            pkg = th_reader.read_th(th_path)
            if pkg["klass"] != "HTDemucs":
                raise AlsepError(f"{th_path}: a {pkg['klass']} package -- only HTDemucs is implemented")
            cfg = th_reader.htdemucs_config_from_kwargs(pkg["kwargs"])
            sd, weights = pkg["state"], "real"
        else:
            sd = self._weights_file(model_filename) if not model_filename.endswith(".yaml") else None
            if sd is None and os.path.isfile(pt):
                sd = torch.load(pt, map_location="cpu", weights_only=True)
                sd = sd.get("state", sd) if isinstance(sd, dict) and isinstance(sd.get("state"), dict) else sd
        if sd is not None:
            weights = "real"
        elif self.allow_synthetic:
            from .htdemucs import synthetic_state_dict as demucs_synth
            seed = int.from_bytes(hashlib.sha256(model_filename.encode()).digest()[:4], "little")
            sd, weights = demucs_synth(cfg, seed=seed), "synthetic"
            logger.warning("%s: no weight file under %s -- SYNTHETIC random-init weights (allow_synthetic=True)", model_filename,
                           self.model_file_dir)
        else:
            raise AlsepError(f"model '{model_filename}': no weight file ({os.path.join(self.model_file_dir, model_filename)} naming a .th "
                             f"package, or {pt}); random-init weights are only used with Separator(allow_synthetic=True)")
        net = HTDemucs(cfg, sd, ctx=self.ctx)
        inst = _ModelInstance(model_filename, net, None, cfg.sources[0].capitalize(), None)
        inst.demucs = DemucsRunner(net, shifts=int(opts.get("shifts", 2)), overlap=float(opts.get("overlap", 0.25)), sharded=self.sharded,
                                   contraction=self.nn_contraction)
        inst.output_dir = self.output_dir
        inst.weights = weights
        self._cache[model_filename] = inst
        self.model_instance = inst

    def _weights_file(self, model_filename: str):
        """state_dict of ``<dir>/<name>`` (a .ckpt / .pth / .pt written with torch.save; plain tensors only: weights_only) or ``<name>.pt``"""
        for cand in (os.path.join(self.model_file_dir, model_filename), os.path.join(self.model_file_dir, model_filename + ".pt")):
            if os.path.isfile(cand) and not cand.lower().endswith((".onnx", ".yaml")):
                sd = torch.load(cand, map_location="cpu", weights_only=True)
                if isinstance(sd, dict) and isinstance(sd.get("state_dict"), dict):
                    sd = sd["state_dict"]
                if isinstance(sd, dict) and isinstance(sd.get("state"), dict):
                    sd = sd["state"]
                return sd
        return None

    def _load_roformer(self, model_filename: str, entry: tuple) -> None:
        """("roformer", RoformerConfig, {labels, secondary}).  A ``<name>.yaml`` in the training project's layout beside the weights
        (model: dim / depth / heads / dim_head / num_bands | freqs_per_bands / stft_n_fft / stft_hop_length / num_stems /
        mask_estimator_depth / mlp_expansion_factor; audio: chunk_size; inference: num_overlap) overrides the roster's hyper-parameters."""
        import dataclasses
        cfg, opts = entry[1], (entry[2] if len(entry) > 2 else {})
        y = load_model_yaml(self.model_file_dir, model_filename)
        if y is not None:
            m, over = y.get("model", {}) or {}, {}
            for src, dst in (("dim", "dim"), ("depth", "depth"), ("heads", "heads"), ("dim_head", "dim_head"), ("num_bands", "num_bands"),
                             ("num_stems", "num_stems"), ("stft_n_fft", "n_fft"), ("stft_hop_length", "hop"),
                             ("mask_estimator_depth", "mask_estimator_depth"), ("mlp_expansion_factor", "mlp_expansion_factor"),
                             ("sample_rate", "sample_rate")):
                if src in m:
                    over[dst] = int(m[src])
            if "freqs_per_bands" in m:
                over["freqs_per_bands"], over["kind"] = tuple(int(v) for v in m["freqs_per_bands"]), "bs"
            elif "num_bands" in m:
                over["kind"] = "mel"
            if "chunk_size" in (y.get("audio") or {}):
                over["chunk_size"] = int(y["audio"]["chunk_size"])
            if "num_overlap" in (y.get("inference") or {}):
                over["num_overlap"] = int(y["inference"]["num_overlap"])
            cfg = dataclasses.replace(cfg, **over)
        sd, weights = self._weights_file(model_filename), "real"
        if sd is None:
            if not self.allow_synthetic:
                raise AlsepError(f"model '{model_filename}': no weight file under {self.model_file_dir}; random-init weights are only used "
                                 f"with Separator(allow_synthetic=True)")
            from .roformer import synthetic_state_dict as roformer_synth
            seed = int.from_bytes(hashlib.sha256(model_filename.encode()).digest()[:4], "little")
            sd, weights = roformer_synth(cfg, seed=seed), "synthetic"
            logger.warning("%s: no weight file under %s -- SYNTHETIC random-init weights (allow_synthetic=True)", model_filename, self.model_file_dir)
        # use_autocast=True (the reference's GPU setting, stem_separator.py:106): the network's half-precision mode; float32 otherwise
        # (a checkpoint whose head dimension is not the 64 the one-pass attention kernel is written for runs in float32 with a WARNING)
        half_ok = self.dtype != torch.float32 and cfg.dim_head == 64
        if self.dtype != torch.float32 and not half_ok:
            logger.warning("%s: dim_head %d -- running this Roformer in float32 (its half-precision attention kernel handles 64)", model_filename, cfg.dim_head)
        net = Roformer(cfg, sd, ctx=self.ctx, precision="f16" if half_ok else "f32")
        labels = tuple(opts.get("labels", ("Vocals",)))[: cfg.num_stems]
        inst = _ModelInstance(model_filename, net, None, labels[0], opts.get("secondary") if cfg.num_stems == 1 else None)
        inst.roformer = RoformerRunner(net, labels, sharded=self.sharded, contraction=self.nn_contraction)
        inst.output_dir = self.output_dir
        inst.weights = weights
        self._cache[model_filename] = inst
        self.model_instance = inst

    def _load_mdx23c(self, model_filename: str, entry: tuple) -> None:
        """("mdx23c", MDX23CConfig, {labels}); a yaml of the training project beside the weights (audio: n_fft / hop_length / dim_f /
        chunk_size; model: num_subbands / num_scales / num_blocks_per_scale / num_channels / growth / bottleneck_factor; training:
        instruments; inference: num_overlap) overrides the roster's hyper-parameters."""
        import dataclasses
        cfg, opts = entry[1], (entry[2] if len(entry) > 2 else {})
        labels = tuple(opts.get("labels", tuple(n.capitalize() for n in cfg.instruments)))
        y = load_model_yaml(self.model_file_dir, model_filename)
        if y is not None:
            a, m, over = y.get("audio", {}) or {}, y.get("model", {}) or {}, {}
            for src, dst in (("n_fft", "n_fft"), ("hop_length", "hop"), ("dim_f", "dim_f"), ("chunk_size", "chunk_size")):
                if src in a:
                    over[dst] = int(a[src])
            for key in ("num_subbands", "num_scales", "num_blocks_per_scale", "num_channels", "growth", "bottleneck_factor"):
                if key in m:
                    over[key] = int(m[key])
            if "instruments" in (y.get("training") or {}):
                over["instruments"] = tuple(str(v) for v in y["training"]["instruments"])
                if "labels" not in opts or len(opts["labels"]) != len(over["instruments"]):
                    labels = tuple(n.capitalize() for n in over["instruments"])
            if "num_overlap" in (y.get("inference") or {}):
                over["num_overlap"] = int(y["inference"]["num_overlap"])
            cfg = dataclasses.replace(cfg, **over)
        sd, weights = self._weights_file(model_filename), "real"
        if sd is None:
            if not self.allow_synthetic:
                raise AlsepError(f"model '{model_filename}': no weight file under {self.model_file_dir}; random-init weights are only used "
                                 f"with Separator(allow_synthetic=True)")
            from .mdx23c import synthetic_state_dict as mdx23c_synth
            seed = int.from_bytes(hashlib.sha256(model_filename.encode()).digest()[:4], "little")
            sd, weights = mdx23c_synth(cfg, seed=seed), "synthetic"
            logger.warning("%s: no weight file under %s -- SYNTHETIC random-init weights (allow_synthetic=True)", model_filename, self.model_file_dir)
        # use_autocast: the half-precision mode -- where the checkpoint's channel counts allow it (its convolution wants input channels in
        # multiples of 64: true of the published models); a model that does not fit runs in float32 with a WARNING instead of failing the stage
        half_ok = self.dtype != torch.float32 and cfg.num_channels % 64 == 0 and cfg.growth % 64 == 0
        if self.dtype != torch.float32 and not half_ok:
            logger.warning("%s: num_channels %d / growth %d are not multiples of 64 -- running this MDX23C model in float32", model_filename,
                           cfg.num_channels, cfg.growth)
        net = MDX23C(cfg, sd, ctx=self.ctx, precision="f16" if half_ok else "f32")
        inst = _ModelInstance(model_filename, net, None, labels[0], None)
        inst.roformer = RoformerRunner(net, labels, sharded=self.sharded, contraction=self.nn_contraction)   # the same chunked runner (demix_track)
        inst.output_dir = self.output_dir
        inst.weights = weights
        self._cache[model_filename] = inst
        self.model_instance = inst

    def _load_vr(self, model_filename: str, entry: tuple) -> None:
        """("vr", {arch, params, nout, nout_lstm}, {labels: (predicted stem, residual stem), aggression, window_size, tta,
        high_end_process}): a VR network behind the multi-band front / back end (vr_frontend.VRSeparator).  The runner options default
        to the values the reference's engine uses when AudioLab passes none (window 512, aggression 5, no TTA, no mirrored high end --
        audio-separator's ``vr_params`` defaults; upstream, uncited); the in-tree runner (vr.py:20-37: agg 10, mirroring) is
        ``VRSeparator(net, params, agg=10, high_end_process=True)``.  float32 (the kernels of this family are fp32)."""
        from .vr_frontend import MODEL_PARAMS, VRSeparator
        from . import vrnet
        spec, opts = entry[1], (entry[2] if len(entry) > 2 else {})
        n_fft = 2 * MODEL_PARAMS[spec["params"]]["bins"]
        sd, weights = self._weights_file(model_filename), "real"
        if sd is None:
            if not self.allow_synthetic:
                raise AlsepError(f"model '{model_filename}': no weight file under {self.model_file_dir}; random-init weights are only used "
                                 f"with Separator(allow_synthetic=True)")
            seed = int.from_bytes(hashlib.sha256(model_filename.encode()).digest()[:4], "little")
            if spec["arch"] == "new":
                sd = vrnet.random_state_dict_new(n_fft, spec["nout"], spec["nout_lstm"], seed=seed)
            else:
                sd = vrnet.random_state_dict(vrnet.WIDTHS[spec["arch"]], seed=seed)
            weights = "synthetic"
            logger.warning("%s: no weight file under %s -- SYNTHETIC random-init weights (allow_synthetic=True)", model_filename, self.model_file_dir)
        if spec["arch"] == "new":
            net = vrnet.VRNetNew(n_fft, sd, nout=spec["nout"], nout_lstm=spec["nout_lstm"], ctx=self.ctx)
        else:
            net = vrnet.VRNet(n_fft, sd, variant=spec["arch"], ctx=self.ctx)
        labels = tuple(opts.get("labels", ("Instrumental", "Vocals")))
        inst = _ModelInstance(model_filename, net, None, labels[0], labels[1])
        inst.vr = VRSeparator(net, spec["params"], agg=int(opts.get("aggression", 5)), window_size=int(opts.get("window_size", 512)),
                              tta=bool(opts.get("tta", False)), max_batch=min(self.max_batch, 4),
                              high_end_process=bool(opts.get("high_end_process", False)))
        inst.output_dir = self.output_dir
        inst.weights = weights
        self._cache[model_filename] = inst
        self.model_instance = inst

    # -- inference --------------------------------------------------------------------------------
    def separate_array(self, mix) -> Dict[str, torch.Tensor]:
        """mix [C,N] (numpy or tensor) -> {stem label: [C,N] device tensor}.  Mono is duplicated to stereo (and comes
        back as [2,N], as the reference's loaders do); more than two channels are separated as consecutive stereo pairs
        (BASELINE configs[4]: 8-channel long-form), an odd last channel as a duplicated pair."""
        if self.model_instance is None:
            raise AlsepError("load_model() first")
        m = torch.as_tensor(mix, dtype=torch.float32)
        if m.dim() == 1:
            m = torch.stack([m, m])
        if m.dim() != 2:
            raise AlsepError("separate_array expects [channels, samples]")
        if m.shape[0] == 1:
            m = torch.cat([m, m])
        if m.shape[0] > 2:
            c = m.shape[0]
            parts: Dict[str, List[torch.Tensor]] = {}
            for c0 in range(0, c, 2):
                pair = m[c0:c0 + 2] if c0 + 2 <= c else torch.cat([m[c0:c0 + 1], m[c0:c0 + 1]])
                for label, t in self._separate_pair(pair).items():
                    parts.setdefault(label, []).append(t if c0 + 2 <= c else t[:1])
            return {label: torch.cat(ts) for label, ts in parts.items()}
        return self._separate_pair(m)

    def _separate_pair(self, m: torch.Tensor) -> Dict[str, torch.Tensor]:
        inst = self.model_instance
        if (self.chunker == "ola" and self.sharded and inst.predictor is not None and not inst.extra and inst.demucs is None
                and inst.vr is None and inst.roformer is None):
            return self._separate_pair_ola(m.contiguous(), inst)     # the runner uploads this rank's share only (m may live on the host)
        m = m.to(self.ctx.device).contiguous()
        if inst.demucs is not None:                             # all sources from one pass; labels as DemucsSeparator names its files
            return {name.capitalize(): t for name, t in inst.demucs.separate(m).items()}
        if inst.vr is not None:                                 # both stems come from the network's spectrogram split; the back end
            y, v = inst.vr.separate(m)                          # yields 480 * (n // 480) samples (as the reference's files): zero tail
            pad = lambda t: torch.nn.functional.pad(t, (0, m.shape[1] - t.shape[1]))
            return {inst.primary_stem_name: pad(y), inst.secondary_stem_name: pad(v)}
        if inst.roformer is not None:
            out = inst.roformer.separate(m)
            if inst.secondary_stem_name:                        # single-target model: the other stem is mix - target
                sec = m.clone()
                first = out[inst.primary_stem_name].contiguous()
                self.ctx.check(self.ctx.lib.alsep_axpby(self.ctx.handle, -1.0, _lib.ptr(first), 1.0, _lib.ptr(sec), sec.numel()), "alsep_axpby")
                out[inst.secondary_stem_name] = sec
            return out
        if self.chunker == "ola" and not inst.extra:
            return self._separate_pair_ola(m, inst)
        primary = inst.predictor.demix(m)
        if primary.dim() == 3:                                  # Predictor returns [1,2,N] like the reference
            primary = primary[0]
        out = {inst.primary_stem_name: primary}
        if inst.secondary_stem_name:
            sec = m.clone()                                     # secondary = mix - primary (mdxnet.py:211)
            self.ctx.check(self.ctx.lib.alsep_axpby(self.ctx.handle, -1.0, _lib.ptr(primary.contiguous()), 1.0, _lib.ptr(sec),
                                                    sec.numel()), "alsep_axpby")
            out[inst.secondary_stem_name] = sec
        for label, _, pred in inst.extra:                       # multi-stem model: every further stem from the same input
            t = pred.demix(m)
            out[label] = t[0] if t.dim() == 3 else t
        return out

    def _separate_pair_ola(self, m: torch.Tensor, inst: "_ModelInstance") -> Dict[str, torch.Tensor]:
        """``chunker="ola"``: the sequence of the third-party ``MDXSeparator.separate`` that AudioLab reaches at stem_separator.py:281
        (upstream, uncited -- UNPINNED; restated in oracle/mdx_oracle.py ``separate_ola``): the mix is normalised to ``normalization``
        (0.9: scaled down only if its peak is higher), the primary stem is the overlap-add of the model outputs, normalised the same
        way; the secondary stem is, with ``invert_using_spec`` (AudioLab's setting, stem_separator.py:104), the spectral inversion of
        the primary against the model-path rendition of the mix (the ``is_match_mix`` pass), else ``mix - primary``."""
        from . import ensemble
        if self.sharded and m.device != self.ctx.device:
            # host-resident programme, chunks sharded: the gain of the input normalisation is a scalar (global peak, one pass on the host),
            # applied by each rank to the samples it uploads; the stems come back whole from the runner's all-gather
            peak = float(m.abs().max()) if m.numel() else 0.0
            gain = self.normalization / peak if (self.normalization and peak > self.normalization) else 1.0
            primary = inst.predictor.demix(m, in_scale=gain)
            if self.normalization:
                primary = ensemble.normalize(self.ctx, primary, self.normalization)
            out = {inst.primary_stem_name: primary}
            if inst.secondary_stem_name:
                if self.invert_using_spec:
                    raw_mix = inst.predictor.demix(m, match_mix=True, in_scale=gain)
                    out[inst.secondary_stem_name] = ensemble.invert_stem(self.ctx, raw_mix, primary)
                else:
                    sec = m.to(self.ctx.device, dtype=torch.float32)
                    self.ctx.check(self.ctx.lib.alsep_axpby(self.ctx.handle, -1.0, _lib.ptr(primary.contiguous()), float(gain), _lib.ptr(sec),
                                                            sec.numel()), "alsep_axpby")
                    out[inst.secondary_stem_name] = sec
            return out
        m = m.to(self.ctx.device)
        mixn = ensemble.normalize(self.ctx, m.clone(), self.normalization) if self.normalization else m
        primary = inst.predictor.demix(mixn)
        if self.normalization:
            primary = ensemble.normalize(self.ctx, primary, self.normalization)
        out = {inst.primary_stem_name: primary}
        if inst.secondary_stem_name:
            if self.invert_using_spec:
                raw_mix = inst.predictor.demix(mixn, match_mix=True)
                out[inst.secondary_stem_name] = ensemble.invert_stem(self.ctx, raw_mix, primary)
            else:
                sec = mixn.clone()
                self.ctx.check(self.ctx.lib.alsep_axpby(self.ctx.handle, -1.0, _lib.ptr(primary.contiguous()), 1.0, _lib.ptr(sec), sec.numel()),
                               "alsep_axpby")
                out[inst.secondary_stem_name] = sec
        return out

    def separate(self, audio_file_path: str) -> List[str]:
        """File in, files out (the reference's calling convention, stem_separator.py:281): returns
        BASENAMES relative to ``output_dir``, each containing its ``(Label)`` tag."""
        if self.model_instance is None:
            raise AlsepError("load_model() first")
        audio, sr = wavio.read_wav(audio_file_path)
        if sr != self.sample_rate:                              # the models run at 44.1 kHz: resample as the reference's loader does
            from . import ensemble
            audio = ensemble.resample(self.ctx, torch.from_numpy(audio).to(self.ctx.device), sr, self.sample_rate)
            sr = self.sample_rate
        stems = self.separate_array(audio)
        out_dir = self.model_instance.output_dir or self.output_dir or os.path.dirname(audio_file_path)
        os.makedirs(out_dir, exist_ok=True)
        base = os.path.splitext(os.path.basename(audio_file_path))[0]
        model_tag = os.path.splitext(self.model_instance.model_name)[0]
        names = []
        for label, t in stems.items():
            name = f"{base}_({label})_{model_tag}.wav"
            wavio.write_wav(os.path.join(out_dir, name), t.cpu().numpy(), sr, subtype="FLOAT")
            names.append(name)
        return names
