"""MDX runner seam (reference handlers/patch_separate.py:11-78): replace ``MDXSeparator.load_model`` so
that ``self.model_run(spek[B,4,dim_f,T]) -> array-like`` is served by the HIP TFC-TDF network,
device tensor in / device tensor out (no ``spek.cpu().numpy()`` hop, :52).  Idempotent (:71-78).

What the patched method reads from the ``MDXSeparator`` instance is what the reference's patch reads
(``model_path, segment_size, dim_t, torch_device, logger``, :19,47,52,61) plus, when present, the
STFT geometry the third-party class keeps next to them (``n_fft``, ``hop_length``, ``dim_f``) -- those
are not stored in an ``.onnx`` file.  The weights and the network hyper-parameters come from
``self.model_path`` itself (audiolab_amd.onnx_reader); nothing is looked up by name, nothing is
random-initialised, and a file that cannot be read RAISES instead of leaving ``model_run = None``
(the reference's bare ``except``, :65-67 -- SURVEY Appendix E.10).

``segment_size != dim_t`` (:55-64: the reference converts the graph to a torch module so that it accepts
another frame count): the TFC-TDF network is convolutional along time, so the same weights are bound at
``dim_t = segment_size`` (must be divisible by 2**n).

One network per (file, mtime, frame count, device) is kept for the life of the process: the reference
reloads the session on every ``load_model`` (stem_separator.py:394).

When the third-party ``audio_separator`` package is not importable (as in this image)
``patch_separator`` returns False; ``patched_load_model`` itself works on any object with the
attributes above (tests bind it to a stand-in class registered under the third-party module path)."""
from __future__ import annotations

import dataclasses
import os
from typing import Dict, Tuple

og_load_model = None
_NETS: Dict[Tuple, object] = {}


def bind_model_run(obj, net) -> None:
    """Set ``obj.model_run`` to the GPU network (what patched_load_model leaves behind, :52,58-62)."""
    obj.model_run = net                      # TDFNet.__call__(spek) -> pred, reference layout


def _roster_n_fft(name: str):
    from audiolab_amd.engine import MODEL_ROSTER
    entry = MODEL_ROSTER.get(name)
    return entry[2].n_fft if entry and entry[0] != "multi" and entry[2] is not None else None


def patched_load_model(self):
    import torch

    from audiolab_amd import _lib
    from audiolab_amd._lib import AlsepError
    from audiolab_amd.onnx_reader import load_mdx_onnx
    from audiolab_amd.tdfnet import TDFNet
    path = self.model_path
    if not os.path.isfile(path):
        raise AlsepError(f"MDX model file not found: {path}")
    name = os.path.basename(path)
    n_fft = getattr(self, "n_fft", None) or _roster_n_fft(name)
    if n_fft is None:
        raise AlsepError(f"{name}: n_fft is neither on the MDXSeparator instance nor in this build's roster")
    hop = getattr(self, "hop_length", None) or 1024
    m = load_mdx_onnx(path, n_fft=int(n_fft), hop=int(hop))
    cfg = m.config
    want_f = getattr(self, "dim_f", None)
    if want_f is not None and int(want_f) != cfg.dim_f:
        raise AlsepError(f"{name}: the file's dim_f={cfg.dim_f} differs from MDXSeparator.dim_f={want_f}")
    dim_t = getattr(self, "dim_t", None)
    seg = getattr(self, "segment_size", None)
    frames = int(seg if seg is not None else (dim_t if dim_t is not None else cfg.dim_t))
    if dim_t is not None and int(dim_t) != cfg.dim_t:
        self.logger.warning(f"{name}: the file was exported for dim_t={cfg.dim_t}, MDXSeparator.dim_t={dim_t}")
    if frames != cfg.dim_t:                                    # :55-64 -- another segment size on the same weights
        if frames % (1 << cfg.n):
            raise AlsepError(f"{name}: segment_size {frames} is not divisible by 2**{cfg.n}")
        cfg = dataclasses.replace(cfg, dim_t=frames)
        self.logger.warning("segment_size differs from the model's dim_t: the network is bound at the segment size")
    dev = torch.device(getattr(self, "torch_device", None) or _lib.DEVICE_TYPE)
    if dev.type != _lib.DEVICE_TYPE:
        raise AlsepError(f"MDXSeparator.torch_device={dev}: this build runs on {_lib.DEVICE_TYPE} devices only (no CPU fallback)")
    key = (os.path.abspath(path), os.path.getmtime(path), frames, str(dev))
    net = _NETS.get(key)
    if net is None:
        net = TDFNet(cfg, m.state_dict, ctx=_lib.default_context(dev), dtype=torch.float32, max_batch=4)
        _NETS[key] = net
    bind_model_run(self, net)
    self.logger.debug("MDX model bound to the HIP TFC-TDF network")


def patch_separator() -> bool:
    global og_load_model
    if og_load_model is not None:
        return True
    try:
        from audio_separator.separator.architectures.mdx_separator import MDXSeparator
    except Exception:
        return False
    og_load_model = MDXSeparator.load_model
    MDXSeparator.load_model = patched_load_model
    return True
