"""MDX runner seam (reference handlers/patch_separate.py:11-78): replace ``MDXSeparator.load_model`` so
that ``self.model_run(spek[B,4,dim_f,dim_t]) -> array-like`` is served by the HIP TFC-TDF network,
device tensor in / device tensor out (no ``spek.cpu().numpy()`` hop, :52).  Idempotent.  When the
third-party ``audio_separator`` package is not importable (as in this image) it is a no-op that
returns False; ``bind_model_run`` is the same binding for any object with the attributes the
reference's patch reads (``segment_size, dim_t, model_path, torch_device, logger``)."""
from __future__ import annotations

import os

og_load_model = None


def bind_model_run(obj, net) -> None:
    """Set ``obj.model_run`` to the GPU network (what patched_load_model leaves behind, :52,58-62)."""
    obj.model_run = net                      # TDFNet.__call__(spek) -> pred, reference layout


def patched_load_model(self):
    from audiolab_amd.engine import MODEL_ROSTER, Separator
    name = os.path.basename(self.model_path)
    if name not in MODEL_ROSTER:
        raise RuntimeError(f"{name}: not in this build's MDX-Net roster")      # no silent model_run=None (:65-67)
    eng = Separator(use_autocast=False)
    eng.load_model(name)
    bind_model_run(self, eng.model_instance.net)
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
