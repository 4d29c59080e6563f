"""Reader for demucs ``.th`` model packages -- the files behind ``self.separator.load_model("htdemucs_6s.yaml")`` of the reference
(modules/separator/stem_separator.py:110, :466): the yaml names a bag of model signatures, each a ``<signature>-<checksum>.th`` written by
``demucs.states.save_with_checksum`` = ``torch.save({"klass": <class>, "args": ..., "kwargs": {...}, "state": state_dict,
"training_args": ...})`` (demucs>=4.0.1, requirements.txt:19; upstream, uncited -- PARITY UNPINNED).

Such a file pickles references to demucs / omegaconf classes, so ``torch.load(weights_only=True)`` refuses it and a plain ``torch.load``
would import -- and run -- whatever the file names.  This reader unpickles with an allow-list: tensors and their storages, the
containers and scalars a package holds, ``fractions.Fraction`` (demucs stores ``segment`` as one); every global from the packages that are
expected to appear but are not needed (demucs, omegaconf, dora, ...) resolves to an inert placeholder that only records how it was
built; anything else aborts the load.  Nothing from the file is ever executed.
"""
from __future__ import annotations

import collections
import fractions
import os
import pickle
from typing import Dict, Tuple

import torch

from ._lib import AlsepError

_STUB_PACKAGES = ("demucs", "omegaconf", "dora", "hydra", "submitit", "julius", "openunmix", "typing", "pathlib", "numpy", "argparse", "enum")


class Placeholder:
    """stands in for a class / function of a package that is not loaded: callable, settable, inert"""
    _alsep_path = "?"

    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __setstate__(self, state):
        self.state = state

    def __call__(self, *args, **kwargs):
        return self

    def __repr__(self):
        return f"<placeholder {self._alsep_path}>"


def _placeholder(module: str, name: str):
    return type(name, (Placeholder,), {"_alsep_path": f"{module}.{name}"})


def _build_allowed() -> Dict[Tuple[str, str], object]:
    import torch._utils
    table = {
        ("collections", "OrderedDict"): collections.OrderedDict,
        ("collections", "defaultdict"): collections.defaultdict,
        ("fractions", "Fraction"): fractions.Fraction,
        ("builtins", "set"): set, ("builtins", "frozenset"): frozenset, ("builtins", "slice"): slice, ("builtins", "complex"): complex,
        ("builtins", "dict"): dict, ("builtins", "list"): list, ("builtins", "tuple"): tuple, ("builtins", "int"): int,
        ("builtins", "float"): float, ("builtins", "bool"): bool, ("builtins", "str"): str, ("builtins", "bytes"): bytes,
        ("torch", "Size"): torch.Size, ("torch", "device"): torch.device,
    }
    # symbols that come and go between torch builds: allow the ones this build has
    for module, owner, name in (("torch._utils", torch._utils, "_rebuild_tensor_v2"), ("torch._utils", torch._utils, "_rebuild_tensor"),
                                ("torch._utils", torch._utils, "_rebuild_parameter"),
                                ("torch.serialization", torch.serialization, "_get_layout"),
                                ("torch.storage", torch.storage, "UntypedStorage"), ("torch.storage", torch.storage, "TypedStorage")):
        obj = getattr(owner, name, None)
        if obj is not None:
            table[(module, name)] = obj
    for name in ("float32", "float16", "bfloat16", "float64", "int64", "int32", "int16", "int8", "uint8", "bool",
                 "FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage", "IntStorage", "ShortStorage", "CharStorage",
                 "ByteStorage", "BoolStorage"):
        obj = getattr(torch, name, None)
        if obj is not None:
            table[("torch", name)] = obj
    return table


_ALLOWED: Dict[Tuple[str, str], object] = {}


def _allowed() -> Dict[Tuple[str, str], object]:
    """the allow-list, built once per process"""
    if not _ALLOWED:
        _ALLOWED.update(_build_allowed())
    return _ALLOWED


class RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module: str, name: str):
        hit = _allowed().get((module, name))
        if hit is not None:
            return hit
        if module.split(".")[0] in _STUB_PACKAGES:
            return _placeholder(module, name)
        raise pickle.UnpicklingError(f"global '{module}.{name}' is not allowed in a model package (only tensors, plain containers and "
                                     f"placeholders for demucs / omegaconf classes are)")


class _PickleModule:
    """what ``torch.load(pickle_module=...)`` needs"""
    __name__ = "audiolab_amd.th_reader"
    Unpickler = RestrictedUnpickler
    UnpicklingError = pickle.UnpicklingError

    @staticmethod
    def load(f, **kwargs):
        return RestrictedUnpickler(f, **kwargs).load()


def read_th(path: str) -> dict:
    """-> {"klass": class name, "kwargs": dict, "state": {name: float32 CPU tensor}}; raises AlsepError on anything else"""
    try:
        pkg = torch.load(path, map_location="cpu", pickle_module=_PickleModule, weights_only=False)
    except pickle.UnpicklingError as e:
        raise AlsepError(f"{path}: refused -- {e}") from e
    if not isinstance(pkg, dict) or "state" not in pkg or "klass" not in pkg:
        raise AlsepError(f"{path}: not a demucs model package (expected the keys klass / args / kwargs / state)")
    klass = pkg["klass"]
    klass_name = klass.__name__ if isinstance(klass, type) else type(klass).__name__
    state = pkg["state"]
    if not isinstance(state, dict) or not all(isinstance(v, torch.Tensor) for v in state.values()):
        raise AlsepError(f"{path}: 'state' is not a dictionary of tensors (quantised packages are not supported)")
    kwargs = pkg.get("kwargs") or {}
    if not isinstance(kwargs, dict):
        raise AlsepError(f"{path}: 'kwargs' is not a dictionary")
    return {"klass": klass_name, "kwargs": dict(kwargs), "state": {k: v.detach().float() for k, v in state.items()}}


# demucs.htdemucs.HTDemucs.__init__'s own defaults (demucs 4.0.1, upstream; restated -- unpinned) for every option that is not a field of
# HTDemucsConfig: a package that omits a key gets THIS value, which is then held against what this build implements.
_DEMUCS_DEFAULTS = {
    "channels_time": None, "wiener_iters": 0, "end_iters": 0, "wiener_residual": False, "cac": True, "rewrite": True, "multi_freqs": None,
    "multi_freqs_depth": 3, "emb_smooth": True, "time_stride": 2, "norm_starts": 4, "norm_groups": 4, "dconv_mode": 1, "bottom_channels": 0,
    "t_emb": "sin", "t_dropout": 0.0, "t_max_positions": 10000, "t_norm_in": True, "t_norm_in_group": False, "t_group_norm": False,
    "t_norm_first": True, "t_norm_out": True, "t_weight_decay": 0.0, "t_lr": None, "t_layer_scale": True, "t_gelu": True,
    "t_sin_random_shift": 0, "t_cape_mean_normalize": True, "t_cape_augment": True, "t_cape_glob_loc_scale": [5000.0, 1.0, 1.4],
    "t_sparse_self_attn": False, "t_sparse_cross_attn": False, "t_mask_type": "diag", "t_mask_random_seed": 42, "t_sparse_attn_window": 500,
    "t_global_window": 100, "t_sparsity": 0.95, "t_auto_sparsity": False, "t_cross_first": False, "rescale": 0.1, "use_train_segment": True,
}
# options that only act at training time or at construction (initial scales, dropout, optimiser groups, the sparse-attention mask of
# switched-off sparse attention, augmentation of a positional embedding that is not the one in use): any value is fine
_TRAINING_ONLY = {"t_dropout", "t_weight_decay", "t_lr", "rescale", "t_sin_random_shift", "t_cape_mean_normalize", "t_cape_augment",
                  "t_cape_glob_loc_scale", "t_mask_type", "t_mask_random_seed", "t_sparse_attn_window", "t_global_window", "t_sparsity",
                  "t_auto_sparsity", "use_train_segment", "t_max_positions", "multi_freqs_depth", "end_iters", "wiener_residual"}


def htdemucs_config_from_kwargs(kwargs: dict):
    """demucs.htdemucs.HTDemucs(**kwargs) -> HTDemucsConfig.  An option this build's network does not implement must sit at the value
    htdemucs_6s was trained with -- whether the package spells it out or leaves it to demucs' own default -- and a keyword this reader
    does not know at all is an error too: nothing is silently dropped."""
    import dataclasses
    from .htdemucs import HTDemucsConfig
    base = HTDemucsConfig()
    fields = {f.name for f in dataclasses.fields(HTDemucsConfig)}
    over = {}
    for k, v in kwargs.items():
        if k == "sources":
            over["sources"] = tuple(str(s) for s in v)
        elif k == "segment":
            sr = int(kwargs.get("samplerate", base.samplerate))
            over["segment_samples"] = int(fractions.Fraction(v) * sr) if not isinstance(v, float) else int(v * sr)
        elif k in fields:
            over[k] = type(getattr(base, k))(v)
        elif k not in _DEMUCS_DEFAULTS:
            raise AlsepError(f"demucs package: HTDemucs({k}={v!r}) is not an option this build knows")
    # bottom_channels is a field here with htdemucs_6s' value as its default; demucs' own default is 0 (no bottleneck convolutions)
    if "bottom_channels" not in kwargs:
        over["bottom_channels"] = _DEMUCS_DEFAULTS["bottom_channels"]
    # structural switches of demucs' HTDemucs that must sit at the values htdemucs_6s was trained with
    expect = {"cac": True, "wiener_iters": 0, "multi_freqs": None, "channels_time": None, "emb_smooth": True, "time_stride": 2,
              "t_cross_first": False, "t_norm_in": True, "t_norm_in_group": False, "t_group_norm": False, "t_norm_first": True,
              "t_norm_out": True, "t_layer_scale": True, "t_gelu": True, "t_sparse_self_attn": False, "t_sparse_cross_attn": False,
              "t_emb": "sin", "dconv_mode": 3, "rewrite": True, "norm_starts": 4, "norm_groups": 4}
    for k, want in expect.items():
        have = kwargs.get(k, _DEMUCS_DEFAULTS[k])
        if have != want and not (want is None and not have):
            where = "" if k in kwargs else " (left at demucs' default by the package)"
            raise AlsepError(f"demucs package: HTDemucs({k}={have!r}){where} is not implemented by this build (expects {want!r})")
    if over.get("bottom_channels", base.bottom_channels) <= 0:
        raise AlsepError("demucs package: HTDemucs(bottom_channels=0) is not implemented by this build (htdemucs_6s uses 512)")
    return dataclasses.replace(base, **over)


def resolve_demucs_yaml(model_file_dir: str, yaml_name: str):
    """``<dir>/<name>.yaml`` of demucs' remote model zoo (``models: [signature, ...]``, optional ``weights``) -> the .th path of its one
    model; None when the yaml is absent.  Bags of several models (htdemucs_ft) are not implemented."""
    ypath = os.path.join(model_file_dir, yaml_name)
    if not os.path.isfile(ypath):
        return None
    import yaml
    y = yaml.safe_load(open(ypath)) or {}
    sigs = list(y.get("models") or [])
    if len(sigs) != 1:
        raise AlsepError(f"{ypath}: a bag of {len(sigs)} models -- only single-model bags (htdemucs_6s) are implemented")
    sig = str(sigs[0])
    hits = sorted(f for f in os.listdir(model_file_dir) if f.startswith(sig) and f.endswith(".th"))
    if not hits:
        raise AlsepError(f"{ypath} names model '{sig}', but no {sig}*.th is in {model_file_dir}")
    return os.path.join(model_file_dir, hits[0])
