"""CPU: the gfx950 library loads (no GPU needed to dlopen it) and exports every symbol that
include/alsep.h declares; the product loader has no fallback."""
import os
import re

import pytest


def header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "alsep.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(alsep_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from audiolab_amd import _lib
    ge.build()
    lib = _lib.bind(_lib.LIB_PATH)
    declared = header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in alsep.h but not exported"
        assert name in _lib.EXPORTS, f"{name} has no ctypes prototype"
    assert lib.alsep_abi_version() == _lib.ABI_VERSION
    assert lib.alsep_plan_supported_nfft(6144) == 1 and lib.alsep_plan_supported_nfft(7680) == 1
    assert lib.alsep_plan_supported_nfft(1000) == 0


def test_missing_extension_fails_loudly(monkeypatch, tmp_path):
    from audiolab_amd import _lib
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.AlsepError, match="no CPU fallback"):
        _lib.get_lib()


def test_cpu_tensors_are_rejected():
    import torch
    from audiolab_amd import _lib
    with pytest.raises(_lib.AlsepError):
        _lib.ptr(torch.zeros(4))
