import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


# ------------------------------------------------------------------------------------------
# CPU emulation of the HIP kernels (tests/cpu_emul): TEST-ONLY.  The product never loads it;
# the fixture below swaps the bound library inside audiolab_amd._lib for the duration of a
# test so that the unchanged Python host code drives the unchanged kernel sources on the CPU.
# ------------------------------------------------------------------------------------------
EMUL_DIR = os.path.join(ROOT, "tests", "cpu_emul")
EMUL_SO = os.path.join(EMUL_DIR, "libalsep_emul.so")


def _emul_stale() -> bool:
    if not os.path.exists(EMUL_SO):
        return True
    built = os.path.getmtime(EMUL_SO)
    srcs = [os.path.join(ROOT, "audiolab_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "audiolab_amd", "csrc"))]
    srcs += [os.path.join(EMUL_DIR, "emul_runtime.cpp"), os.path.join(EMUL_DIR, "hip", "hip_runtime.h"),
             os.path.join(EMUL_DIR, "alsep_gfx950_asm.h"),
             os.path.join(ROOT, "include", "alsep.h")]
    return any(os.path.getmtime(s) > built for s in srcs)


@pytest.fixture(scope="session")
def emul_lib_path():
    import subprocess
    if _emul_stale():
        subprocess.run([os.path.join(EMUL_DIR, "build_emul.sh")], check=True, capture_output=True)
    return EMUL_SO


@pytest.fixture()
def emul(emul_lib_path, monkeypatch):
    """audiolab_amd bound to the CPU-emulated kernels; yields a Context on the CPU."""
    from audiolab_amd import _lib
    lib = _lib.bind(emul_lib_path)
    monkeypatch.setattr(_lib, "_LIB", lib)
    monkeypatch.setattr(_lib, "DEVICE_TYPE", "cpu")
    monkeypatch.setattr(_lib, "_DEFAULT_CTX", {})
    ctx = _lib.Context("cpu")
    yield ctx
    ctx.close()


# ------------------------------------------------------------------------------------------
# ``dev``: the same test body on the CPU emulation (-m "not gpu") and on the real GPU (-m gpu).
# The pinned comparisons (golden vectors of the reference, oracle compositions) are therefore
# executed by the HIP kernels themselves on the GPU box, not only by their host emulation.
# ------------------------------------------------------------------------------------------
@pytest.fixture(scope="session")
def gpu_ctx():
    import torch
    from audiolab_amd import _lib
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return _lib.Context("cuda:0")


@pytest.fixture(params=["emul", pytest.param("gpu", marks=pytest.mark.gpu)])
def dev(request):
    """A Context: the emulated kernels on CPU tensors, or libalsep.so on cuda:0."""
    if request.param == "emul":
        return request.getfixturevalue("emul")
    return request.getfixturevalue("gpu_ctx")


def on(ctx, a):
    """numpy array / tensor -> tensor on the context's device."""
    import numpy as np
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a)) if isinstance(a, np.ndarray) else a
    return t.to(ctx.device)


def host(t):
    """device tensor -> numpy"""
    return t.detach().cpu().numpy()
