"""The MDX runner seam (reference handlers/patch_separate.py:11-78) on the emulated kernels and on the GPU:
a stand-in ``MDXSeparator`` registered under the third-party module path is patched by ``patch_separator()``; its
``load_model()`` must bind ``model_run`` to the HIP network built from THE FILE at ``self.model_path`` (not from a roster
name, not from random weights), honour ``segment_size != dim_t`` (:55-64), be idempotent (:71-78) and raise on an
unreadable file (instead of the reference's silent ``model_run = None``, :65-67)."""
import logging
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import tdfnet_oracle
from tests.conftest import host, on
from tests.onnx_writer import write_mdx_onnx


class FakeMDXSeparator:
    """the attributes the reference's patch reads (:19,47,52,61) + the STFT geometry MDXSeparator keeps beside them"""

    def __init__(self, model_path, dim_t, segment_size, device, n_fft, hop_length, dim_f):
        self.model_path, self.dim_t, self.segment_size, self.torch_device = model_path, dim_t, segment_size, device
        self.n_fft, self.hop_length, self.dim_f = n_fft, hop_length, dim_f
        self.logger = logging.getLogger("fake_mdx")
        self.model_run = None

    def load_model(self):
        raise RuntimeError("the unpatched loader must not run")


@pytest.fixture()
def fake_package(monkeypatch):
    from audiolab_amd.handlers import patch_separate as ps
    names = ["audio_separator", "audio_separator.separator", "audio_separator.separator.architectures",
             "audio_separator.separator.architectures.mdx_separator"]
    for n in names:
        monkeypatch.setitem(sys.modules, n, types.ModuleType(n))
    sys.modules[names[-1]].MDXSeparator = FakeMDXSeparator
    monkeypatch.setattr(ps, "og_load_model", None)
    monkeypatch.setattr(ps, "_NETS", {})
    monkeypatch.setattr(FakeMDXSeparator, "load_model", FakeMDXSeparator.load_model)    # restored after the test
    return ps


def _write(tmp_path, seed=3, **kw):
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNetConfig
    base = dict(dim_f=64, dim_t=16, n_fft=256, hop=64, num_blocks=5, l=2, g=16, bn=4)
    base.update(kw)
    cfg = TDFNetConfig(**base)
    sd = synthetic_state_dict(cfg, seed=seed)
    gen = torch.Generator().manual_seed(seed)
    for k in list(sd):
        if k.endswith("running_mean"):
            sd[k] = 0.05 * torch.randn(sd[k].shape, generator=gen)
        elif k.endswith("running_var"):
            sd[k] = 0.8 + 0.4 * torch.rand(sd[k].shape, generator=gen)
    path = os.path.join(tmp_path, "UVR-MDX-NET-Voc_FT.onnx")       # a roster NAME: the file must win over the table
    write_mdx_onnx(path, sd, cfg)
    return path, cfg, sd


def test_patched_load_model_binds_the_file(dev, tmp_path, fake_package):
    ps = fake_package
    path, cfg, sd = _write(tmp_path)
    assert ps.patch_separator() is True and ps.patch_separator() is True           # idempotent
    assert FakeMDXSeparator.load_model is ps.patched_load_model
    obj = FakeMDXSeparator(path, cfg.dim_t, cfg.dim_t, dev.device, cfg.n_fft, cfg.hop, cfg.dim_f)
    obj.load_model()
    x = torch.randn(2, 4, cfg.dim_f, cfg.dim_t, generator=torch.Generator().manual_seed(5))
    want = tdfnet_oracle.forward(sd, x, cfg.num_blocks, cfg.l, cfg.bn)
    got = obj.model_run(on(dev, x))
    assert got.device.type == dev.device.type and tuple(got.shape) == tuple(x.shape)   # device tensor out, no host hop
    assert float(np.max(np.abs(host(got) - want.numpy()))) < 1e-4 * max(1.0, float(want.abs().max()))
    # one network per file for the life of the process
    again = FakeMDXSeparator(path, cfg.dim_t, cfg.dim_t, dev.device, cfg.n_fft, cfg.hop, cfg.dim_f)
    again.load_model()
    assert again.model_run is obj.model_run


def test_segment_size_other_than_dim_t(dev, tmp_path, fake_package):
    ps = fake_package
    path, cfg, sd = _write(tmp_path, seed=8)
    ps.patch_separator()
    seg = 2 * cfg.dim_t
    obj = FakeMDXSeparator(path, cfg.dim_t, seg, dev.device, cfg.n_fft, cfg.hop, cfg.dim_f)
    obj.load_model()
    x = torch.randn(1, 4, cfg.dim_f, seg, generator=torch.Generator().manual_seed(6))
    want = tdfnet_oracle.forward(sd, x, cfg.num_blocks, cfg.l, cfg.bn)
    got = obj.model_run(on(dev, x))
    assert float(np.max(np.abs(host(got) - want.numpy()))) < 1e-4 * max(1.0, float(want.abs().max()))


def test_load_failures_raise(dev, tmp_path, fake_package):
    from audiolab_amd._lib import AlsepError
    ps = fake_package
    ps.patch_separator()
    missing = FakeMDXSeparator(os.path.join(tmp_path, "nope.onnx"), 16, 16, dev.device, 256, 64, 64)
    with pytest.raises(AlsepError, match="not found"):
        missing.load_model()
    bad = os.path.join(tmp_path, "bad.onnx")
    with open(bad, "wb") as f:
        f.write(b"\x08\x07")
    broken = FakeMDXSeparator(bad, 16, 16, dev.device, 256, 64, 64)
    with pytest.raises(AlsepError):
        broken.load_model()
    assert broken.model_run is None                          # nothing was bound
    path, cfg, _ = _write(tmp_path)
    wrong_f = FakeMDXSeparator(path, cfg.dim_t, cfg.dim_t, dev.device, cfg.n_fft, cfg.hop, cfg.dim_f * 2)
    with pytest.raises(AlsepError, match="dim_f"):
        wrong_f.load_model()


def test_no_package_is_a_noop(monkeypatch):
    from audiolab_amd.handlers import patch_separate as ps
    monkeypatch.setattr(ps, "og_load_model", None)
    for n in list(sys.modules):
        if n.startswith("audio_separator"):
            monkeypatch.delitem(sys.modules, n)
    assert ps.patch_separator() is False
