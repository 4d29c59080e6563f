"""CPU: the ONNX protobuf reader + TFC-TDF graph walker (audiolab_amd/onnx_reader.py) against files written by
tests/onnx_writer.py in the torch.onnx export style, checked through the torch oracle and the emulated kernels."""
import os

import numpy as np
import pytest
import torch

from oracle import tdfnet_oracle
from tests.onnx_writer import node, tensor, value_info, write_mdx_onnx, _i, _ld


def _cfg(**kw):
    from audiolab_amd.tdfnet import TDFNetConfig
    base = dict(dim_f=64, dim_t=16, n_fft=128, num_blocks=5, l=2, g=16, bn=4)
    base.update(kw)
    return TDFNetConfig(**base)


def _randomised(cfg, seed):
    """synthetic weights with non-trivial BatchNorm statistics and biases everywhere"""
    from audiolab_amd.synth import synthetic_state_dict
    sd = synthetic_state_dict(cfg, seed=seed)
    gen = torch.Generator().manual_seed(seed + 1)
    for k in list(sd):
        if k.endswith("running_mean") or (k.endswith(".bias") and ".1." in k or k.endswith(("1.bias", "4.bias"))):
            sd[k] = 0.1 * torch.randn(sd[k].shape, generator=gen)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(sd[k].shape, generator=gen)
        elif k.endswith(".weight") and sd[k].dim() == 1:
            sd[k] = sd[k] * (0.75 + 0.5 * torch.rand(sd[k].shape, generator=gen))
    return sd


@pytest.mark.parametrize("kw,style,as_inputs", [({}, "raw", False), ({"bn": 0, "l": 3}, "packed", True),
                                                ({"num_blocks": 3, "g": 32, "dim_f": 96, "bn": 8, "n_fft": 256}, "raw", False)])
def test_reader_recovers_config_and_function(tmp_path, kw, style, as_inputs):
    from audiolab_amd.onnx_reader import load_mdx_onnx, read_graph
    cfg = _cfg(**kw)
    sd = _randomised(cfg, 7)
    path = os.path.join(tmp_path, "toy.onnx")
    write_mdx_onnx(path, sd, cfg, float_style=style, weights_as_inputs=as_inputs)
    g = read_graph(path)
    assert [n for n, _ in g.inputs] == ["input"] and g.inputs[0][1] == [None, 4, cfg.dim_f, cfg.dim_t]
    m = load_mdx_onnx(path, n_fft=cfg.n_fft)
    assert m.input_name == "input"                           # the name the reference feeds (patch_separate.py:52)
    assert m.config == cfg
    assert "first_conv.1.weight" not in m.state_dict          # exporter-fused BatchNorm stays fused
    x = torch.randn(2, 4, cfg.dim_f, cfg.dim_t, generator=torch.Generator().manual_seed(3))
    want = tdfnet_oracle.forward(sd, x, cfg.num_blocks, cfg.l, cfg.bn)
    got = tdfnet_oracle.forward(m.state_dict, x, cfg.num_blocks, cfg.l, cfg.bn)
    assert float((got - want).abs().max()) < 2e-5 * max(1.0, float(want.abs().max()))


def test_loaded_weights_run_on_the_kernels(dev, tmp_path):
    from audiolab_amd.onnx_reader import load_mdx_onnx
    from audiolab_amd.tdfnet import TDFNet
    cfg = _cfg(n_fft=256, hop=64)
    sd = _randomised(cfg, 11)
    path = os.path.join(tmp_path, "toy.onnx")
    write_mdx_onnx(path, sd, cfg)
    m = load_mdx_onnx(path, n_fft=cfg.n_fft, hop=cfg.hop)
    assert m.config == cfg
    net = TDFNet(m.config, m.state_dict, ctx=dev, dtype=torch.float32)
    x = torch.randn(1, 4, cfg.dim_f, cfg.dim_t, generator=torch.Generator().manual_seed(5))
    want = tdfnet_oracle.forward(sd, x, cfg.num_blocks, cfg.l, cfg.bn)
    got = net(x.to(dev.device)).cpu()
    assert float((got - want).abs().max()) < 1e-4 * max(1.0, float(want.abs().max()))


def test_foreign_graphs_are_refused(tmp_path):
    from audiolab_amd._lib import AlsepError
    from audiolab_amd.onnx_reader import load_mdx_onnx
    path = os.path.join(tmp_path, "bad.onnx")
    with open(path, "wb") as f:
        f.write(b"\x08\x07")                                  # ir_version only
    with pytest.raises(AlsepError, match="no graph"):
        load_mdx_onnx(path)
    # a graph with an operator that is not part of MDX-Net
    graph = _ld(1, node("Softmax", ["input"], ["output"], name="sm", axis=1)) + _ld(11, value_info("input", [1, 4, 8, 8]))
    with open(path, "wb") as f:
        f.write(_i(1, 7) + _ld(7, graph))
    with pytest.raises(AlsepError, match="Softmax"):
        load_mdx_onnx(path)
    # right operators, wrong order: a lone convolution
    w = np.zeros((4, 4, 1, 1), np.float32)
    graph = (_ld(1, node("Conv", ["input", "w"], ["output"], name="c", kernel_shape=[1, 1], pads=[0] * 4, strides=[1, 1]))
             + _ld(5, tensor("w", w)) + _ld(11, value_info("input", [1, 4, 8, 8])))
    with open(path, "wb") as f:
        f.write(_i(1, 7) + _ld(7, graph))
    with pytest.raises(AlsepError, match="not a TFC-TDF"):
        load_mdx_onnx(path)
    # truncated file
    cfg = _cfg()
    good = os.path.join(tmp_path, "good.onnx")
    write_mdx_onnx(good, _randomised(cfg, 1), cfg)
    blob = open(good, "rb").read()
    with open(path, "wb") as f:
        f.write(blob[:len(blob) // 2])
    with pytest.raises(AlsepError):
        load_mdx_onnx(path)


def test_engine_prefers_the_model_file(dev, tmp_path):
    """Separator.load_model (stem_separator.py:394 call site): an .onnx present in model_file_dir supplies weights and
    hyper-parameters; the roster only contributes n_fft and the stem labels."""
    from audiolab_amd.engine import Separator
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    cfg = _cfg(n_fft=256, hop=64)
    sd = _randomised(cfg, 23)
    write_mdx_onnx(os.path.join(tmp_path, "toy_vocals.onnx"), sd, cfg)
    roster = {"toy_vocals.onnx": ("Vocals", "Instrumental", TDFNetConfig(dim_f=64, dim_t=16, n_fft=256, hop=64, g=48))}
    sep = Separator(model_file_dir=str(tmp_path), ctx=dev, dtype=torch.float32, roster=roster)
    sep.load_model("toy_vocals.onnx")
    assert sep.model_instance.net.cfg == cfg                  # L, l, g, bn from the graph, not from the roster
    mix = torch.randn(2, 3000, generator=torch.Generator().manual_seed(2)) * 0.1
    got = sep.separate_array(mix)
    ref = Separator(model_file_dir=str(tmp_path / "none"), ctx=dev, dtype=torch.float32, allow_synthetic=True,
                    roster={"toy_vocals.onnx": ("Vocals", "Instrumental", cfg)})
    ref.load_model("toy_vocals.onnx")                         # synthetic weights: a different network
    direct = TDFNet(cfg, sd, ctx=dev, dtype=torch.float32)
    ref.model_instance.predictor.model = direct               # ... swapped for the original weights
    want = ref.separate_array(mix)
    assert float((got["Vocals"] - want["Vocals"]).abs().max()) < 1e-5
    assert set(got) == {"Vocals", "Instrumental"}
    # outside the roster and not on disk: refused
    from audiolab_amd._lib import AlsepError
    with pytest.raises(AlsepError):
        sep.load_model("missing.onnx")
    # in the roster but without a weight file: refused unless synthetic weights were asked for explicitly
    bare = Separator(model_file_dir=str(tmp_path / "none"), ctx=dev, dtype=torch.float32,
                     roster={"toy_vocals.onnx": ("Vocals", "Instrumental", cfg)})
    with pytest.raises(AlsepError, match="allow_synthetic"):
        bare.load_model("toy_vocals.onnx")
    assert sep.weights_provenance() == "real" and ref.weights_provenance() == "synthetic"
