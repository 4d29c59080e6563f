"""Host-side loader logic (no GPU): the hyper-parameter yaml of a .ckpt model (``!!python/tuple``, alternative file names, the
missing-file warning) and the shape check every network runs on its state_dict before anything is uploaded -- a checkpoint whose
hyper-parameters differ from the configuration must be refused with the name of the hyper-parameter, never read out of bounds or run
truncated."""
import dataclasses
import logging

import pytest
import torch

from audiolab_amd._lib import AlsepError


def test_yaml_python_tuple_and_alternative_names(tmp_path, caplog):
    from audiolab_amd.engine import YAML_NAMES, load_model_yaml
    name = "model_bs_roformer_ep_368_sdr_12.9628.ckpt"
    (tmp_path / "model_bs_roformer_ep_368_sdr_12.9628.yaml").write_text(
        "audio:\n  chunk_size: 352800\nmodel:\n  dim: 512\n  depth: 12\n  freqs_per_bands: !!python/tuple\n  - 2\n  - 2\n  - 4\n"
        "  multi_stft_resolutions_window_sizes: !!python/tuple [4096, 2048]\n")
    y = load_model_yaml(str(tmp_path), name)
    assert y["model"]["freqs_per_bands"] == (2, 2, 4) and y["model"]["multi_stft_resolutions_window_sizes"] == (4096, 2048)
    # arbitrary python objects stay refused (only the tuple tag is added to the safe schema)
    (tmp_path / "evil.yaml").write_text("a: !!python/object/apply:os.system ['true']\n")
    import yaml
    with pytest.raises(yaml.YAMLError):
        load_model_yaml(str(tmp_path), "evil.ckpt")
    # a model whose yaml carries another name than the checkpoint
    ck = "MDX23C-8KFFT-InstVoc_HQ.ckpt"
    (tmp_path / YAML_NAMES[ck][0]).write_text("audio:\n  dim_f: 4096\n")
    assert load_model_yaml(str(tmp_path), ck) == {"audio": {"dim_f": 4096}}
    # nothing there: None and a WARNING naming the candidates
    with caplog.at_level(logging.WARNING, logger="audiolab_amd.engine"):
        assert load_model_yaml(str(tmp_path), "melband_roformer_big_beta4.ckpt") is None
    assert any("config_melbandroformer_big_beta4.yaml" in r.getMessage() for r in caplog.records)


def _cases():
    from audiolab_amd import htdemucs as H, mdx23c as M, roformer as R
    return [
        ("roformer-depth", R, R.RoformerConfig(kind="bs", dim=32, depth=2, heads=2, dim_head=16), dict(depth=3), "depth"),
        ("roformer-dim", R, R.RoformerConfig(kind="mel", dim=32, depth=1, heads=2, dim_head=16, num_bands=12, n_fft=512, hop=128), dict(dim=48), "dim="),
        ("roformer-heads", R, R.RoformerConfig(kind="bs", dim=32, depth=1, heads=2, dim_head=16), dict(heads=4), "heads"),
        ("roformer-bands", R, R.RoformerConfig(kind="mel", dim=32, depth=1, heads=2, dim_head=16, num_bands=12, n_fft=512, hop=128), dict(num_bands=16), "band"),
        ("roformer-stems", R, R.RoformerConfig(kind="bs", dim=32, depth=1, heads=2, dim_head=16), dict(num_stems=2), "num_stems"),
        ("mdx23c-dimf", M, M.MDX23CConfig(dim_f=256, n_fft=512, hop=128, num_scales=2, num_channels=16, growth=8), dict(dim_f=512, n_fft=1024), "dim_f"),
        ("mdx23c-scales", M, M.MDX23CConfig(dim_f=256, n_fft=512, hop=128, num_scales=2, num_channels=16, growth=8), dict(num_scales=3), "num_scales"),
        ("mdx23c-instruments", M, M.MDX23CConfig(dim_f=256, n_fft=512, hop=128, num_scales=2, num_channels=16, growth=8),
         dict(instruments=("kick", "snare", "toms", "hh", "ride", "crash")), "instruments"),
        ("htdemucs-depth", H, H.HTDemucsConfig(channels=16, depth=2, nfft=256, bottom_channels=32, t_layers=3, t_heads=4), dict(depth=3), "depth"),
        ("htdemucs-tlayers", H, H.HTDemucsConfig(channels=16, depth=2, nfft=256, bottom_channels=32, t_layers=3, t_heads=4), dict(t_layers=5), "t_layers"),
        ("htdemucs-sources", H, H.HTDemucsConfig(channels=16, depth=2, nfft=256, bottom_channels=32, t_layers=3, t_heads=4),
         dict(sources=("drums", "bass", "other", "vocals")), "sources"),
    ]


@pytest.mark.parametrize("tag", [c[0] for c in _cases()])
def test_state_dict_of_other_hyper_parameters_is_refused(emul, tag):
    """weights made for configuration B loaded under configuration A: AlsepError naming the hyper-parameter (the yaml key to set)"""
    _, mod, cfg, other, word = next(c for c in _cases() if c[0] == tag)
    net_cls = {"audiolab_amd.roformer": "Roformer", "audiolab_amd.mdx23c": "MDX23C", "audiolab_amd.htdemucs": "HTDemucs"}[mod.__name__]
    sd_other = mod.synthetic_state_dict(dataclasses.replace(cfg, **other), 0)
    with pytest.raises(AlsepError) as e:
        getattr(mod, net_cls)(cfg, sd_other, ctx=emul)
    assert word in str(e.value), str(e.value)
    # and the matching weights load
    getattr(mod, net_cls)(cfg, mod.synthetic_state_dict(cfg, 0), ctx=emul)


def test_expected_shapes_cover_the_synthetic_state_dicts():
    """the shape tables and the synthetic-weight builders describe the same networks, at the default (full) sizes too"""
    from audiolab_amd import htdemucs as H, mdx23c as M, roformer as R
    for mod, cfg in ((H, H.HTDemucsConfig()), (R, R.RoformerConfig(kind="bs", depth=1)), (R, R.RoformerConfig(kind="mel", depth=1)),
                     (M, M.MDX23CConfig(num_channels=16, growth=8))):
        exp = mod.expected_shapes(cfg)
        sd = mod.synthetic_state_dict(cfg, 0)
        assert set(exp) == set(sd)
        assert all(tuple(sd[k].shape) == tuple(v[0]) for k, v in exp.items())
