"""CPU (-m "not gpu"): the VR-architecture network (audiolab_amd/vrnet.py over csrc/vrnet.hip, emulated) against the outputs
of the REFERENCE's own ``CascadedASPPNet`` (modules/rvc/infer/lib/uvr5_pack/lib_v5/nets*.py) recorded in
tests/golden/vrnet.npz by oracle/make_golden_vr.py.  Weights are regenerated from the seed on this side."""
import os

import numpy as np
import pytest
import torch


def load_case(z, k):
    n_fft, frames, seed, split = (int(v) for v in z[f"c{k}_cfg"])
    value = float(z[f"c{k}_aggr"][0])
    aggr = None if split < 0 else {"split_bin": split, "value": value}
    return str(z[f"c{k}_variant"]), n_fft, seed, aggr, z[f"c{k}_x"], z[f"c{k}_y"]


@pytest.mark.parametrize("k", [0, 1, 2])
def test_vrnet_matches_reference_module(emul, golden_dir, k):
    from audiolab_amd.vrnet import WIDTHS, VRNet, random_state_dict
    z = np.load(os.path.join(golden_dir, "vrnet.npz"))
    variant, n_fft, seed, aggr, x, want = load_case(z, k)
    net = VRNet(n_fft, random_state_dict(WIDTHS[variant], seed=seed), variant=variant, ctx=emul)
    got = net.forward(torch.from_numpy(x), aggr).numpy()
    assert got.shape == want.shape
    err = float(np.max(np.abs(got - want)))
    assert err < 1e-4 * max(1.0, float(np.max(np.abs(want)))), f"max |delta| {err:.3e}"


def test_vrnet_rejects_bad_state(emul):
    from audiolab_amd._lib import AlsepError
    from audiolab_amd.vrnet import WIDTHS, VRNet, random_state_dict
    sd = random_state_dict(WIDTHS["nets"], seed=1)
    bad = dict(sd)
    del bad["stg2_bridge.conv.0.weight"]
    with pytest.raises(AlsepError):
        VRNet(128, bad, variant="nets", ctx=emul)
    with pytest.raises(AlsepError):
        VRNet(128, sd, variant="nets_61968KB", ctx=emul)        # widths do not match the tensors
    net = VRNet(128, sd, variant="nets", ctx=emul)
    with pytest.raises(AlsepError):
        net.forward_nhwc(torch.zeros((1, 40, 16, 2)))           # fewer bins than n_fft / 2 + 1
