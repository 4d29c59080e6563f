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


@pytest.mark.parametrize("k", [1])       # all three cases run on the GPU (tests/test_gpu_parity.py); one keeps the CPU suite short
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


@pytest.mark.parametrize("tag,tta,aggr", [("plain", False, None)])      # the TTA case runs on the GPU
def test_vr_runner_matches_reference_inference(emul, golden_dir, tag, tta, aggr):
    """utils.py:25-100 ``inference`` (normalise by the peak, pad, window every roi_size frames, predict, concat, TTA) run by
    the reference on its own net with offset 8 / window 48 -- against vr_inference on the HIP network."""
    from audiolab_amd.vrnet import WIDTHS, VRNet, random_state_dict, vr_inference
    z = np.load(os.path.join(golden_dir, "vrnet.npz"))
    net = VRNet(64, random_state_dict(WIDTHS["nets"], seed=21), variant="nets", ctx=emul)
    net.offset = 8
    pred, mag, phase = vr_inference(net, torch.from_numpy(z["inf_x"]), aggr, window_size=48, tta=tta, max_batch=3)
    want = z[f"inf_{tag}_pred"]
    assert pred.shape == want.shape
    assert float(np.max(np.abs(pred.numpy() - want))) < 1e-4 * max(1.0, float(np.max(np.abs(want))))
    assert float(np.max(np.abs(mag.numpy() - z["inf_mag"]))) < 1e-6
    assert float(np.max(np.abs(phase.numpy() - z["inf_phase"]))) < 1e-5


@pytest.mark.parametrize("k", [0])        # the second case runs on the GPU
def test_vrnet_new_matches_reference_module(emul, golden_dir, k):
    """nets_new.py ``CascadedNet`` (BaseNet + bidirectional LSTM branch, per-axis dilations): ``predict`` of the reference
    module on seeded weights against VRNetNew."""
    from audiolab_amd.vrnet import VRNetNew, random_state_dict_new
    z = np.load(os.path.join(golden_dir, "vrnet.npz"))
    n_fft, nout, nout_lstm, frames, seed = (int(v) for v in z[f"n{k}_cfg"])
    net = VRNetNew(n_fft, random_state_dict_new(n_fft, nout, nout_lstm, seed=seed), nout=nout, nout_lstm=nout_lstm, ctx=emul)
    got = net.forward(torch.from_numpy(z[f"n{k}_x"])).numpy()
    want = z[f"n{k}_y"]
    assert got.shape == want.shape
    err = float(np.max(np.abs(got - want)))
    assert err < 1e-4 * max(1.0, float(np.max(np.abs(want)))), f"max |delta| {err:.3e}"
