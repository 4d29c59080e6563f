"""CPU (-m "not gpu"): the N>1 path -- window sharding + one all-gather of stem segments -- with
world_size 2 over gloo, kernels emulated on the CPU, checked against the reference's demix output."""
import os
import socket
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, emul_so, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audiolab_amd import _lib
    from audiolab_amd.mdx import Predictor
    from oracle.toy import synth_mix, toy_net
    _lib._LIB = _lib.bind(emul_so)          # test-only: emulated kernels on the CPU
    _lib.DEVICE_TYPE = "cpu"
    ctx = _lib.Context("cpu")

    class Seam:
        def run(self, _n, feed):
            return [torch.from_numpy(toy_net(feed["input"].numpy()))]
    z = np.load(os.path.join(ROOT, "tests", "golden", "demix.npz"))
    n, chunks, margin, denoise = (int(v) for v in z["small_a_cfg"])
    n_fft, hop, dta, dim_f = (int(v) for v in z["small_geom"])
    args = types.SimpleNamespace(margin=margin, chunks=chunks, denoise=bool(denoise), dim_f=dim_f, dim_t=dta, n_fft=n_fft)
    pred = Predictor(args, Seam(), ctx=ctx, hop=hop, max_batch=3, sharded=True)
    out = pred.demix(torch.from_numpy(synth_mix(n, seed=300 + n + chunks))).numpy()
    err = float(np.max(np.abs(out - z["small_a_out"])))
    deferred = pred.demix(torch.from_numpy(synth_mix(n, seed=300 + n + chunks)), defer=True)   # async all-gather path
    err = max(err, float(np.max(np.abs(deferred().numpy() - z["small_a_out"]))))
    res = torch.tensor([err])
    dist.all_reduce(res, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(out_path, np.array([float(res[0])]))
    dist.barrier()
    dist.destroy_process_group()


def test_window_ranges_cover_everything():
    from audiolab_amd.dist import sample_range, window_range
    for n_win in (0, 1, 5, 8, 53):
        for world in (1, 2, 3, 8):
            r = [window_range(n_win, world, k) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == n_win
            assert all(r[i][1] == r[i + 1][0] for i in range(world - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1
            s = [sample_range(n_win, 100, max(n_win * 100 - 37, 0), world, k) for k in range(world)]
            assert s[0][0] == 0 and s[-1][1] == max(n_win * 100 - 37, 0)


def test_sharded_demix_world2_gloo(emul_lib_path, tmp_path):
    out_path = str(tmp_path / "err.npy")
    mp.spawn(_worker, args=(2, _free_port(), emul_lib_path, out_path), nprocs=2, join=True)
    err = float(np.load(out_path)[0])
    assert err < 1e-5


def _worker_ola_demucs(rank, world, port, emul_so, out_path):
    """world-2 over gloo: (1) Hann overlap-add MDX runner at overlap 0.75 with the chunks sharded and the seam sums exchanged;
    (2) the HTDemucs runner with its (shift, segment) units sharded.  Both against the single-process oracle."""
    import dataclasses
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audiolab_amd import _lib
    _lib._LIB = _lib.bind(emul_so)
    _lib.DEVICE_TYPE = "cpu"
    ctx = _lib.Context("cpu")
    from audiolab_amd.htdemucs import DemucsRunner, HTDemucs, HTDemucsConfig
    from audiolab_amd.mdx import OlaRunner
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    from oracle import htdemucs_oracle as ho
    from oracle import mdx_oracle as mo
    from oracle import tdfnet_oracle
    from oracle.toy import synth_mix
    errs = []
    cfg = TDFNetConfig(dim_f=64, dim_t=32, n_fft=256, hop=64, num_blocks=3, g=16)
    sd = synthetic_state_dict(cfg, seed=3)
    net = TDFNet(cfg, sd, ctx=ctx, dtype=torch.float32, max_batch=2)
    mix = synth_mix(4000, seed=31)
    got = OlaRunner(net, ctx=ctx, overlap=0.75, compensate=1.02, max_batch=2, sharded=True).demix(torch.from_numpy(mix)).numpy()
    g = mo.MDXGeometry(cfg.dim_f, cfg.dim_t, cfg.n_fft, cfg.hop)

    def run(spek):
        return tdfnet_oracle.forward(sd, torch.from_numpy(np.ascontiguousarray(spek, dtype=np.float32)), cfg.num_blocks, cfg.l, cfg.bn).numpy()
    want = mo.demix_ola(mix, g, run, overlap=0.75, denoise=False, zero_low_bins=3, compensate=1.02)
    errs.append(float(np.max(np.abs(got - want))))
    # the engine's sequence on a HOST-resident loud programme: each rank uploads only the samples under its chunks, the seam sums are
    # exchanged, the stem segments all-gathered; normalisation 0.9 and the spectral inversion of the second stem as in one process
    from audiolab_amd.engine import Separator
    loud = (mix * (1.4 / np.max(np.abs(mix)))).astype(np.float32)
    eng = Separator(ctx=ctx, use_autocast=False, allow_synthetic=True, roster={"m.onnx": ("Vocals", "Instrumental", cfg)}, max_batch=2,
                    chunker="ola", overlap=0.75, compensate=1.02, sharded=True)
    eng.load_model("m.onnx")
    import hashlib
    sd2 = synthetic_state_dict(cfg, seed=int.from_bytes(hashlib.sha256(b"m.onnx").digest()[:4], "little"))

    def run2(spek):
        return tdfnet_oracle.forward(sd2, torch.from_numpy(np.ascontiguousarray(spek, dtype=np.float32)), cfg.num_blocks, cfg.l, cfg.bn).numpy()
    out2 = eng.separate_array(torch.from_numpy(loud))
    w1, w2 = mo.separate_ola(loud, g, run2, overlap=0.75, compensate=1.02)
    errs[-1] = max(errs[-1], float(np.max(np.abs(out2["Vocals"].numpy() - w1))), float(np.max(np.abs(out2["Instrumental"].numpy() - w2))))
    # a track so short that a rank's chunks reach across the whole range of the next one (and ranks without any chunk)
    tiny = synth_mix(300, seed=5)
    got_t = OlaRunner(net, ctx=ctx, overlap=0.75, compensate=1.0, max_batch=2, sharded=True).demix(torch.from_numpy(tiny)).numpy()
    errs[-1] = max(errs[-1], float(np.max(np.abs(got_t - mo.demix_ola(tiny, g, run, overlap=0.75, zero_low_bins=3, compensate=1.0)))))
    ocfg = ho.HTDemucsConfig(sources=("drums", "bass"), channels=16, nfft=256, depth=2, dconv_comp=4, bottom_channels=32, t_layers=2,
                             t_heads=4, segment_samples=2560, samplerate=4000)
    hsd = ho.synthetic_state_dict(ocfg, 5)
    hnet = HTDemucs(HTDemucsConfig(**dataclasses.asdict(ocfg)), hsd, ctx=ctx)
    hm = torch.randn(2, 5000, generator=torch.Generator().manual_seed(8)) * 0.2
    out = DemucsRunner(hnet, shifts=1, overlap=0.25, seed=0, sharded=True).separate(hm)
    hw = ho.separate(ocfg, hsd, hm, shifts=1, overlap=0.25, seed=0).numpy()
    errs.append(float(max(np.max(np.abs(out[k].numpy() - hw[i])) for i, k in enumerate(ocfg.sources))))
    # two shift passes: a rank's run of units crosses the pass boundary (its span differs per pass, the other rank has none in one of them)
    out = DemucsRunner(hnet, shifts=2, overlap=0.25, seed=3, sharded=True).separate(hm[:, :3100])
    hw = ho.separate(ocfg, hsd, hm[:, :3100], shifts=2, overlap=0.25, seed=3).numpy()
    errs[-1] = max(errs[-1], float(max(np.max(np.abs(out[k].numpy() - hw[i])) for i, k in enumerate(ocfg.sources))))
    res = torch.tensor(errs)
    dist.all_reduce(res, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(out_path, res.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ola_and_demucs_world2_gloo(emul_lib_path, tmp_path):
    out_path = str(tmp_path / "err2.npy")
    mp.spawn(_worker_ola_demucs, args=(2, _free_port(), emul_lib_path, out_path), nprocs=2, join=True)
    errs = np.load(out_path)
    assert errs[0] < 1e-4, f"sharded Hann overlap-add: {errs[0]:.3e}"
    assert errs[1] < 1e-4, f"sharded Demucs runner: {errs[1]:.3e}"


def _worker_world3(rank, world, port, emul_so, out_path):
    """world-3 over gloo (an odd rank count: uneven ranges, a middle rank whose span has neighbours on both sides): the chunked Roformer /
    MDX23C runner with its chunks sharded (span all-gather + seam all-gather), a track with fewer chunks than ranks (ranks without any),
    and the Demucs runner -- each against the single-process oracle."""
    import dataclasses
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from audiolab_amd import _lib
    _lib._LIB = _lib.bind(emul_so)
    _lib.DEVICE_TYPE = "cpu"
    ctx = _lib.Context("cpu")
    from audiolab_amd.htdemucs import DemucsRunner, HTDemucs, HTDemucsConfig
    from audiolab_amd.roformer import Roformer, RoformerConfig, RoformerRunner
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    from oracle import htdemucs_oracle as ho
    from oracle import mdx_oracle as mo
    from oracle import roformer_oracle as ro
    from oracle import tdfnet_oracle
    from oracle.toy import synth_mix
    errs = []
    # (1) Roformer runner: chunk 1200, overlap 4 -> step 300; 7 000 samples -> 30 chunks over 3 ranks; then 500 samples -> 2 chunks (rank 2 has none)
    rcfg = ro.RoformerConfig(kind="bs", dim=32, depth=1, heads=2, dim_head=16, n_fft=256, hop=60, sample_rate=8000, chunk_size=1200, num_overlap=4,
                             freqs_per_bands=(4,) * 16 + (8,) * 8 + (1,), mlp_expansion_factor=2)
    rsd = ro.synthetic_state_dict(rcfg, 4)
    rnet = Roformer(RoformerConfig(**dataclasses.asdict(rcfg)), rsd, ctx=ctx)
    worst = 0.0
    for n in (7000, 500):
        rm = torch.from_numpy(synth_mix(n, seed=60 + n))
        got = RoformerRunner(rnet, ("Vocals",), sharded=True).demix(rm).numpy()
        want = ro.demix_track(rcfg, rsd, rm).numpy()
        worst = max(worst, float(np.max(np.abs(got - want))))
    errs.append(worst)
    # (2) Demucs units over three ranks, two shift passes (the overlap-add runner's span logic is the same code as in the world-2 test)
    ocfg = ho.HTDemucsConfig(sources=("drums", "bass"), channels=16, nfft=256, depth=2, dconv_comp=4, bottom_channels=32, t_layers=2,
                             t_heads=4, segment_samples=2560, samplerate=4000)
    hsd = ho.synthetic_state_dict(ocfg, 5)
    hnet = HTDemucs(HTDemucsConfig(**dataclasses.asdict(ocfg)), hsd, ctx=ctx)
    hm = torch.randn(2, 6000, generator=torch.Generator().manual_seed(9)) * 0.2
    out = DemucsRunner(hnet, shifts=2, overlap=0.25, seed=1, sharded=True).separate(hm)
    hw = ho.separate(ocfg, hsd, hm, shifts=2, overlap=0.25, seed=1).numpy()
    errs.append(float(max(np.max(np.abs(out[k].numpy() - hw[i])) for i, k in enumerate(ocfg.sources))))
    res = torch.tensor(errs)
    dist.all_reduce(res, op=dist.ReduceOp.MAX)
    if rank == 0:
        np.save(out_path, res.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_runners_world3_gloo(emul_lib_path, tmp_path):
    out_path = str(tmp_path / "err3.npy")
    mp.spawn(_worker_world3, args=(3, _free_port(), emul_lib_path, out_path), nprocs=3, join=True)
    errs = np.load(out_path)
    assert errs[0] < 1e-4, f"sharded Roformer runner: {errs[0]:.3e}"
    assert errs[1] < 1e-4, f"sharded Demucs runner (world 3): {errs[1]:.3e}"
