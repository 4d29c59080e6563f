"""CPU (-m "not gpu"): the unchanged TFC-TDF U-Net kernel sources (MFMA emulated lane by lane)
against the plain torch fp32 reference forward (oracle/tdfnet_oracle.py) at small sizes."""
import numpy as np
import pytest
import torch

from oracle import tdfnet_oracle


def make(cfg_kwargs, seed=0):
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNetConfig
    cfg = TDFNetConfig(**cfg_kwargs)
    sd = synthetic_state_dict(cfg, seed=seed, calib_frames=cfg.dim_t)
    return cfg, sd


def run_case(emul, cfg_kwargs, dtype, batch, tol, denoise=False):
    from audiolab_amd.tdfnet import TDFNet
    cfg, sd = make(cfg_kwargs)
    net = TDFNet(cfg, sd, ctx=emul, dtype=dtype, max_batch=2)
    g = torch.Generator().manual_seed(7)
    x_ref = torch.randn((batch, 4, cfg.dim_f, cfg.dim_t), generator=g) * 4.0
    if dtype == torch.bfloat16:
        x_ref = x_ref.to(torch.bfloat16).float()            # same rounded input for both sides
    want = tdfnet_oracle.forward(sd, x_ref, cfg.num_blocks, cfg.l, cfg.bn)
    if denoise:
        want = 0.5 * want - 0.5 * tdfnet_oracle.forward(sd, -x_ref, cfg.num_blocks, cfg.l, cfg.bn)
    x = x_ref.permute(0, 3, 2, 1).contiguous().to(dtype)    # NHWC [B,T,F,4]
    got = net.forward_nhwc(x, denoise=denoise).float().permute(0, 3, 2, 1)
    scale = float(want.abs().max())
    err = float((got - want).abs().max())
    assert scale > 1e-3, "degenerate reference output"
    assert err < tol * scale, f"max err {err:.3e} vs scale {scale:.3e}"
    return net, x_ref, want


def test_net_f32_small_generic_tiles(emul):
    """g=16: exercises the 16-channel fall-back conv tiles, padded TDF rows (f/bn = 2, 4, 8) and K."""
    run_case(emul, dict(dim_f=64, dim_t=16, n_fft=256, hop=64, num_blocks=5, g=16), torch.float32, 3, 2e-4)


def test_net_f32_g48_main_tiles_and_seam(emul):
    """g=48: the main fp32 conv tile (KC=16, BN=48), odd tile counts (F=96 -> TW=32), denoise
    average, and the ORT-shaped run() seam in the reference layout."""
    kw = dict(dim_f=96, dim_t=8, n_fft=256, hop=64, num_blocks=3, g=48, bn=4)
    net, x_ref, _ = run_case(emul, kw, torch.float32, 2, 2e-4, denoise=True)
    cfg, sd = make(kw)
    plain = tdfnet_oracle.forward(sd, x_ref, cfg.num_blocks, cfg.l, cfg.bn)
    pred = net.run(None, {"input": x_ref})[0]              # reference layout in/out (patch_separate.py:52)
    assert pred.shape == x_ref.shape
    assert float((pred - plain).abs().max()) < 2e-4 * float(plain.abs().max())


def test_net_bf16_g48(emul):
    """bf16 storage + bf16 MFMA (KC=48, BN=48 tile with the zero-padded 14th k-step).  The bound
    is bf16 rounding (2^-8 per stored activation) amplified by the random-weight network."""
    from audiolab_amd.tdfnet import TDFNet
    for nb, tol in ((1, 2e-2), (3, 6e-2)):
        cfg, sd = make(dict(dim_f=64, dim_t=16, n_fft=256, hop=64, num_blocks=nb, g=48))
        net = TDFNet(cfg, sd, ctx=emul, dtype=torch.bfloat16, max_batch=2)
        g = torch.Generator().manual_seed(7)
        x_ref = (torch.randn((2, 4, cfg.dim_f, cfg.dim_t), generator=g) * 4.0).to(torch.bfloat16)
        want = tdfnet_oracle.forward(sd, x_ref.float(), cfg.num_blocks, cfg.l, cfg.bn)
        got = net.forward_nhwc(x_ref.permute(0, 3, 2, 1).contiguous()).float().permute(0, 3, 2, 1)
        rel = float((got - want).norm() / want.norm())
        assert rel < tol, f"num_blocks={nb}: rel L2 {rel:.3e}"


def test_net_bf16_persistent_conv_two_chunks(emul, monkeypatch):
    """Persistent conv variants: register-weight kernel with two input chunks (opt-in) and the software-
    pipelined kernel for Cout = 96 / 144 (level 1: 9*4*4 = 144 tiles >= 128; level 2 falls back)."""
    import subprocess, sys, os
    code = (
        "import os, sys, torch; sys.path.insert(0, %r); os.environ['ALSEP_CONV_REGW']='2'; os.environ['ALSEP_CONV_PIPE']='2'; os.environ['ALSEP_CONV_BIG']=sys.argv[1]; os.environ['ALSEP_CONV_MNY']=sys.argv[2]; os.environ['ALSEP_CONV_MQ']=sys.argv[3]; os.environ['ALSEP_CONV_M0']=sys.argv[4]\n"
        "from audiolab_amd import _lib\n"
        "_lib._LIB=_lib.bind(%r); _lib.DEVICE_TYPE='cpu'\n"
        "from audiolab_amd.synth import synthetic_state_dict\n"
        "from audiolab_amd.tdfnet import TDFNet, TDFNetConfig\n"
        "from oracle import tdfnet_oracle\n"
        "cfg=TDFNetConfig(dim_f=256, dim_t=16, n_fft=512, hop=64, num_blocks=5, g=48)\n"
        "sd=synthetic_state_dict(cfg, calib_frames=16)\n"
        "net=TDFNet(cfg, sd, ctx=_lib.Context('cpu'), dtype=torch.bfloat16, max_batch=9)\n"
        "x=(torch.randn((9,4,256,16), generator=torch.Generator().manual_seed(3))*4).to(torch.bfloat16)\n"
        "want=tdfnet_oracle.forward(sd, x.float(), 5, 3, 8)\n"
        "got=net.forward_nhwc(x.permute(0,3,2,1).contiguous()).float().permute(0,3,2,1)\n"
        "rel=float((got-want).norm()/want.norm()); print('rel', rel); assert rel < 8e-2\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_emul", "libalsep_emul.so"))
    # 0: pipelined 4-wave kernel at level 1; 2: big-tile 8-wave kernel at level 1; 2 + MNY: its merged form
    # ... 2 + MQ: the fully double-buffered level-1 kernel
    # the last: + the LDS-resident-weight level-0 kernel (48-pixel tiles: dim_f 256 is not a multiple of 48 -> it needs its own shape below)
    for big, mny, mq, m0 in (("0", "0", "0", "0"), ("2", "3", "0", "0"), ("2", "0", "1", "2")):
        r = subprocess.run([sys.executable, "-c", code, big, mny, mq, m0], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr


def test_net_bf16_wide_tdf_matches_narrow(emul, tmp_path):
    """Wide-tile TDF kernel (register-resident weight fragments, 3-stage activation ring) against the
    128-row kernel: same k order, so the outputs must be bit-identical; and against the oracle.
    dim_f=768, bn=4: first linear 768 -> 192 (M = 192: 4 waves), second 192 -> 768 (M = 768: 4 or, forced, 8 waves)."""
    import subprocess, sys, os
    code = (
        "import os, sys, torch; sys.path.insert(0, %r); os.environ['ALSEP_TDF_WIDE']=sys.argv[1]\n"
        "from audiolab_amd import _lib\n"
        "_lib._LIB=_lib.bind(%r); _lib.DEVICE_TYPE='cpu'\n"
        "from audiolab_amd.synth import synthetic_state_dict\n"
        "from audiolab_amd.tdfnet import TDFNet, TDFNetConfig\n"
        "from oracle import tdfnet_oracle\n"
        "cfg=TDFNetConfig(dim_f=768, dim_t=8, n_fft=2048, hop=64, num_blocks=1, g=48, bn=4)\n"
        "sd=synthetic_state_dict(cfg, calib_frames=8)\n"
        "net=TDFNet(cfg, sd, ctx=_lib.Context('cpu'), dtype=torch.bfloat16, max_batch=1)\n"
        "x=(torch.randn((1,4,768,8), generator=torch.Generator().manual_seed(3))*4).to(torch.bfloat16)\n"
        "want=tdfnet_oracle.forward(sd, x.float(), 1, 3, 4)\n"
        "got=net.forward_nhwc(x.permute(0,3,2,1).contiguous()).float().permute(0,3,2,1)\n"
        "rel=float((got-want).norm()/want.norm()); print('rel', rel); assert rel < 3e-2\n"
        "torch.save(got, sys.argv[2])\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_emul", "libalsep_emul.so"))
    outs = {}
    for mode in ("0", "1", "8"):
        path = str(tmp_path / f"out{mode}.pt")
        r = subprocess.run([sys.executable, "-c", code, mode, path], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[mode] = torch.load(path)
    assert torch.equal(outs["0"], outs["1"]) and torch.equal(outs["0"], outs["8"])


def test_net_rejects_bad_inputs(emul):
    from audiolab_amd._lib import AlsepError
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    cfg, sd = make(dict(dim_f=64, dim_t=16, n_fft=256, hop=64, num_blocks=3, g=16))
    net = TDFNet(cfg, sd, ctx=emul)
    with pytest.raises(AlsepError):
        net.forward_nhwc(torch.zeros((1, 16, 64, 3)))                  # wrong channel count
    with pytest.raises(AlsepError):
        net.forward_nhwc(torch.zeros((1, 16, 64, 4), dtype=torch.bfloat16))   # wrong dtype
    bad = dict(sd)
    del bad["ds.0.0.weight"]
    with pytest.raises(AlsepError):
        TDFNet(cfg, bad, ctx=emul)
    with pytest.raises(AlsepError):                                     # dim_f not divisible by 2^n
        TDFNet(TDFNetConfig(dim_f=36, dim_t=16, n_fft=256, hop=64, num_blocks=5, g=16), sd, ctx=emul)


def test_net_f16_g48_vs_storage_oracle(emul):
    """IEEE-half storage + f16 MFMA (tdfnet_f16.hip: the bf16 kernel sources compiled with _Float16): against the oracle in its
    half-precision storage mode the only disagreement is flipped roundings; against the fp32 oracle the error is ~8x below bf16's."""
    from audiolab_amd.tdfnet import TDFNet
    for nb in (1, 3):
        cfg, sd = make(dict(dim_f=64, dim_t=16, n_fft=256, hop=64, num_blocks=nb, g=48))
        net = TDFNet(cfg, sd, ctx=emul, dtype=torch.float16, max_batch=2)
        x = (torch.randn((2, 4, cfg.dim_f, cfg.dim_t), generator=torch.Generator().manual_seed(7)) * 4.0).to(torch.float16).float()
        w32 = tdfnet_oracle.forward(sd, x, cfg.num_blocks, cfg.l, cfg.bn)
        wst = tdfnet_oracle.forward(sd, x, cfg.num_blocks, cfg.l, cfg.bn, storage=torch.float16)
        got = net.forward_nhwc(x.permute(0, 3, 2, 1).contiguous().to(torch.float16)).float().permute(0, 3, 2, 1)
        r = lambda a, b: float((a - b).norm() / b.norm())
        print(f"f16 num_blocks={nb}: vs storage oracle {r(got, wst):.3e}, vs fp32 oracle {r(got, w32):.3e}")
        assert r(got, wst) < 1e-3 and r(got, w32) < 3e-3


def test_level0_lds_weight_kernel_bit_identical_on_emulation(emul, tmp_path):
    """conv3x3_bf16_m0_kernel (8 x 48-pixel tiles, weights resident in LDS, double-buffered patches) against the register-weight kernel on
    the same one-block network: same k order, so the outputs must be bit-identical (the GPU twin: tests/test_gpu_conv_variants.py)"""
    import subprocess, sys, os
    code = (
        "import os, sys, torch, numpy as np; sys.path.insert(0, %r); os.environ['ALSEP_CONV_M0']=sys.argv[1]\n"
        "from audiolab_amd import _lib\n"
        "_lib._LIB=_lib.bind(%r); _lib.DEVICE_TYPE='cpu'\n"
        "from audiolab_amd.synth import synthetic_state_dict\n"
        "from audiolab_amd.tdfnet import TDFNet, TDFNetConfig\n"
        "cfg=TDFNetConfig(dim_f=192, dim_t=16, n_fft=512, hop=64, num_blocks=1, g=48)\n"
        "sd=synthetic_state_dict(cfg, calib_frames=16)\n"
        "ctx=_lib.Context('cpu')\n"
        "net=TDFNet(cfg, sd, ctx=ctx, dtype=torch.bfloat16, max_batch=3)\n"
        "x=(torch.randn((3,16,192,4), generator=torch.Generator().manual_seed(3))*4).to(torch.bfloat16)\n"
        "y=net.forward_nhwc(x).float().numpy()\n"
        "print('m0 launches', ctx.launch_count('conv3x3_bf16_m0_kernel'))\n"
        "np.save(sys.argv[2], y)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.join(os.path.dirname(os.path.abspath(__file__)), "cpu_emul", "libalsep_emul.so"))
    outs = []
    for m0 in ("0", "2"):
        path = str(tmp_path / f"y{m0}.npy")
        r = subprocess.run([sys.executable, "-c", code, m0, path], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("m0 launches 3" in r.stdout) == (m0 == "2"), r.stdout
        outs.append(np.load(path))
    assert np.isfinite(outs[0]).all() and np.abs(outs[0]).max() > 1e-3
    assert np.array_equal(outs[0], outs[1]), f"max diff {np.abs(outs[0] - outs[1]).max()}"
