"""STFT + first network layer as ONE kernel (alsep_net_forward_pcm; fft_r16.h FUSE) against the two separate kernels, on the emulated
kernels (-m "not gpu") and on the GPU (-m gpu): the fused path must reproduce stft -> [zero_low_bins] -> network BIT FOR BIT (same
rounding points, same fma chains), for both half-precision storage types, interior and reflect-padded frames, ragged dim_f."""
import numpy as np
import pytest
import torch

from tests.conftest import on


@pytest.mark.parametrize("dtype,n_fft,dim_f,zl", [(torch.bfloat16, 4096, 96, 0), (torch.float16, 4096, 64, 3), (torch.bfloat16, 6144, 112, 3),
                                                  (torch.float16, 7680, 80, 0)])
def test_fused_front_end_is_bit_identical(dev, dtype, n_fft, dim_f, zl):
    from audiolab_amd import _lib
    from audiolab_amd.mdx import StftPlan
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    from oracle.toy import synth_mix
    if dev.device.type == "cpu" and n_fft != 4096:
        pytest.skip("emulated suite keeps the 4096-point cases (the 6144 / 7680 kernels run on the GPU)")
    cfg = TDFNetConfig(dim_f=dim_f, dim_t=8, n_fft=n_fft, hop=1024, num_blocks=1, g=48, bn=8)
    sd = synthetic_state_dict(cfg, seed=2, calib="noise")
    net = TDFNet(cfg, sd, ctx=dev, dtype=dtype, max_batch=2)
    plan = StftPlan(dev, cfg.n_fft, cfg.hop, cfg.dim_f, cfg.dim_t)
    chunk, step, nb = plan.chunk_size, 3000, 3
    total = (nb - 1) * step + chunk + 17
    pcm = on(dev, synth_mix(total, seed=4) * 3.0)
    dev.launch_counts_reset()
    got = net.forward_pcm(plan, pcm, total, step, nb, pcm_offset=5, zero_low_bins=zl)
    assert got is not None and dev.launch_count("stft_first_conv_kernel") == 2 and dev.launch_count("first_conv_kernel") == 0   # max_batch 2: 2 + 1
    spek = plan.stft_strided(pcm, total, step, nb, dtype, _lib.LAYOUT_NHWC, pcm_offset=5)
    if zl:
        dev.check(dev.lib.alsep_zero_low_bins(dev.handle, _lib.ptr(spek), _lib.dtype_code(dtype), _lib.LAYOUT_NHWC, nb, plan.dim_f, plan.dim_t, zl),
                  "alsep_zero_low_bins")
    want = net.forward_nhwc(spek)
    assert got.shape == want.shape == (nb, cfg.dim_t, cfg.dim_f, 4)
    assert float(want.float().abs().max()) > 1e-3
    assert torch.equal(got.cpu(), want.cpu())


def test_fused_front_end_declines_what_it_has_no_kernel_for(dev):
    """float32 networks and geometries without a three-pass kernel: forward_pcm returns None (the runners then take the two-step path)"""
    from audiolab_amd.mdx import StftPlan
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    cfg = TDFNetConfig(dim_f=64, dim_t=8, n_fft=256, hop=64, num_blocks=1, g=48, bn=8)
    sd = synthetic_state_dict(cfg, seed=2, calib="noise")
    pcm = on(dev, np.zeros((2, 4000), np.float32))
    for dt in (torch.float32, torch.bfloat16):
        net = TDFNet(cfg, sd, ctx=dev, dtype=dt, max_batch=2)
        plan = StftPlan(dev, cfg.n_fft, cfg.hop, cfg.dim_f, cfg.dim_t)
        assert net.forward_pcm(plan, pcm, 4000, 100, 2) is None


@pytest.mark.gpu
def test_bench_geometry_runner_uses_the_fused_kernel(gpu_ctx):
    """configs[1] geometry through the production runner: the fused kernel runs (no separate STFT / first conv launch) and the stems equal
    the two-step path's bit for bit"""
    import types
    from audiolab_amd.mdx import Predictor
    from audiolab_amd.synth import synth_mix, synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    cfg = TDFNetConfig()
    net = TDFNet(cfg, synthetic_state_dict(cfg, seed=0), ctx=gpu_ctx, dtype=torch.bfloat16, max_batch=8)
    args = types.SimpleNamespace(margin=44100, chunks=0, denoise=False, dim_f=cfg.dim_f, dim_t=8, n_fft=cfg.n_fft)
    mix = torch.from_numpy(synth_mix(600000)).cuda()
    pred = Predictor(args, net, ctx=gpu_ctx)
    gpu_ctx.launch_counts_reset()
    fused = pred.demix(mix)
    assert gpu_ctx.launch_count("stft_first_conv_kernel") == 1 and gpu_ctx.launch_count("stft_r16_kernel") == 0
    assert gpu_ctx.launch_count("first_conv_kernel") == 0
    net.forward_pcm = lambda *a, **k: None                   # the two-step path
    plain = pred.demix(mix)
    assert gpu_ctx.launch_count("stft_r16_kernel") == 1 and gpu_ctx.launch_count("first_conv_kernel") == 1
    assert torch.equal(fused, plain)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_fused_front_end_7680_persistent_workgroups_bit_identical(gpu_ctx, dtype):
    """n_fft 7680 runs the fused front end with PERSISTENT workgroups (fft_r16.h PERSIST: two per CU, each walking several frames with the
    frame buffer reused behind a barrier): 1 664 frames over 512 workgroups = 3-4 frames each, ragged dim_f, reflect-padded edges and
    zeroed low bins -- bit for bit the two-step path"""
    from audiolab_amd import _lib
    from audiolab_amd.mdx import StftPlan
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    from oracle.toy import synth_mix
    cfg = TDFNetConfig(dim_f=200, dim_t=64, n_fft=7680, hop=1024, num_blocks=1, g=48, bn=8)
    sd = synthetic_state_dict(cfg, seed=3, calib="noise")
    net = TDFNet(cfg, sd, ctx=gpu_ctx, dtype=dtype, max_batch=26)
    plan = StftPlan(gpu_ctx, cfg.n_fft, cfg.hop, cfg.dim_f, cfg.dim_t)
    chunk, step, nb = plan.chunk_size, 20000, 26
    total = (nb - 1) * step + chunk + 9
    pcm = torch.from_numpy(synth_mix(total, seed=5) * 2.0).cuda()
    gpu_ctx.launch_counts_reset()
    got = net.forward_pcm(plan, pcm, total, step, nb, pcm_offset=3, zero_low_bins=2)
    assert got is not None and gpu_ctx.launch_count("stft_first_conv_kernel") == 1
    spek = plan.stft_strided(pcm, total, step, nb, dtype, _lib.LAYOUT_NHWC, pcm_offset=3)
    gpu_ctx.check(gpu_ctx.lib.alsep_zero_low_bins(gpu_ctx.handle, _lib.ptr(spek), _lib.dtype_code(dtype), _lib.LAYOUT_NHWC, nb, plan.dim_f, plan.dim_t, 2),
                  "alsep_zero_low_bins")
    want = net.forward_nhwc(spek)
    assert float(want.float().abs().max()) > 1e-3
    assert torch.equal(got.cpu(), want.cpu())


def test_fused_front_end_7680_persistent_loop_on_the_emulation(dev):
    """the frame loop of the persistent 7680 kernel on the emulated kernels: 560 frames over the 512 workgroups a 256-CU device takes, so 48
    workgroups walk two frames (frame buffer reused behind the barrier) -- bit for bit the two-step path"""
    from audiolab_amd import _lib
    from audiolab_amd.mdx import StftPlan
    from audiolab_amd.synth import synthetic_state_dict
    from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
    from oracle.toy import synth_mix
    if dev.device.type != "cpu":
        pytest.skip("the GPU runs test_fused_front_end_7680_persistent_workgroups_bit_identical")
    cfg = TDFNetConfig(dim_f=80, dim_t=8, n_fft=7680, hop=1024, num_blocks=1, g=48, bn=8)
    net = TDFNet(cfg, synthetic_state_dict(cfg, seed=2, calib="noise"), ctx=dev, dtype=torch.float16, max_batch=70)
    plan = StftPlan(dev, cfg.n_fft, cfg.hop, cfg.dim_f, cfg.dim_t)
    chunk, step, nb = plan.chunk_size, 1500, 70
    total = (nb - 1) * step + chunk + 17
    pcm = on(dev, synth_mix(total, seed=4) * 3.0)
    got = net.forward_pcm(plan, pcm, total, step, nb, pcm_offset=5, zero_low_bins=1)
    assert got is not None
    spek = plan.stft_strided(pcm, total, step, nb, torch.float16, _lib.LAYOUT_NHWC, pcm_offset=5)
    dev.check(dev.lib.alsep_zero_low_bins(dev.handle, _lib.ptr(spek), _lib.dtype_code(torch.float16), _lib.LAYOUT_NHWC, nb, plan.dim_f, plan.dim_t, 1),
              "alsep_zero_low_bins")
    want = net.forward_nhwc(spek)
    assert float(want.float().abs().max()) > 1e-3
    assert torch.equal(got.cpu(), want.cpu())
