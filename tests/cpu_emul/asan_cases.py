"""TEST-ONLY: drive the AddressSanitizer build of the emulated kernels (libalsep_emul_asan.so) over small
cases of every kernel family.  Run through tests/cpu_emul/run_asan.sh (needs the ASan runtime preloaded)."""
import ctypes as C
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ALSEP_CONV_REGW", "2")      # exercise both persistent conv variants
os.environ.setdefault("ALSEP_CONV_PIPE", "2")
from audiolab_amd import _lib  # noqa: E402

_lib._LIB = _lib.bind(os.path.join(ROOT, "tests", "cpu_emul", "libalsep_emul_asan.so"))
_lib.DEVICE_TYPE = "cpu"
ctx = _lib.Context("cpu")
from audiolab_amd import ensemble  # noqa: E402
from audiolab_amd.mdx import ConvTDFNetTrim, OlaRunner, Predictor, StftPlan  # noqa: E402
from audiolab_amd.synth import synth_mix, synthetic_state_dict  # noqa: E402
from audiolab_amd.tdfnet import TDFNet, TDFNetConfig  # noqa: E402

for (n_fft, hop, dta, dim_f) in [(256, 64, 4, 96), (384, 64, 4, 160), (480, 64, 4, 192), (256, 64, 5, 129)]:
    net = ConvTDFNetTrim("cpu", "Conv-TDF", "vocals", 11, dim_f, dta, n_fft, hop=hop, ctx=ctx)
    x = torch.randn(2, 2, net.chunk_size)
    net.istft(net.stft(x))
plan = StftPlan(ctx, 2048, 1024, 640, 12)                  # register-ring iSTFT
sp = plan.stft_strided(torch.randn(2, 2, plan.chunk_size), plan.chunk_size, 2 * plan.chunk_size, 2, torch.bfloat16, _lib.LAYOUT_NHWC)
out = ctx.empty((2, 2, plan.chunk_size))
plan.istft_strided(sp, _lib.LAYOUT_NHWC, out, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size, 3 * plan.chunk_size)
print("stft/istft ok")
for kw, dt, b in [(dict(dim_f=64, dim_t=16, n_fft=256, hop=64, num_blocks=5, g=16), torch.float32, 3),
                  (dict(dim_f=96, dim_t=8, n_fft=256, hop=64, num_blocks=3, g=48, bn=4), torch.float32, 2),
                  (dict(dim_f=256, dim_t=16, n_fft=512, hop=64, num_blocks=5, g=48), torch.bfloat16, 5),
                  (dict(dim_f=64, dim_t=16, n_fft=256, hop=64, num_blocks=3, g=32), torch.bfloat16, 3)]:
    cfg = TDFNetConfig(**kw)
    net = TDFNet(cfg, synthetic_state_dict(cfg, calib_frames=16), ctx=ctx, dtype=dt, max_batch=3)
    y = net.forward_nhwc(torch.randn(b, cfg.dim_t, cfg.dim_f, 4).to(dt), denoise=True)
    assert torch.isfinite(y.float()).all()
    print("net ok", kw, dt)
cfg = TDFNetConfig(dim_f=96, dim_t=32, n_fft=256, hop=64, num_blocks=3, g=16)
net = TDFNet(cfg, synthetic_state_dict(cfg), ctx=ctx)
args = types.SimpleNamespace(margin=441, chunks=1, denoise=True, dim_f=96, dim_t=5, n_fft=256)
print("demix ok", Predictor(args, net, ctx=ctx, hop=64, max_batch=3).demix(torch.from_numpy(synth_mix(50000))).shape)
print("ola ok", OlaRunner(net, overlap=0.75, max_batch=4).demix(torch.from_numpy(synth_mix(7000))).shape)
a, b = torch.randn(2, 50007) * 0.1, torch.randn(2, 50007) * 0.1
ensemble.blend_tracks(ctx, [a, b[:, :40001]], [1.0, 2.0])
ensemble.debleed(ctx, a + b, a, b, 44100, 0.2)
print("ensemble ok")
# three-pass STFT / iSTFT (fft_r16.h): production sizes, small dim_t (edge + interior frames), narrow / full / Nyquist bands
for (n_fft, dim_f, dim_t) in [(6144, 3072, 8), (6144, 3073, 7), (6144, 500, 7), (4096, 2049, 6), (7680, 3072, 10), (7680, 3841, 9)]:
    plan = StftPlan(ctx, n_fft, 1024, dim_f, dim_t)
    x = torch.randn(2, 2, plan.chunk_size)
    for dt in (torch.float32, torch.bfloat16):
        sp = plan.stft_strided(x, plan.chunk_size, 2 * plan.chunk_size, 2, dt, _lib.LAYOUT_NHWC)
        out = ctx.empty((2, 2, plan.chunk_size))
        plan.istft_strided(sp, _lib.LAYOUT_NHWC, out, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size, 3 * plan.chunk_size)
        assert torch.isfinite(out).all()
    ref = plan.stft_strided(x, plan.chunk_size, 2 * plan.chunk_size, 2, torch.float32, _lib.LAYOUT_REF)
    plan.istft_strided(ref, _lib.LAYOUT_REF, out, plan.chunk_size, 2 * plan.chunk_size, 0, plan.chunk_size, 3 * plan.chunk_size)
print("three-pass stft/istft ok")
# streaming ds / us kernels of levels 0<->1<->2<->3 (Fp % 64 == 0 at every level) and the two-workgroup register-weight conv
cfg = TDFNetConfig(dim_f=512, dim_t=8, n_fft=1024, hop=64, num_blocks=7, g=48)
net = TDFNet(cfg, synthetic_state_dict(cfg, calib_frames=8), ctx=ctx, dtype=torch.bfloat16, max_batch=2)
y = net.forward_nhwc(torch.randn(2, cfg.dim_t, cfg.dim_f, 4).to(torch.bfloat16))
assert torch.isfinite(y.float()).all()
print("stream ds/us ok")
# VR networks
from audiolab_amd.vrnet import WIDTHS, VRNet, VRNetNew, random_state_dict, random_state_dict_new, vr_inference  # noqa: E402
vr = VRNet(64, random_state_dict(WIDTHS["nets"], seed=1), variant="nets", ctx=ctx)
assert torch.isfinite(vr.forward(torch.rand(1, 2, 33, 32), {"split_bin": 10, "value": 0.1})).all()   # frames: a multiple of 16
vr.offset = 8
pred, _, _ = vr_inference(vr, torch.randn(2, 33, 40, dtype=torch.complex64), None, window_size=32, tta=True, max_batch=2)
assert torch.isfinite(pred).all()
vn = VRNetNew(64, random_state_dict_new(64, 16, 64, seed=2), nout=16, nout_lstm=64, ctx=ctx)
assert torch.isfinite(vn.forward(torch.rand(1, 2, 33, 32))).all()
print("vr ok")
