import sys, types, numpy as np, torch
sys.path.insert(0, '/root/repo')
from audiolab_amd import _lib
if len(sys.argv) > 1 and sys.argv[1] == 'emul':
    lib = _lib.bind('/root/repo/tests/cpu_emul/libalsep_emul.so')
    _lib._LIB = lib; _lib.DEVICE_TYPE = 'cpu'; _lib._DEFAULT_CTX = {}
    ctx = _lib.Context('cpu'); dev = 'cpu'
else:
    ctx = _lib.Context('cuda:0'); dev = 'cuda'
from audiolab_amd.tdfnet import TDFNet, TDFNetConfig
from audiolab_amd.synth import synthetic_state_dict, synth_mix
from audiolab_amd.mdx import Predictor
cfg = TDFNetConfig(dim_f=256, dim_t=32, n_fft=512, hop=128, num_blocks=3, g=48)
sd = synthetic_state_dict(cfg, seed=0, calib_frames=32)
args = types.SimpleNamespace(margin=2205, chunks=0, denoise=False, dim_f=cfg.dim_f, dim_t=5, n_fft=cfg.n_fft)
mix = torch.from_numpy(synth_mix(12000)).to(dev)
net = TDFNet(cfg, sd, ctx=ctx, dtype=torch.float32)
for sc in (1.0, 3e5, 1e9):
    x = torch.randn(1, 4, cfg.dim_f, cfg.dim_t, device=dev) * sc
    y = net(x)
    print('scale', sc, 'net out finite', bool(torch.isfinite(y).all()), 'absmax', float(y.abs().max()), 'nans', int(torch.isnan(y).sum()))
    try:
        o = Predictor(args, net, ctx=ctx, hop=cfg.hop).demix(mix * sc)
        print('   demix finite', bool(torch.isfinite(o).all()), float(o.abs().max()))
    except Exception as e:
        print('   demix raised', str(e)[:80])
