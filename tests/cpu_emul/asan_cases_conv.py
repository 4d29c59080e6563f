"""TEST-ONLY: AddressSanitizer pass over the round-2 conv kernels (level-0 LDS-resident-weight, level-1 double-buffered and merged,
level-2 merged / big-tile) at small shapes where every tile touches a border, on the emulation's ASan build.
    LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tests/cpu_emul/asan_cases_conv.py <mode>
mode: mq (default kernels: m0 + mq + big<3>), mny (merged kernels at c = 96 / 144), big (big-tile kernels)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
mode = sys.argv[1] if len(sys.argv) > 1 else "mq"
os.environ["ALSEP_CONV_BIG"] = "2"                         # no minimum tile count
os.environ["ALSEP_CONV_M0"] = "2" if mode == "mq" else "0"
os.environ["ALSEP_CONV_MQ"] = "1" if mode == "mq" else "0"
os.environ["ALSEP_CONV_MNY"] = "3" if mode == "mny" else "0"
from audiolab_amd import _lib  # noqa: E402

_lib._LIB = _lib.bind(os.path.join(ROOT, "tests", "cpu_emul", "libalsep_emul_asan.so"))
_lib.DEVICE_TYPE = "cpu"
ctx = _lib.Context("cpu")
from audiolab_amd.synth import synthetic_state_dict  # noqa: E402
from audiolab_amd.tdfnet import TDFNet, TDFNetConfig  # noqa: E402

# dim_f 768 = 16 x 48 = 12 x 64: level 0 (c = 48) 32 x 768, level 1 (c = 96) 16 x 384, level 2 (c = 144) 8 x 192: every conv class dispatches
for kw, b in [(dict(dim_f=768, dim_t=32, n_fft=2048, hop=64, num_blocks=5, g=48), 2),
              (dict(dim_f=192, dim_t=16, n_fft=512, hop=64, num_blocks=3, g=48), 3)]:
    cfg = TDFNetConfig(**kw)
    net = TDFNet(cfg, synthetic_state_dict(cfg, calib_frames=16), ctx=ctx, dtype=torch.bfloat16, max_batch=b)
    ctx.launch_counts_reset()
    y = net.forward_nhwc(torch.randn(b, cfg.dim_t, cfg.dim_f, 4).to(torch.bfloat16))
    assert torch.isfinite(y.float()).all()
    names = ("conv3x3_bf16_m0_kernel", "conv3x3_bf16_mq_kernel", "conv3x3_bf16_mny_kernel<2>", "conv3x3_bf16_mny_kernel<3>",
             "conv3x3_bf16_big_kernel<2>", "conv3x3_bf16_big_kernel<3>", "conv3x3_bf16_regw_kernel", "conv3x3_bf16_kernel<64>")
    print(mode, kw, {n: ctx.launch_count(n) for n in names if ctx.launch_count(n)})
print("conv asan ok", mode)
