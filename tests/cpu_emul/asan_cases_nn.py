"""TEST-ONLY: AddressSanitizer pass over the fp32 model families' kernels (tiled GEMM incl. the [K][N] / ragged-K forms, tiled implicit-
GEMM conv with both tile sizes, InstanceNorm statistics, softmax with a leading dimension) on the emulation's ASan build: one small
forward of Roformer (Mel and BS), MDX23C and HTDemucs.  Shapes chosen so that the tiled kernels dispatch (channel counts % 16 == 0).
    LD_PRELOAD=$(clang++ -print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 python tests/cpu_emul/asan_cases_nn.py"""
import dataclasses
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from audiolab_amd import _lib  # noqa: E402

_lib._LIB = _lib.bind(os.path.join(ROOT, "tests", "cpu_emul", "libalsep_emul_asan.so"))
_lib.DEVICE_TYPE = "cpu"
ctx = _lib.Context("cpu")
from audiolab_amd.htdemucs import HTDemucs, HTDemucsConfig  # noqa: E402
from audiolab_amd.mdx23c import MDX23C, MDX23CConfig  # noqa: E402
from audiolab_amd.roformer import Roformer, RoformerConfig  # noqa: E402
from oracle import htdemucs_oracle as ho  # noqa: E402
from oracle import mdx23c_oracle as mo  # noqa: E402
from oracle import roformer_oracle as ro  # noqa: E402

names = ("nn_gemm_tn_kernel", "nn_bgemm_kernel", "nn_conv2d_tiled_kernel", "nn_conv2d_kernel", "nn_softmax_rows_kernel")
BS_BANDS = (2,) * 8 + (4,) * 6 + (8,) * 5 + (16,) * 3 + (1,)          # 129 bins
for kind in ("mel", "bs"):
    ocfg = ro.RoformerConfig(kind=kind, dim=32, depth=2, heads=2, dim_head=16, n_fft=256, hop=60, num_bands=10, freqs_per_bands=BS_BANDS,
                             sample_rate=8000, chunk_size=60 * 30, num_overlap=2, num_stems=2 if kind == "bs" else 1)
    net = Roformer(RoformerConfig(**dataclasses.asdict(ocfg)), ro.synthetic_state_dict(ocfg, 1), ctx=ctx)
    ctx.launch_counts_reset()
    y = net.forward(torch.randn(2, ocfg.chunk_size) * 0.2)
    assert torch.isfinite(y).all()
    print("roformer", kind, {n: ctx.launch_count(n) for n in names if ctx.launch_count(n)})
ocfg = mo.MDX23CConfig(instruments=("vocals", "other"), n_fft=256, hop=64, dim_f=96, num_subbands=2, num_scales=2, num_blocks_per_scale=2,
                       num_channels=16, growth=16, bottleneck_factor=4, chunk_size=64 * 15, num_overlap=2, sample_rate=8000)
net = MDX23C(MDX23CConfig(**dataclasses.asdict(ocfg)), mo.synthetic_state_dict(ocfg, 1), ctx=ctx)
ctx.launch_counts_reset()
y = net.forward(torch.randn(2, ocfg.chunk_size) * 0.2)
assert torch.isfinite(y).all()
print("mdx23c", {n: ctx.launch_count(n) for n in names if ctx.launch_count(n)})
ocfg = ho.HTDemucsConfig(sources=("drums", "bass", "other"), channels=16, nfft=256, depth=2, dconv_comp=4, bottom_channels=32, t_layers=3,
                         t_heads=4, segment_samples=2560, samplerate=4000)
net = HTDemucs(HTDemucsConfig(**dataclasses.asdict(ocfg)), ho.synthetic_state_dict(ocfg, 1), ctx=ctx)
ctx.launch_counts_reset()
y = net.forward(torch.randn(2, ocfg.segment_samples) * 0.2)
assert torch.isfinite(y).all()
print("htdemucs", {n: ctx.launch_count(n) for n in names if ctx.launch_count(n)})
print("nn asan ok")
