"""Plain torch-CPU fp32 reference of the TFC-TDF U-Net (MDX-Net "ConvTDFNet") forward.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Not imported by the product.

PARITY UNPINNED: the network that ``Separator.load_model('*.onnx')`` executes
(/root/reference/modules/separator/stem_separator.py:394,512 via
handlers/patch_separate.py:46-62) is stored in downloaded ONNX files that are absent
from /root/reference and this image.  The topology below restates the published
KUIELab TFC-TDF-U-Net v2 design those files were exported from (package
``audio-separator[gpu]>=0.32.0``, setup.sh:96): first 1x1 conv, n encoder TFC_TDF blocks
with 2x2/stride-2 down convs, bottleneck, n decoder blocks with 2x2 transposed up convs
and *multiplicative* skips, final 1x1 conv; the network body runs on [B,C,T,F].
The only in-tree facts used: model I/O is ``[B,4,dim_f,dim_t]`` float32 with input name
"input" (patch_separate.py:52; mdxnet.py:56,170) and L=11 blocks (mdxnet.py:83).

This file is the "plain PyTorch fp32 reference of the same op" for the floating-point
HIP kernels: weights are a flat ``{name: tensor}`` dict using torch ``state_dict`` names,
BatchNorm applied un-folded in eval mode.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x: torch.Tensor, w: Dict[str, torch.Tensor], p: str) -> torch.Tensor:
    if p + ".weight" not in w:                            # BatchNorm folded into the convolution by an ONNX export
        return x
    return F.batch_norm(x, w[p + ".running_mean"], w[p + ".running_var"],
                        w[p + ".weight"], w[p + ".bias"], training=False, eps=BN_EPS)


def _st(x: torch.Tensor, storage) -> torch.Tensor:
    """what a kernel's store leaves in HBM: the fp32 result rounded to the storage type (None: fp32, no rounding)"""
    return x if storage is None else x.to(storage).float()


def _tfc_tdf(x: torch.Tensor, w: Dict[str, torch.Tensor], p: str, l: int, bn: int, storage=None) -> torch.Tensor:
    for j in range(l):                                   # TFC: l x (conv3x3, BN, ReLU)
        q = f"{p}.tfc.H.{j}"
        x = F.conv2d(x, _st(w[q + ".0.weight"], storage), w.get(q + ".0.bias"), padding=1)
        x = _st(F.relu(_bn(x, w, q + ".1")), storage)
    if bn is None:
        return x
    t = x                                                # TDF: linear over the F axis
    n_lin = 1 if bn == 0 else 2
    for j in range(n_lin):
        q = f"{p}.tdf"
        t = F.linear(t, _st(w[f"{q}.{3 * j}.weight"], storage), w.get(f"{q}.{3 * j}.bias"))
        t = F.relu(_bn(t, w, f"{q}.{3 * j + 1}"))
        if j + 1 < n_lin:
            t = _st(t, storage)                          # the hidden activation is stored; the last linear adds x before its store
    return _st(x + t, storage)


def forward(w: Dict[str, torch.Tensor], x: torch.Tensor, num_blocks: int = 11, l: int = 3,
            bn: int = 8, storage=None) -> torch.Tensor:
    """x [B,4,dim_f,dim_t] -> [B,4,dim_f,dim_t] (fp32, CPU).

    ``storage=torch.bfloat16`` (or float16) restates the HALF-PRECISION STORAGE mode of the kernels: every weight
    matrix and every activation a kernel writes to HBM is rounded to that type, all arithmetic in between stays fp32
    (the MFMA accumulators, the folded BatchNorm, the residual add and the skip multiply happen before the store).
    Against this variant a bf16 kernel must agree to accumulation-order noise; against ``storage=None`` the
    difference is the cost of the storage type itself."""
    n = num_blocks // 2
    x = _st(x, storage)
    x = F.conv2d(x, w["first_conv.0.weight"], w.get("first_conv.0.bias"))    # the two 1x1 convs keep fp32 weights
    x = _st(F.relu(_bn(x, w, "first_conv.1")), storage)
    x = x.transpose(-1, -2)                              # [B,C,T,F]
    skips = []
    for i in range(n):
        x = _tfc_tdf(x, w, f"encoding_blocks.{i}", l, bn, storage)
        skips.append(x)
        x = F.conv2d(x, _st(w[f"ds.{i}.0.weight"], storage), w.get(f"ds.{i}.0.bias"), stride=2)
        x = _st(F.relu(_bn(x, w, f"ds.{i}.1")), storage)
    x = _tfc_tdf(x, w, "bottleneck_block", l, bn, storage)
    for i in range(n):
        x = F.conv_transpose2d(x, _st(w[f"us.{i}.0.weight"], storage), w.get(f"us.{i}.0.bias"), stride=2)
        x = F.relu(_bn(x, w, f"us.{i}.1"))
        x = _st(x * skips[-i - 1], storage)              # the skip multiply is fused into the up-conv's epilogue
        x = _tfc_tdf(x, w, f"decoding_blocks.{i}", l, bn, storage)
    x = x.transpose(-1, -2)
    return _st(F.conv2d(x, w["final_conv.0.weight"], w.get("final_conv.0.bias")), storage)
