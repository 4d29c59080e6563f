"""ctypes binding of libalsep.so (include/alsep.h) -- the only door from the Python host to
the HIP kernels.  There is NO fallback: if the in-tree gfx950 library is missing or a call
fails, an exception is raised.

PyTorch is used here only as the device allocator / stream provider: every pointer handed to
the C ABI is ``tensor.data_ptr()`` of a contiguous tensor on the runtime's device.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Iterable, Optional, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libalsep.so")

F32, BF16, F16 = 0, 1, 2
PROF_CONV3X3, PROF_CONV3X3_SMALL, PROF_TDF, PROF_PIX, PROF_POINTWISE, PROF_STFT, PROF_ISTFT = 1, 2, 3, 4, 5, 6, 7
PROF_CONV3X3_REGW, PROF_CONV3X3_PIPE, PROF_CONV3X3_BIG, PROF_CONV3X3_BIG3 = 8, 9, 10, 11
PROF_NN_GEMM, PROF_NN_CONV, PROF_NN_GEMM_H, PROF_NN_ATTN_H, PROF_NN_CONV_H = 12, 13, 14, 15, 16
LAYOUT_REF, LAYOUT_NHWC = 0, 1
ABI_VERSION = 1

# device type every tensor argument must live on
DEVICE_TYPE = "cuda"

_LIB: Optional[C.CDLL] = None


class AlsepError(RuntimeError):
    pass


NET_SPLIT_F16 = 1          # alsep_net_config.flags: float32 storage, contractions as split-half products on the 16-bit matrix pipe


class NetConfig(C.Structure):
    _fields_ = [("dim_f", C.c_int32), ("dim_t", C.c_int32), ("num_blocks", C.c_int32),
                ("l", C.c_int32), ("g", C.c_int32), ("bn", C.c_int32), ("dtype", C.c_int32),
                ("flags", C.c_int32)]


class TensorEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("numel", C.c_int64)]


_SIGNATURES = {
    "alsep_abi_version": (C.c_int, []),
    "alsep_experiments_enabled": (C.c_int, []),
    "alsep_create": (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    "alsep_destroy": (C.c_int, [C.c_void_p]),
    "alsep_last_error": (C.c_char_p, [C.c_void_p]),
    "alsep_profile_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "alsep_profile_end": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "alsep_profile_work": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "alsep_launch_count": (C.c_int64, [C.c_void_p, C.c_char_p]),
    "alsep_launch_counts_reset": (C.c_int, [C.c_void_p]),
    "alsep_plan_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "alsep_plan_destroy": (C.c_int, [C.c_void_p]),
    "alsep_plan_supported_nfft": (C.c_int, [C.c_int]),
    "alsep_stft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64,
                             C.c_void_p, C.c_int, C.c_int]),
    "alsep_istft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_void_p,
                              C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "alsep_spec_convert": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64,
                                     C.c_int64, C.c_int64]),
    "alsep_ola_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_float,
                                    C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "alsep_ola_partial": (C.c_int, [C.c_void_p, C.c_void_p] + [C.c_int64] * 5 + [C.c_int, C.c_void_p, C.c_int64, C.c_int64]),
    "alsep_ola_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_int64]),
    "alsep_resample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int,
                                 C.c_float, C.c_float]),
    "alsep_resample_poly": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int]),
    "alsep_resample_fft_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "alsep_resample_fft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p,
                                     C.c_int64]),
    "alsep_zero_low_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int]),
    "alsep_net_range_flag": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "alsep_nn_set_contraction": (C.c_int, [C.c_void_p, C.c_int]),
    "alsep_nn_range_flag": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "alsep_net_create": (C.c_int, [C.c_void_p, C.POINTER(NetConfig), C.POINTER(TensorEntry), C.c_int64,
                                   C.POINTER(C.c_void_p)]),
    "alsep_net_destroy": (C.c_int, [C.c_void_p]),
    "alsep_net_workspace_bytes": (C.c_int64, [C.c_void_p, C.c_int64]),
    "alsep_net_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_float, C.c_float, C.c_float]),
    "alsep_net_forward_pcm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                        C.c_int64, C.c_float, C.c_float, C.c_float, C.c_int]),
    "alsep_axpby": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int64]),
    "alsep_peak_abs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "alsep_scale_by_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_float]),
    "alsep_dot3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "alsep_xcorr_window": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "alsep_shift_subtract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float,
                                       C.c_void_p]),
    "alsep_vr_conv2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] +
                        [C.c_int] * 14),
    "alsep_vr_lstm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 6),
    "alsep_vr_depthwise": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 7),
    "alsep_vr_resize_bilinear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 7),
    "alsep_vr_copy_slice": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 6),
    "alsep_vr_mean_h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int]),
    "alsep_vr_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 5 + [C.c_float]),
    "alsep_nn_conv2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 15),
    "alsep_nn_bgemm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 +
                       [C.POINTER(C.c_int64)] * 3 + [C.c_float]),
    "alsep_nn_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "alsep_nn_softmax_rows_ld": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "alsep_nn_stats_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "alsep_nn_norm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_float,
                                C.c_int, C.c_void_p]),
    "alsep_nn_meanstd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "alsep_nn_affine_stats": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_float, C.c_int]),
    "alsep_nn_act": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "alsep_nn_scale_add": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "alsep_nn_add_bcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int64, C.c_int, C.c_int]),
    "alsep_nn_vec_fma": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "alsep_nn_vec_div": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "alsep_nn_swap_last2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64]),
    "alsep_nn_reflect_pad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "alsep_nn_tconv_fold": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64] + [C.c_int] * 7),
    "alsep_demucs_spec_in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float]),
    "alsep_demucs_spec_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_float]),
    "alsep_demucs_mix_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64]),
    "alsep_nn_bgemm_bias": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5 +
                            [C.POINTER(C.c_int64)] * 3 + [C.c_float, C.c_void_p, C.c_int]),
    "alsep_nn_rmsnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64]),
    "alsep_nn_rotary": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    "alsep_nn_gate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int]),
    "alsep_roformer_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "alsep_roformer_mask": (C.c_int, [C.c_void_p] * 7 + [C.c_int] * 3),
    "alsep_nn_instnorm_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int]),
    "alsep_nn_instnorm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_int, C.c_void_p]),
    "alsep_nn_instnorm_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_int,
                                        C.c_void_p]),
    "alsep_nn_instnorm_f16_t": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_float, C.c_int,
                                          C.c_void_p]),
    "alsep_nn_conv2d_f16_workspace_bytes": (C.c_int64, [C.c_int64] + [C.c_int] * 10),
    "alsep_nn_conv2d_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64] + [C.c_int] * 12 +
                            [C.c_void_p, C.c_int64]),
    "alsep_nn_mul": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "alsep_nn_depth_to_space2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 5),
    "alsep_mdx23c_spec_in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "alsep_mdx23c_spec_out": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "alsep_vr_band_crop": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8),
    "alsep_vr_split_pred": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "alsep_vr_mirror": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 4),
    "alsep_vr_band_spec": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p] + [C.c_int] * 8),
    "alsep_nn_to_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "alsep_nn_rmsnorm_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64, C.c_int64]),
    "alsep_nn_gemm_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int64,
                                    C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_int, C.c_void_p]),
    "alsep_nn_attention_f16": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64,
                                         C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "alsep_nn_rotary_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "alsep_roformer_bandsplit_in": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "alsep_reverb_workspace_bytes": (C.c_int64, [C.c_int64, C.c_int64]),
    "alsep_reverb_xcorr_argmax": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                            C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]),
    "alsep_reverb_wiener_ir": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_int64,
                                         C.c_double, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.POINTER(C.c_int64)]),
    "alsep_reverb_envelope_db": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "alsep_rfft_mag_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]),
    "alsep_dft_f64_workspace_bytes": (C.c_int64, [C.c_int64]),
    "alsep_dft_f64": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64]),
}

EXPORTS: Tuple[str, ...] = tuple(_SIGNATURES)


def bind(path: str) -> C.CDLL:
    """dlopen ``path`` and attach the prototypes of include/alsep.h; raises if any symbol is missing."""
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the export is missing
        fn.restype = res
        fn.argtypes = args
    if lib.alsep_abi_version() != ABI_VERSION:
        raise AlsepError(f"{path}: ABI version {lib.alsep_abi_version()} != {ABI_VERSION}")
    return lib


def get_lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise AlsepError(
                f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; "
                f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback.")
        _LIB = bind(LIB_PATH)
    return _LIB


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if t.device.type != DEVICE_TYPE:
        raise AlsepError(f"tensor on {t.device}, expected a {DEVICE_TYPE} tensor")
    if not t.is_contiguous():
        raise AlsepError("tensor must be contiguous")
    return t.data_ptr()


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return F32
    if dt == torch.bfloat16:
        return BF16
    if dt == torch.float16:
        return F16
    raise AlsepError(f"unsupported dtype {dt}")


def torch_dtype(code: int) -> torch.dtype:
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[code]


class Context:
    """One alsep_ctx per (device, stream); all launches go to ``stream`` (default: the current
    torch stream of ``device`` at construction)."""

    def __init__(self, device: torch.device | str | None = None, stream: Optional[int] = None):
        self.lib = get_lib()
        if device is None:
            device = torch.device(DEVICE_TYPE, 0) if DEVICE_TYPE == "cuda" else torch.device("cpu")
        self.device = torch.device(device)
        if self.device.type != DEVICE_TYPE:
            raise AlsepError(f"alsep runs on {DEVICE_TYPE} devices, got {self.device}")
        index = self.device.index or 0
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream if DEVICE_TYPE == "cuda" else 0
        self.stream = stream
        h = C.c_void_p()
        rc = self.lib.alsep_create(index, C.c_void_p(stream), C.byref(h))
        if rc != 0:
            raise AlsepError(f"alsep_create failed ({rc})")
        self.handle = h
        self.nn_split = False

    def check(self, rc: int, what: str) -> None:
        if rc != 0:
            msg = self.lib.alsep_last_error(self.handle)
            raise AlsepError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")

    def set_nn_contraction(self, split: bool) -> None:
        """the generic float32 GEMM / convolution entry points of this context: split-half products on the f16 matrix pipe (True) or
        exact f32 MFMA (False, the library's default).  Not inside a stream capture (the first switch to True allocates a device word)."""
        self.check(self.lib.alsep_nn_set_contraction(self.handle, 1 if split else 0), "alsep_nn_set_contraction")
        self.nn_split = bool(split)

    def nn_range_exceeded(self) -> bool:
        """True when a split-contraction launch of this context met an operand beyond the half range since the last call (the results
        since then are invalid); reads and clears the word, synchronises the stream"""
        flag = C.c_int32(0)
        self.check(self.lib.alsep_nn_range_flag(self.handle, C.byref(flag)), "alsep_nn_range_flag")
        return bool(flag.value)

    def profile_begin(self, category: int) -> None:
        self.check(self.lib.alsep_profile_begin(self.handle, category), "alsep_profile_begin")

    def profile_end(self):
        """-> (summed kernel milliseconds, launches) of the class selected by profile_begin."""
        ms, n = C.c_double(), C.c_int64()
        self.check(self.lib.alsep_profile_end(self.handle, C.byref(ms), C.byref(n)), "alsep_profile_end")
        return ms.value, n.value

    def profile_work(self):
        """-> (flops, bytes) the launch sites of the profiled class counted since profile_begin (call before profile_end)"""
        f, b = C.c_double(), C.c_double()
        self.check(self.lib.alsep_profile_work(self.handle, C.byref(f), C.byref(b)), "alsep_profile_work")
        return f.value, b.value

    def launch_count(self, kernel: str) -> int:
        """launches of ``kernel`` (name as reported by its launch site) since creation / the last reset"""
        return int(self.lib.alsep_launch_count(self.handle, kernel.encode()))

    def launch_counts_reset(self) -> None:
        self.check(self.lib.alsep_launch_counts_reset(self.handle), "alsep_launch_counts_reset")

    def synchronize(self) -> None:
        if DEVICE_TYPE == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def empty(self, shape, dtype=torch.float32) -> torch.Tensor:
        return torch.empty(shape, dtype=dtype, device=self.device)

    def zeros(self, shape, dtype=torch.float32) -> torch.Tensor:
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def close(self) -> None:
        if getattr(self, "handle", None):
            self.lib.alsep_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT_CTX: Dict[str, Context] = {}


def default_context(device=None) -> Context:
    key = str(device)
    if key not in _DEFAULT_CTX:
        _DEFAULT_CTX[key] = Context(device)
    return _DEFAULT_CTX[key]
