"""Real-weight loader for MDX-Net ``*.onnx`` files.

The reference hands the file name to ``audio_separator`` (modules/separator/stem_separator.py:394,512) and then
swaps the ONNX session in through handlers/patch_separate.py:46-62 (``ort.InferenceSession(self.model_path)``, input
name ``"input"``).  Neither ``onnx`` nor ``onnxruntime`` exists in this image, so this module reads the protobuf wire
format itself (ModelProto -> GraphProto -> NodeProto / TensorProto, field numbers of the public onnx.proto3) and maps
the exported TFC-TDF U-Net graph onto the ``state_dict`` names ``audiolab_amd.tdfnet`` consumes.

What an exported MDX-Net graph looks like (torch.onnx export of the KUIELab ConvTDFNet in eval mode; PARITY
UNPINNED: no model file is reachable offline, the test vectors are written by tests/onnx_writer.py in that style):
``Conv`` nodes carry their BatchNorm folded into weight + bias (the exporter fuses Conv+BN); ``MatMul`` (the TDF
linears, weight stored transposed as [f_in, f_out]) is followed by an optional bias ``Add``, a ``BatchNormalization``
over the channel axis and ``Relu``; ``ConvTranspose`` keeps a separate ``BatchNormalization``; the skips are ``Mul``;
two ``Transpose`` nodes wrap the body.  The walker below does not rely on node or tensor names, only on data flow:
every Conv / ConvTranspose / MatMul opens a unit, bias ``Add`` and ``BatchNormalization`` nodes that consume the
unit's value attach to it, and the ordered unit kinds must spell
``c1 (c3^l mm^m ds)^n c3^l mm^m (us c3^l mm^m)^n c1`` -- anything else is refused loudly.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np
import torch

from ._lib import AlsepError
from .tdfnet import BN_EPS, TDFNetConfig


# ---- protobuf wire format -------------------------------------------------------------------------
def _varint(buf, i: int) -> Tuple[int, int]:
    shift = result = 0
    while True:
        if i >= len(buf):
            raise AlsepError("onnx: truncated varint")
        b = buf[i]
        i += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, i
        shift += 7
        if shift > 70:
            raise AlsepError("onnx: malformed varint")


def _fields(buf) -> Iterator[Tuple[int, int, object]]:
    """(field number, wire type, value) over one message; length-delimited values are memoryview slices (no copy)."""
    i, n = 0, len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
        elif wt == 1:
            v, i = buf[i:i + 8], i + 8
        elif wt == 2:
            ln, i = _varint(buf, i)
            v, i = buf[i:i + ln], i + ln
        elif wt == 5:
            v, i = buf[i:i + 4], i + 4
        else:
            raise AlsepError(f"onnx: unsupported wire type {wt}")
        if i > n:
            raise AlsepError("onnx: truncated message")
        yield fno, wt, v


def _sint(v: int) -> int:
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_ints(wt: int, v) -> List[int]:
    if wt == 0:
        return [_sint(v)]
    out, i = [], 0
    while i < len(v):
        x, i = _varint(v, i)
        out.append(_sint(x))
    return out


_DTYPES = {1: "<f4", 6: "<i4", 7: "<i8", 10: "<f2", 11: "<f8"}


def _tensor(buf) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype, name, raw = 1, "", None
    floats: List[np.ndarray] = []
    ints: List[int] = []
    external = False
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims += _packed_ints(wt, v)
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = v
        elif fno == 4:                                       # float_data, packed or one by one
            floats.append(np.frombuffer(v, "<f4"))
        elif fno in (5, 7):                                  # int32_data / int64_data
            ints += _packed_ints(wt, v)
        elif fno == 10:
            floats.append(np.frombuffer(v, "<f8"))
        elif fno == 14 and v == 1:
            external = True
    if external:
        raise AlsepError(f"onnx: tensor '{name}' uses external data, which is not supported")
    if dtype == 16:                                          # bfloat16 raw -> float32
        arr = (np.frombuffer(raw, "<u2").astype(np.uint32) << 16).view("<f4") if raw is not None else np.zeros(0, "<f4")
    elif dtype not in _DTYPES:
        raise AlsepError(f"onnx: tensor '{name}' has unsupported data type {dtype}")
    elif raw is not None:
        arr = np.frombuffer(raw, _DTYPES[dtype])
    elif floats:
        arr = np.concatenate(floats)
    else:
        arr = np.asarray(ints, dtype=_DTYPES[dtype])
    n = int(np.prod(dims)) if dims else arr.size
    if arr.size != n:
        raise AlsepError(f"onnx: tensor '{name}' holds {arr.size} values for shape {dims}")
    return name, arr.reshape(dims)


@dataclass
class Node:
    op: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)
    name: str = ""


def _attribute(buf) -> Tuple[str, object]:
    name, val = "", None
    ints: List[int] = []
    floats: List[float] = []
    have_ints = have_floats = False
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", bytes(v))[0]
        elif fno == 3:
            val = _sint(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            have_floats = True
            floats += list(np.frombuffer(v, "<f4")) if wt == 2 else [struct.unpack("<f", bytes(v))[0]]
        elif fno == 8:
            have_ints = True
            ints += _packed_ints(wt, v)
    if have_ints:
        val = ints
    elif have_floats:
        val = floats
    return name, val


def _node(buf) -> Node:
    nd = Node("", [], [])
    for fno, _wt, v in _fields(buf):
        if fno == 1:
            nd.inputs.append(bytes(v).decode())
        elif fno == 2:
            nd.outputs.append(bytes(v).decode())
        elif fno == 3:
            nd.name = bytes(v).decode()
        elif fno == 4:
            nd.op = bytes(v).decode()
        elif fno == 5:
            k, a = _attribute(v)
            nd.attrs[k] = a
    return nd


def _value_info(buf) -> Tuple[str, List[Optional[int]]]:
    """(name, dims) with None for symbolic dimensions."""
    name, dims = "", []
    for fno, _wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            for f2, _w2, v2 in _fields(v):
                if f2 != 1:                                  # TypeProto.tensor_type
                    continue
                for f3, _w3, v3 in _fields(v2):
                    if f3 != 2:                              # Tensor.shape
                        continue
                    for f4, _w4, v4 in _fields(v3):
                        if f4 != 1:                          # TensorShapeProto.dim
                            continue
                        d = None
                        for f5, w5, v5 in _fields(v4):
                            if f5 == 1 and w5 == 0:
                                d = _sint(v5)
                        dims.append(d)
    return name, dims


@dataclass
class Graph:
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]
    inputs: List[Tuple[str, List[Optional[int]]]]
    outputs: List[Tuple[str, List[Optional[int]]]]


def read_graph(path: str) -> Graph:
    """Parse an ONNX file into nodes (in stored = topological order) and initializers (numpy views of the file bytes)."""
    with open(path, "rb") as f:
        data = memoryview(f.read())
    graph = None
    for fno, wt, v in _fields(data):
        if fno == 7 and wt == 2:
            graph = v
    if graph is None:
        raise AlsepError(f"{path}: not an ONNX model (no graph)")
    g = Graph([], {}, [], [])
    for fno, wt, v in _fields(graph):
        if wt != 2:
            continue
        if fno == 1:
            g.nodes.append(_node(v))
        elif fno == 5:
            name, arr = _tensor(v)
            g.initializers[name] = arr
        elif fno == 11:
            g.inputs.append(_value_info(v))
        elif fno == 12:
            g.outputs.append(_value_info(v))
    g.inputs = [(n, d) for n, d in g.inputs if n not in g.initializers]    # old exporters list weights as inputs too
    for nd in g.nodes:                                       # Constant nodes act as initializers
        if nd.op == "Constant" and "value" in nd.attrs and nd.outputs:
            g.initializers[nd.outputs[0]] = nd.attrs["value"]
    return g


# ---- TFC-TDF U-Net graph -> state_dict --------------------------------------------------------------
@dataclass
class _Unit:
    kind: str                       # conv | convT | matmul
    weight: np.ndarray
    bias: Optional[np.ndarray] = None
    bn: Optional[Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]] = None
    kernel: Tuple[int, int] = (1, 1)
    stride: Tuple[int, int] = (1, 1)
    relu: bool = False

    @property
    def token(self) -> str:
        if self.kind == "matmul":
            return "mm"
        if self.kind == "convT":
            return "us"
        if self.stride != (1, 1):
            return "ds"
        return "c1" if self.kernel == (1, 1) else "c3"


_PASS_THROUGH = {"Identity", "Cast", "Dropout"}


def _collect_units(g: Graph) -> Tuple[List[_Unit], int, int]:
    """Ordered linear units, number of activation x activation Mul (skip) nodes, number of Transpose nodes."""
    init = g.initializers
    units: List[_Unit] = []
    open_unit: Dict[str, _Unit] = {}                         # value name -> unit still accepting bias / BN / ReLU
    skips = transposes = 0

    def t(x) -> Tuple[int, int]:
        return (int(x[0]), int(x[1]))

    for nd in g.nodes:
        op = nd.op
        if op in ("Conv", "ConvTranspose"):
            if nd.inputs[1] not in init:
                raise AlsepError(f"onnx: {op} '{nd.name}' has a non-constant weight")
            w = init[nd.inputs[1]]
            if w.ndim != 4:
                raise AlsepError(f"onnx: {op} '{nd.name}' is not 2-D")
            a = nd.attrs
            if a.get("group", 1) != 1 or any(d != 1 for d in a.get("dilations", [1, 1])):
                raise AlsepError(f"onnx: {op} '{nd.name}': grouped / dilated convolutions are not part of a TFC-TDF U-Net")
            kernel = t(a.get("kernel_shape", w.shape[2:]))
            stride = t(a.get("strides", [1, 1]))
            pads = list(a.get("pads", [0, 0, 0, 0]))
            if op == "Conv" and stride == (1, 1) and pads != [kernel[0] // 2, kernel[1] // 2] * 2:
                raise AlsepError(f"onnx: Conv '{nd.name}': pads {pads} are not 'same' padding for kernel {kernel}")
            if (op == "ConvTranspose" or stride != (1, 1)) and (stride != kernel or any(pads)):
                raise AlsepError(f"onnx: {op} '{nd.name}': scale convolutions must have stride == kernel and no padding")
            u = _Unit("conv" if op == "Conv" else "convT", w, kernel=kernel, stride=stride)
            if len(nd.inputs) > 2 and nd.inputs[2]:
                u.bias = init[nd.inputs[2]]
            units.append(u)
            open_unit[nd.outputs[0]] = u
        elif op == "MatMul":
            a_in, b_in = nd.inputs
            if b_in in init and a_in not in init:
                u = _Unit("matmul", np.ascontiguousarray(init[b_in].T))          # stored [f_in, f_out] -> torch [f_out, f_in]
            else:
                raise AlsepError(f"onnx: MatMul '{nd.name}' without a constant right operand (attention?) is not MDX-Net")
            if u.weight.ndim != 2:
                raise AlsepError(f"onnx: MatMul '{nd.name}' weight is not a matrix")
            units.append(u)
            open_unit[nd.outputs[0]] = u
        elif op == "Add":
            a_in, b_in = nd.inputs
            src, const = (a_in, b_in) if a_in in open_unit else (b_in, a_in)
            if src in open_unit and const in init:
                u = open_unit.pop(src)
                if u.bias is not None or u.bn is not None:
                    raise AlsepError(f"onnx: Add '{nd.name}': second bias on one layer")
                u.bias = init[const].reshape(-1)
                open_unit[nd.outputs[0]] = u
        elif op == "BatchNormalization":
            if nd.inputs[0] not in open_unit:
                raise AlsepError(f"onnx: BatchNormalization '{nd.name}' does not follow a convolution or linear layer")
            u = open_unit.pop(nd.inputs[0])
            gamma, beta, mean, var = (init[n].astype(np.float32) for n in nd.inputs[1:5])
            eps = float(nd.attrs.get("epsilon", 1e-5))
            u.bn = (gamma, beta, mean, (var.astype(np.float64) + (eps - BN_EPS)).astype(np.float32))
            open_unit[nd.outputs[0]] = u
        elif op == "Relu":
            if nd.inputs[0] in open_unit:
                open_unit.pop(nd.inputs[0]).relu = True
        elif op == "Mul":
            if all(n not in init for n in nd.inputs):
                skips += 1
            elif any(n in open_unit for n in nd.inputs):
                raise AlsepError(f"onnx: Mul '{nd.name}' scales a layer output by a constant: unsupported export style")
        elif op == "Transpose":
            if list(nd.attrs.get("perm", [])) != [0, 1, 3, 2]:
                raise AlsepError(f"onnx: Transpose '{nd.name}' perm {nd.attrs.get('perm')} is not the T<->F swap of MDX-Net")
            transposes += 1
            if nd.inputs[0] in open_unit:
                raise AlsepError(f"onnx: Transpose '{nd.name}' sits between a layer and its activation")
        elif op in _PASS_THROUGH:
            if nd.inputs[0] in open_unit:
                open_unit[nd.outputs[0]] = open_unit.pop(nd.inputs[0])
        elif op == "Constant":
            pass
        else:
            raise AlsepError(f"onnx: operator '{op}' ('{nd.name}') is not part of a TFC-TDF U-Net (MDX-Net) graph")
    return units, skips, transposes


@dataclass
class MdxOnnxModel:
    config: TDFNetConfig
    state_dict: Dict[str, torch.Tensor]
    input_name: str


def _put(sd: Dict[str, torch.Tensor], prefix: str, u: _Unit) -> None:
    sd[prefix + ".0.weight"] = torch.from_numpy(np.array(u.weight, dtype=np.float32))
    if u.bias is not None:
        sd[prefix + ".0.bias"] = torch.from_numpy(np.array(u.bias, dtype=np.float32).reshape(-1))
    if u.bn is not None:
        for key, arr in zip(("weight", "bias", "running_mean", "running_var"), u.bn):
            sd[f"{prefix}.1.{key}"] = torch.from_numpy(np.array(arr, dtype=np.float32))


def load_mdx_onnx(path: str, n_fft: Optional[int] = None, hop: int = 1024, dim_t: Optional[int] = None) -> MdxOnnxModel:
    """Read an MDX-Net ONNX file: the network hyper-parameters are inferred from the graph, ``n_fft`` (not stored in the
    file; the reference reads it from audio-separator's model table) comes from the caller, default ``2 * dim_f``.
    A layer whose BatchNorm was folded by the exporter appears in the state_dict without ``.1.*`` entries
    (``tdfnet.fold_batchnorm`` treats that as identity)."""
    g = read_graph(path)
    if len(g.inputs) != 1:
        raise AlsepError(f"{path}: expected one graph input, found {[n for n, _ in g.inputs]}")
    in_name, in_dims = g.inputs[0]
    units, skips, transposes = _collect_units(g)
    tokens = [u.token for u in units]
    spell = " ".join(tokens)
    if len(tokens) < 3 or tokens[0] != "c1" or tokens[-1] != "c1":
        raise AlsepError(f"{path}: not a TFC-TDF U-Net (layer sequence: {spell})")
    l = 0
    while tokens[1 + l] == "c3":
        l += 1
    m = 0
    while tokens[1 + l + m] == "mm":
        m += 1
    n = tokens.count("ds")
    block = ["c3"] * l + ["mm"] * m
    want = ["c1"] + (block + ["ds"]) * n + block + (["us"] + block) * n + ["c1"]
    if l == 0 or tokens != want or tokens.count("us") != n:
        raise AlsepError(f"{path}: layer sequence '{spell}' is not c1 (c3^l mm^m ds)^n c3^l mm^m (us c3^l mm^m)^n c1")
    if skips != n:
        raise AlsepError(f"{path}: {skips} multiplicative skip connections for {n} levels")
    if transposes != 2:
        raise AlsepError(f"{path}: {transposes} T<->F transposes (expected 2: the body runs on [B,C,T,F])")
    for u in units[:-1]:
        if not u.relu:
            raise AlsepError(f"{path}: a hidden layer without ReLU")
    if units[-1].relu or units[-1].bn is not None:
        raise AlsepError(f"{path}: the final 1x1 convolution carries an activation")
    first, last = units[0], units[-1]
    gch, dim_c = int(first.weight.shape[0]), int(first.weight.shape[1])
    if tuple(last.weight.shape[:2]) != (dim_c, gch):
        raise AlsepError(f"{path}: final convolution {last.weight.shape} does not map {gch} -> {dim_c} channels")
    k = units[1].kernel[0]
    if any(u.kernel != (k, k) for u in units if u.token == "c3") or any(u.kernel != (2, 2) for u in units if u.token in ("ds", "us")):
        raise AlsepError(f"{path}: mixed kernel sizes")
    dim_f = in_dims[2] if len(in_dims) == 4 and in_dims[2] else None
    if m:
        f0 = int(units[1 + l].weight.shape[1])               # torch layout [f_out, f_in]
        if dim_f is not None and dim_f != f0:
            raise AlsepError(f"{path}: input has {dim_f} bins, the first TDF layer expects {f0}")
        dim_f = f0
    if dim_f is None:
        raise AlsepError(f"{path}: cannot infer dim_f (symbolic input shape and no TDF layer)")
    if dim_t is None:
        dim_t = in_dims[3] if len(in_dims) == 4 and in_dims[3] else 256
    if len(in_dims) == 4 and in_dims[1] not in (None, dim_c):
        raise AlsepError(f"{path}: input has {in_dims[1]} channels, first convolution expects {dim_c}")
    bn: Optional[int]
    if m == 0:
        bn = None
    elif m == 1:
        bn = 0
    elif m == 2:
        bn = dim_f // int(units[1 + l].weight.shape[0])
    else:
        raise AlsepError(f"{path}: {m} linear layers per TDF block")
    bias = any(u.bias is not None for u in units if u.token == "mm")

    sd: Dict[str, torch.Tensor] = {}
    it = iter(units)

    def block_into(prefix: str, c: int, f: int):
        for j in range(l):
            u = next(it)
            if tuple(u.weight.shape[:2]) != (c, c):
                raise AlsepError(f"{path}: {prefix}.tfc.{j} maps {u.weight.shape[1]} -> {u.weight.shape[0]} channels, expected {c}")
            _put(sd, f"{prefix}.tfc.H.{j}", u)
        for j in range(m):
            u = next(it)
            fin = f if j == 0 else (f // bn if bn else f)
            fout = (f // bn if bn else f) if (j == 0 and m == 2) else f
            if tuple(u.weight.shape) != (fout, fin):
                raise AlsepError(f"{path}: {prefix}.tdf.{j} is {tuple(u.weight.shape)}, expected ({fout}, {fin})")
            if u.bn is None or u.bn[0].shape[0] != c:
                raise AlsepError(f"{path}: {prefix}.tdf.{j} lacks its per-channel BatchNorm")
            sd[f"{prefix}.tdf.{3 * j}.weight"] = torch.from_numpy(np.array(u.weight, dtype=np.float32))
            if u.bias is not None:
                sd[f"{prefix}.tdf.{3 * j}.bias"] = torch.from_numpy(np.array(u.bias, dtype=np.float32))
            for key, arr in zip(("weight", "bias", "running_mean", "running_var"), u.bn):
                sd[f"{prefix}.tdf.{3 * j + 1}.{key}"] = torch.from_numpy(np.array(arr, dtype=np.float32))

    _put(sd, "first_conv", next(it))
    c, f = gch, dim_f
    for i in range(n):
        block_into(f"encoding_blocks.{i}", c, f)
        u = next(it)
        if tuple(u.weight.shape[:2]) != (c + gch, c):
            raise AlsepError(f"{path}: ds.{i} maps {u.weight.shape[1]} -> {u.weight.shape[0]} channels, expected {c} -> {c + gch}")
        _put(sd, f"ds.{i}", u)
        c, f = c + gch, f // 2
    block_into("bottleneck_block", c, f)
    for i in range(n):
        u = next(it)
        if tuple(u.weight.shape[:2]) != (c, c - gch):        # ConvTranspose weight is [Cin, Cout, kh, kw]
            raise AlsepError(f"{path}: us.{i} maps {u.weight.shape[0]} -> {u.weight.shape[1]} channels, expected {c} -> {c - gch}")
        _put(sd, f"us.{i}", u)
        c, f = c - gch, f * 2
        block_into(f"decoding_blocks.{i}", c, f)
    _put(sd, "final_conv", next(it))

    cfg = TDFNetConfig(dim_f=dim_f, dim_t=int(dim_t), n_fft=int(n_fft) if n_fft else 2 * dim_f, hop=hop, num_blocks=2 * n + 1,
                       l=l, g=gch, k=k, bn=bn, bias=bias, dim_c=dim_c)
    return MdxOnnxModel(cfg, sd, in_name)
