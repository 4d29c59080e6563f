"""TEST-ONLY: writes a TFC-TDF U-Net as an ONNX file the way ``torch.onnx.export`` lays out the KUIELab ConvTDFNet in
eval mode (Conv+BN fused into the convolution, MatMul [+Add] + BatchNormalization + Relu for the TDF linears,
ConvTranspose + BatchNormalization, Mul skips, two Transpose nodes), with a minimal protobuf encoder -- the image has
no ``onnx`` package.  Used to exercise audiolab_amd.onnx_reader; field numbers follow the public onnx.proto3."""
import struct

import numpy as np

BN_EPS = 1e-5


def _vi(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _f(fno: int, wt: int) -> bytes:
    return _vi((fno << 3) | wt)


def _ld(fno: int, payload: bytes) -> bytes:
    return _f(fno, 2) + _vi(len(payload)) + payload


def _i(fno: int, v: int) -> bytes:
    return _f(fno, 0) + _vi(v)


def tensor(name: str, arr, style: str = "raw") -> bytes:
    arr = np.ascontiguousarray(arr)
    out = b"".join(_i(1, int(d)) for d in arr.shape)
    if arr.dtype == np.int64:
        out += _i(2, 7) + _ld(8, name.encode()) + _ld(9, arr.tobytes())
        return out
    arr = arr.astype("<f4")
    out += _i(2, 1) + _ld(8, name.encode())
    if style == "raw":
        out += _ld(9, arr.tobytes())
    else:                                                    # packed float_data
        out += _ld(4, arr.tobytes())
    return out


def _attr(name: str, v) -> bytes:
    out = _ld(1, name.encode())
    if isinstance(v, float):
        out += _f(2, 5) + struct.pack("<f", v) + _i(20, 1)
    elif isinstance(v, int):
        out += _i(3, v) + _i(20, 2)
    else:
        out += b"".join(_i(8, int(x)) for x in v) + _i(20, 7)
    return out


def node(op: str, ins, outs, name: str = "", **attrs) -> bytes:
    out = b"".join(_ld(1, s.encode()) for s in ins) + b"".join(_ld(2, s.encode()) for s in outs)
    out += _ld(3, name.encode()) + _ld(4, op.encode())
    out += b"".join(_ld(5, _attr(k, v)) for k, v in attrs.items())
    return out


def value_info(name: str, dims) -> bytes:
    shape = b""
    for d in dims:
        shape += _ld(1, _i(1, d) if isinstance(d, int) else _ld(2, str(d).encode()))
    return _ld(1, name.encode()) + _ld(2, _ld(1, _i(1, 1) + _ld(2, shape)))


def write_mdx_onnx(path: str, sd, cfg, batch="batch_size", float_style: str = "raw", weights_as_inputs: bool = False) -> None:
    """sd: torch state_dict in audiolab_amd.tdfnet naming (un-folded BatchNorm everywhere)."""
    nodes, inits, counter = [], [], [0]

    def fresh(prefix="t"):
        counter[0] += 1
        return f"{prefix}_{counter[0]}"

    def init(arr, prefix="onnx::w"):
        name = fresh(prefix)
        inits.append((name, np.asarray(arr)))
        return name

    def npy(key):
        return sd[key].detach().cpu().numpy().astype(np.float64)

    def bn_terms(p):
        scale = npy(p + ".weight") / np.sqrt(npy(p + ".running_var") + BN_EPS)
        return scale, npy(p + ".bias") - npy(p + ".running_mean") * scale

    def conv_fused(x, p, kernel, stride, pad, relu=True, has_bn=True):
        w = npy(p + ".0.weight")
        b = npy(p + ".0.bias") if p + ".0.bias" in sd else np.zeros(w.shape[0])
        if has_bn:
            s, sh = bn_terms(p + ".1")
            w, b = w * s[:, None, None, None], b * s + sh
        y = fresh()
        nodes.append(node("Conv", [x, init(w.astype(np.float32)), init(b.astype(np.float32))], [y], name=fresh("Conv"),
                          dilations=[1, 1], group=1, kernel_shape=[kernel, kernel], pads=[pad] * 4, strides=[stride, stride]))
        if relu:
            z = fresh()
            nodes.append(node("Relu", [y], [z], name=fresh("Relu")))
            return z
        return y

    def bn_node(x, p):
        y = fresh()
        nodes.append(node("BatchNormalization", [x] + [init(sd[f"{p}.{k}"].numpy(), "bn") for k in
                                                       ("weight", "bias", "running_mean", "running_var")], [y],
                          name=fresh("BatchNormalization"), epsilon=float(BN_EPS), momentum=0.9))
        return y

    def relu(x):
        y = fresh()
        nodes.append(node("Relu", [x], [y], name=fresh("Relu")))
        return y

    def block(x, p):
        for j in range(cfg.l):
            x = conv_fused(x, f"{p}.tfc.H.{j}", cfg.k, 1, cfg.k // 2)
        if cfg.bn is None:
            return x
        t = x
        for j in range(1 if cfg.bn == 0 else 2):
            lin = f"{p}.tdf.{3 * j}"
            y = fresh()
            nodes.append(node("MatMul", [t, init(sd[lin + ".weight"].numpy().T, "onnx::MatMul")], [y], name=fresh("MatMul")))
            if lin + ".bias" in sd:
                z = fresh()
                nodes.append(node("Add", [init(sd[lin + ".bias"].numpy(), "bias"), y], [z], name=fresh("Add")))   # exporter order
                y = z
            t = relu(bn_node(y, f"{p}.tdf.{3 * j + 1}"))
        y = fresh()
        nodes.append(node("Add", [x, t], [y], name=fresh("Add")))
        return y

    def transpose(x):
        y = fresh()
        nodes.append(node("Transpose", [x], [y], name=fresh("Transpose"), perm=[0, 1, 3, 2]))
        return y

    x = conv_fused("input", "first_conv", 1, 1, 0)
    x = transpose(x)
    skips = []
    for i in range(cfg.n):
        x = block(x, f"encoding_blocks.{i}")
        skips.append(x)
        x = conv_fused(x, f"ds.{i}", 2, 2, 0)
    x = block(x, "bottleneck_block")
    for i in range(cfg.n):
        p = f"us.{i}"
        y = fresh()
        ins = [x, init(sd[p + ".0.weight"].numpy())]
        if p + ".0.bias" in sd:
            ins.append(init(sd[p + ".0.bias"].numpy()))
        nodes.append(node("ConvTranspose", ins, [y], name=fresh("ConvTranspose"), dilations=[1, 1], group=1, kernel_shape=[2, 2],
                          pads=[0, 0, 0, 0], strides=[2, 2]))
        x = relu(bn_node(y, p + ".1"))
        y = fresh()
        nodes.append(node("Mul", [x, skips[-i - 1]], [y], name=fresh("Mul")))
        x = block(y, f"decoding_blocks.{i}")
    x = transpose(x)
    w = npy("final_conv.0.weight")
    ins = [x, init(w.astype(np.float32))]
    if "final_conv.0.bias" in sd:
        ins.append(init(sd["final_conv.0.bias"].numpy()))
    nodes.append(node("Conv", ins, ["output"], name=fresh("Conv"), dilations=[1, 1], group=1, kernel_shape=[1, 1], pads=[0] * 4,
                      strides=[1, 1]))

    graph = b"".join(_ld(1, n) for n in nodes) + _ld(2, b"torch_jit")
    graph += b"".join(_ld(5, tensor(name, arr, float_style)) for name, arr in inits)
    io_shape = [batch, cfg.dim_c, cfg.dim_f, cfg.dim_t]
    graph += _ld(11, value_info("input", io_shape))
    if weights_as_inputs:                                    # pre-IR4 exporters list every initializer as a graph input
        graph += b"".join(_ld(11, value_info(name, list(arr.shape))) for name, arr in inits)
    graph += _ld(12, value_info("output", io_shape))
    model = _i(1, 7) + _ld(2, b"pytorch") + _ld(3, b"2.1") + _ld(7, graph) + _ld(8, _ld(1, b"") + _i(2, 13))
    with open(path, "wb") as f:
        f.write(model)
