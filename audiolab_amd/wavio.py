"""Minimal RIFF/WAVE reader + writer (numpy only) for the on-disk contract of the Separate path:
stems are float32 WAV (``sf.write(..., subtype="FLOAT")``, stem_separator.py:667-669), inputs are
PCM16/PCM24/PCM32/float32 WAV (non-WAV inputs go through ffmpeg in the reference, :31-54, which is
out of scope here).  Data layout on disk is interleaved [N, C]; arrays here are [C, N] float32."""
from __future__ import annotations

import struct
from typing import Tuple

import numpy as np

WAVE_FORMAT_PCM = 1
WAVE_FORMAT_IEEE_FLOAT = 3
WAVE_FORMAT_EXTENSIBLE = 0xFFFE


def read_wav(path: str) -> Tuple[np.ndarray, int]:
    """-> (audio [C, N] float32 in [-1, 1), sample_rate)."""
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 12 or data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos = 12
    fmt = None
    pcm = None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            if len(body) < 16:
                raise ValueError(f"{path}: fmt chunk of {len(body)} bytes (need 16)")
            tag, ch, sr, _, _, bits = struct.unpack("<HHIIHH", body[:16])
            if ch == 0 or sr == 0:
                raise ValueError(f"{path}: fmt chunk declares {ch} channels at {sr} Hz")
            if tag == WAVE_FORMAT_EXTENSIBLE and len(body) >= 26:
                tag = struct.unpack("<H", body[24:26])[0]
            fmt = (tag, ch, sr, bits)
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt or data chunk")
    tag, ch, sr, bits = fmt
    if tag == WAVE_FORMAT_IEEE_FLOAT and bits == 32:
        x = np.frombuffer(pcm, dtype="<f4").astype(np.float32)
    elif tag == WAVE_FORMAT_IEEE_FLOAT and bits == 64:
        x = np.frombuffer(pcm, dtype="<f8").astype(np.float32)
    elif tag == WAVE_FORMAT_PCM and bits == 16:
        x = np.frombuffer(pcm, dtype="<i2").astype(np.float32) / 32768.0
    elif tag == WAVE_FORMAT_PCM and bits == 32:
        x = np.frombuffer(pcm, dtype="<i4").astype(np.float32) / 2147483648.0
    elif tag == WAVE_FORMAT_PCM and bits == 24:
        b = np.frombuffer(pcm[: len(pcm) // 3 * 3], dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) / 8388608.0
    elif tag == WAVE_FORMAT_PCM and bits == 8:
        x = (np.frombuffer(pcm, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    else:
        raise ValueError(f"{path}: unsupported WAV format tag={tag} bits={bits}")
    n = x.size // ch
    return np.ascontiguousarray(x[: n * ch].reshape(n, ch).T), sr


def write_wav(path: str, audio: np.ndarray, sr: int, subtype: str = "FLOAT") -> None:
    """audio [C, N] (or [N]) -> WAV; subtype "FLOAT" (float32) or "PCM_16" (round-to-nearest, clipped)."""
    a = np.asarray(audio)
    if a.ndim == 1:
        a = a[None, :]
    ch, n = a.shape
    inter = np.ascontiguousarray(a.T)
    if subtype == "FLOAT":
        tag, bits = WAVE_FORMAT_IEEE_FLOAT, 32
        payload = inter.astype("<f4").tobytes()
    elif subtype == "PCM_16":
        tag, bits = WAVE_FORMAT_PCM, 16
        payload = np.clip(np.rint(inter.astype(np.float64) * 32768.0), -32768, 32767).astype("<i2").tobytes()
    else:
        raise ValueError(f"unsupported subtype {subtype}")
    block = ch * bits // 8
    fmt = struct.pack("<HHIIHH", tag, ch, sr, sr * block, block, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt
    if tag == WAVE_FORMAT_IEEE_FLOAT:
        chunks += b"fact" + struct.pack("<II", 4, n)
    chunks += b"data" + struct.pack("<I", len(payload)) + payload + (b"\x00" if len(payload) & 1 else b"")
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)
