"""Identity of the kernel sources a measurement belongs to: bench.py only reports PMC traffic collected from the same
sources it is running (profiles/*pmc_traffic.json carries this hash; a stale file yields ``null``, never an old number)."""
from __future__ import annotations

import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def source_hash() -> str:
    """sha256 over csrc/* and include/alsep.h (names + bytes), first 16 hex digits"""
    h = hashlib.sha256()
    csrc = os.path.join(_HERE, "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc))]
    files.append(os.path.join(os.path.dirname(_HERE), "include", "alsep.h"))
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]
