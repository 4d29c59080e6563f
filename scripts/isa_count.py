#!/usr/bin/env python3
"""Instruction census of one kernel from hipcc's assembly listing (timing-analysis aid, not part of the product).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I audiolab_amd/csrc -I include -S --cuda-device-only audiolab_amd/csrc/fft.hip -o /tmp/fft.s
    python scripts/isa_count.py /tmp/fft.s <mangled-name-substring> [...]

Counts the instructions of the kernel body by issue class.  For a straight-line (fully unrolled) body such as the three-pass
STFT kernel -- one workgroup = one frame, every instruction executed once per wave -- the counts are per frame and wave; a kernel
with loops needs its trip counts applied by hand (the label structure is printed to make that visible)."""
import re
import sys
from collections import Counter


def classify(op: str) -> str:
    if op.startswith("v_pk_"):
        return "valu_packed"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "valu_trans"
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    lines = open(path).read().split("\n")
    for pat in pats:
        starts = [i for i, l in enumerate(lines) if l.startswith("_Z") and ":" in l and pat in l.split(":")[0]]
        for st in starts:
            name = lines[st].split(":")[0]
            body, labels, branches = [], 0, 0
            for l in lines[st + 1:]:
                t = l.strip()
                if t.startswith("s_endpgm"):
                    break
                if t.startswith(".LBB") and t.endswith(":"):
                    labels += 1
                m = re.match(r"([a-z_0-9]+)\b", t)
                if m and not t.startswith((".", ";")):
                    body.append(m.group(1))
                    if m.group(1).startswith(("s_cbranch", "s_branch")):
                        branches += 1
            c = Counter(classify(op) for op in body)
            tot = sum(c.values())
            vmem = Counter(op for op in body if classify(op) == "vmem")
            lds = Counter(op for op in body if classify(op) == "lds")
            print(f"{name[:110]}\n  {tot} instructions, {labels} labels, {branches} branches")
            print("  " + "  ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])))
            print("  vmem: " + ", ".join(f"{k} x{v}" for k, v in vmem.most_common()))
            print("  lds:  " + ", ".join(f"{k} x{v}" for k, v in lds.most_common()))
            # issue-time floor of one wave on its SIMD (MI355X_MICROARCH.md cycle constants): VALU / packed VALU 4 cycles, transcendental 8
            cyc = 4 * (c["valu"] + c["valu_packed"]) + 8 * c["valu_trans"]
            print(f"  VALU issue floor of one wave: {cyc} cycles (4 per VALU / packed VALU instruction, 8 per transcendental)")


if __name__ == "__main__":
    main()
