#!/usr/bin/env python3
"""Check the generated gfx950 code of kernels that use global_load_async_bf16x8 (alsep_gfx950_asm.h).

Those loads are inline asm: hipcc does not know their destination registers are in flight until our own
counted ``s_waitcnt vmcnt``.  This script compiles a .hip file to assembly and verifies that, between each
such load and the next ``s_waitcnt vmcnt(..)``, no instruction reads or writes the destination VGPRs (a copy,
a spill, an early MFMA).  Exit status 1 and a listing on any violation.

usage: check_async_regs.py [file.hip ...]      (default: audiolab_amd/csrc/tdfnet.hip)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def compile_asm(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops", "-I", os.path.join(ROOT, "audiolab_amd", "csrc"),
           "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out, src]
    subprocess.run(cmd, check=True, cwd=tempfile.gettempdir())
    text = open(out).read()
    os.unlink(out)
    return text


def split_kernels(text):
    """{kernel name: [(line number, instruction text, in_asm)]} for every function in the assembly."""
    kernels, cur, in_asm = {}, None, False
    for ln, line in enumerate(text.splitlines(), 1):
        s = line.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s.startswith(";"):
            s = s.split(";")[0].strip()
        if s.endswith(":") and not s.startswith(";"):
            if s.startswith(".L"):
                if cur is not None:
                    kernels[cur].append((ln, s, False))
            elif not s.startswith("."):
                cur = s[:-1]
                kernels[cur] = []
            continue
        if cur is None or not s or s.startswith(";") or s.startswith("."):
            continue
        code = s.split(";")[0].strip()
        if code:
            kernels[cur].append((ln, code, in_asm))
    return kernels


def check_kernel(name, insts):
    """Forward may-analysis over the kernel's basic blocks: the set of VGPRs with an asynchronous load in flight."""
    blocks, labels, cur = [], {}, []
    for item in insts:
        ln, code, in_asm = item
        if code.endswith(":"):                               # label starts a block
            if cur:
                blocks.append(cur)
            cur = []
            labels[code[:-1]] = len(blocks)
            continue
        cur.append(item)
        if code.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    # labels recorded before their block was appended: index == position of the next appended block
    succ = []
    for i, blk in enumerate(blocks):
        last = blk[-1][1] if blk else ""
        if last.startswith("s_endpgm"):
            succ.append([])
        elif last.startswith("s_branch"):
            succ.append([labels[last.split()[1]]])
        elif last.startswith("s_cbranch"):
            succ.append([labels[last.split()[1]]] + ([i + 1] if i + 1 < len(blocks) else []))
        else:
            succ.append([i + 1] if i + 1 < len(blocks) else [])
    state_in = [dict() for _ in blocks]
    loads, bad, seen_bad = 0, [], set()
    work, counted, visited = [0], set(), {0}
    while work:
        i = work.pop()
        inflight = dict(state_in[i])
        for ln, code, in_asm in blocks[i]:
            if code.startswith("s_waitcnt") and "vmcnt(" in code:
                inflight = {}
                continue
            if in_asm and code.startswith("global_load_dwordx4"):
                if ln not in counted:
                    counted.add(ln)
                    loads += 1
                hit = vregs(code.split(",", 1)[1]) & set(inflight)
                for r in vregs(code.split(",")[0]):
                    inflight[r] = ln
            else:
                hit = vregs(code) & set(inflight)
            if hit and ln not in seen_bad:
                seen_bad.add(ln)
                bad.append((name, ln, code, sorted(hit)))
        for j in succ[i]:
            merged = dict(state_in[j])
            merged.update(inflight)
            if merged.keys() != state_in[j].keys() or j not in visited:
                visited.add(j)
                state_in[j] = merged
                work.append(j)
    return loads, bad


def check(text):
    """Returns (number of async loads seen, list of violations)."""
    loads, bad = 0, []
    for name, insts in split_kernels(text).items():
        if any(in_asm and code.startswith("global_load_dwordx4") for _, code, in_asm in insts):
            n, b = check_kernel(name, insts)
            loads += n
            bad += b
    return loads, bad


def main(argv):
    files = argv or [os.path.join(ROOT, "audiolab_amd", "csrc", "tdfnet.hip")]
    status = 0
    for f in files:
        loads, bad = check(compile_asm(f))
        print(f"{os.path.relpath(f, ROOT)}: {loads} asynchronous register loads, {len(bad)} violations")
        for kernel, ln, s, hit in bad[:40]:
            print(f"  {kernel} line {ln}: `{s}` touches in-flight v{hit}")
        status |= bool(bad)
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
