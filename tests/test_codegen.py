"""Generated-code checks (hipcc cross-compiles here): properties of the gfx950 assembly that no numerical test on a lucky schedule would catch."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
@pytest.mark.parametrize("unit", ["nn_half.hip", "tdfnet.hip"])
def test_asynchronous_register_loads_are_left_alone(unit):
    """the kernels that request operands with inline-asm loads (alsep_gfx950_asm.h: invisible to hipcc's waitcnt bookkeeping) wait for them
    with their own counted s_waitcnt; hipcc, which believes the destination registers defined at the asm, must not have copied, spilled
    or reused them in between -- it did once (nn_gemm_h2.h, epilogue operands requested in one conditional and used in another: wrong
    residuals, then wild addresses, on one launch in six).  scripts/check_async_regs.py walks the assembly of every such kernel."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_async_regs.py"), os.path.join(ROOT, "audiolab_amd", "csrc", unit)],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "asynchronous register loads, 0 violations" in r.stdout
    assert " 0 asynchronous register loads" not in r.stdout          # the unit does use them: the checker found what it is there for
