"""Checks on the compiled ISA of kernels whose correctness depends on something the compiler does not know."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_oproj_pending_residual_registers_untouched():
    """oproj_ln_kernel loads its residual fragments by inline asm two stages before its own s_waitcnt: no instruction in
    between may touch the destination registers, and neither hand-counted kernel may spill (tools/check_pending_loads.py)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_pending_loads.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
