"""CPU: the hand-managed register double buffer of the W-direct row-owning GEMM survives code generation.

rowgemm_wd_kernel (jyutvoice_amd/csrc/rowgemm_kernel.h) loads its weight fragments by inline asm and waits for them with a
counted s_waitcnt the compiler knows nothing about; tools/check_rowgemm_isa.py compiles the kernels for gfx950 and asserts
that no instruction but those loads and the MFMAs touches the buffer's registers once a load into them has been issued
(a register-allocator copy there would read a register whose load is still in flight)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import REPO


def test_register_double_buffer_is_untouched_between_load_and_use():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang in this environment")
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "check_rowgemm_isa.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 violations" in r.stdout
