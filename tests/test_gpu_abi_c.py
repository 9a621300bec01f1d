"""The C ABI without Python in the loop: tests/abi_c/abi_consumer.c (plain C, gcc, HIP runtime only for device memory)
links libciao_hip.so through include/ciao_hip.h, runs the sweep, the fused prox step and a SAGA chain, and checks them
against loops written out in C.  This is the boundary a Julia `ccall` wrapper or a C++ host binds."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "ciaoalgorithms.jl_amd", "abi_consumer")


def test_plain_c_host_drives_the_abi():
    assert os.path.exists(EXE), "abi_consumer is not built: python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ABI_C_OK" in r.stdout, r.stdout
    assert "status -1" in r.stdout   # the NULL-argument probe came back as CIAO_ERR_ARG, not as a crash
