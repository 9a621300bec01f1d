"""The record "every kernel of the library is launched by some GPU test"  is kept HONEST on the CPU: the kernels of the
BUILT library must be exactly the ones the committed trace of the GPU suite launched (profiles/r05_launched_kernels.tsv, made by
tools/exp/reached_kernels.sh) -- no exceptions.  A new instantiation that no test launches -- or a removed one still in the
record -- fails here until the trace is re-run on a GPU box and the record committed."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
LIB = os.path.join(ROOT, "ciaoalgorithms.jl_amd", "libciao_hip.so")
RECORD = os.path.join(ROOT, "profiles", "r05_launched_kernels.tsv")
UNTRACED = ()   # (none: the child ranks of tests/test_gpu_multirank.py are traced too -- rocprofv3 -o "t_%pid%")


@pytest.mark.skipif(not os.path.exists(LIB), reason="library not built")
def test_every_kernel_of_the_library_is_in_the_launch_record():
    import kernel_meta
    import reached_kernels
    held = {reached_kernels.norm(k["name"]) for k in kernel_meta.library_kernels(LIB)}
    launched = set()
    for line in open(RECORD):
        cnt, name = line.rstrip("\n").split("\t", 1)
        assert int(cnt) > 0
        launched.add(reached_kernels.norm(name))
    never = sorted(n for n in held - launched if not (UNTRACED and n.startswith(UNTRACED)))
    stale = sorted(launched - held)
    assert not never, ("kernels of the built library that the recorded GPU-suite trace never launched (add a test, or prune; then "
                       "re-run tools/exp/reached_kernels.sh and commit profiles/r05_launched_kernels.tsv):\n  " + "\n  ".join(never))
    assert not stale, "the launch record names kernels the library no longer holds (re-run the trace):\n  " + "\n  ".join(stale)
    assert len(held) > 1000
