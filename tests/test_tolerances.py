"""The tolerances the GPU parity tests STATE (every `scale=` / `scale64=` of a close() call, in units of eps(R) of the result) stay inside
BASELINE.md section 3: 1e-10 relative for Float64, 1e-4 for Float32.  (CPU test: reads the test sources.)

Float32 is two-sided (tests/test_gpu_parity.py, above close()): `scale64=` bounds the device's Float32 result against the oracle run in
FLOAT64 on the same Float32 data -- the accuracy statement, at most 840 eps32 = 1e-4; `scale={64: a, 32: b}`'s b bounds it against the
Float32 oracle, whose own sequential folds are often the less accurate side (documented where b exceeds 840)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
FILES = ("test_gpu_parity.py", "test_gpu_complex.py", "test_gpu_wide_chain.py", "test_gpu_long_rows.py", "test_gpu_small_mfma.py",
         "test_gpu_feature_padding.py", "test_gpu_multi_rhs.py", "test_gpu_every_kernel.py")
F64_MAX = 1e-10 / np.finfo(np.float64).eps      # 450 359 eps64
F32_MAX = 1e-4 / np.finfo(np.float32).eps       # 838.9 eps32


def _sites():
    for f in FILES:
        for n, line in enumerate(open(os.path.join(ROOT, f)).read().split("\n"), 1):
            if "close(" not in line or line.lstrip().startswith(("def ", "#")):
                continue
            yield f, n, line


def test_stated_tolerances_stay_inside_the_baseline():
    n64 = n32 = nacc = 0
    over32 = []
    for f, n, line in _sites():
        m = re.search(r"scale64=([0-9.e+]+)", line)
        if m:
            nacc += 1
            assert float(m.group(1)) <= F32_MAX + 1.2, f"{f}:{n}: Float32 against the Float64 oracle allowed {m.group(1)} eps32 > 1e-4"
        m = re.search(r"scale=\{([^}]*)\}", line)
        if m:
            d = {int(k): float(v) for k, v in (kv.split(":") for kv in m.group(1).split(","))}
            a, b = d.get(64), d.get(32)
        else:
            m = re.search(r"scale=([0-9.e+]+)", line)
            if not m:
                continue
            a = b = float(m.group(1))
        if a is not None:
            n64 += 1
            assert a <= F64_MAX, f"{f}:{n}: Float64 allowed {a} eps64 > 1e-10"
        if b is not None:
            n32 += 1
            if b > F32_MAX + 1.2:
                over32.append((f, n, b))
    assert n64 > 250 and nacc >= 15, (n64, nacc)
    # Float32 against the FLOAT32 oracle beyond 1e-4: only where the same site also carries the Float64-oracle form (`scale64=`, written
    # by tools/retune_scales.py where the twinned oracle gave a Float64 value), or the site says why (a comment with "eps32" or
    # "Float32 oracle" on the line or the three above)
    for f, n, b in over32:
        lines = open(os.path.join(ROOT, f)).read().split("\n")
        ctx = " ".join(lines[max(0, n - 4):n])
        here = lines[n - 1]
        assert "scale64=" in here or "ref64=" in here or "eps32" in ctx or "Float32 oracle" in ctx, \
            f"{f}:{n}: Float32 allowed {b} eps32 > 1e-4 with neither the Float64-oracle form (scale64=) nor a stated reason"
