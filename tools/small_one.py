#!/usr/bin/env python3
"""One GRAD sweep shape on short rows (CIAO_D, CIAO_F32, CIAO_N; CIAO_OPTS k=v,...): GB/s by HIP events; for counter passes."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF
torch.cuda.set_device(0)
ctx = Context(0)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
d = int(os.environ.get("CIAO_D", "50")); N = int(os.environ.get("CIAO_N", "4000000"))
dt = torch.float32 if os.environ.get("CIAO_F32", "1") != "0" else torch.float64
es = 4 if dt == torch.float32 else 8
A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
F = PackedF(L.LOSS_LS, A, b, float(N))
ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
x0 = torch.zeros(d, dtype=dt, device="cuda"); av = torch.empty_like(x0)
ctx.full_gradient(F, x0, av); ctx.timing_enable(True); ctx.timing_read()
for _ in range(int(os.environ.get("CIAO_REPS", "4"))): ctx.full_gradient(F, x0, av)
ms, n = ctx.timing_read()
print(str(dt)[6:], d, round(N * d * es / (ms / n * 1e-3) / 1e9), "GB/s", ctx.last_kernel(), flush=True)
