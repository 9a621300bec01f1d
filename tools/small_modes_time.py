#!/usr/bin/env python3
"""Sweep / SAGA init / Finito init bandwidth on short rows (rows_small_kernel), algorithmic GB/s."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
torch.cuda.set_device(0)
ctx = Context(0)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
for dt, d in ((torch.float64, 50), (torch.float64, 120), (torch.float32, 50), (torch.float32, 100), (torch.float32, 200), (torch.float32, 255)):
    N = 4_000_000
    es = 8 if dt == torch.float64 else 4
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(N))
    ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=dt, device="cuda")
    table = torch.empty((N, d), dtype=dt, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    gam = torch.full((N,), 0.7, dtype=dt, device="cuda")
    hg = ctx.hat_gamma(gam)
    res = {}
    for name, fn, nbytes in (("grad", lambda: ctx.full_gradient(F, x0, av), N * d * es),
                             ("saga_init", lambda: ctx.saga_init(F, g, 0.5, x0, table, av, z), 2 * N * d * es),
                             ("finito_init", lambda: ctx.finito_init(F, g, gam, hg, x0, table, av, z), 2 * N * d * es)):
        fn(); ctx.timing_enable(True); ctx.timing_read()
        for _ in range(4): fn()
        ms, n = ctx.timing_read(); ctx.timing_enable(False)
        res[name] = round(nbytes / (ms / n * 1e-3) / 1e9)
    print(str(dt)[6:], d, res, ctx.last_kernel(), flush=True)
    del A, table
    torch.cuda.empty_cache()
