#!/usr/bin/env python3
"""Wave-per-row vs workgroup-per-row for the table modes and the plain sweep at scale (split_all experiment)."""
import json, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
torch.cuda.set_device(0)
ctx = Context(0)
for dt, d, N in ((torch.float32, 1024, 4_000_000), (torch.float64, 1024, 2_000_000), (torch.float32, 4096, 1_000_000)):
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
    for split, bpc in ((0, 0), (1, 0), (1, 2), (1, 3), (1, 4), (1, 6)):
        ctx.set_option("split_all", split)
        ctx.set_option("split_blocks_per_cu", bpc)
        res = {}
        for name, fn, nbytes in (("grad", lambda: ctx.full_gradient(F, x0, av), N * d * es),
                                 ("saga_init", lambda: ctx.saga_init(F, g, 0.5, x0, table, av, z), 2 * N * d * es),
                                 ("finito_init", lambda: ctx.finito_init(F, g, gam, hg, x0, table, av, z), 2 * N * d * es)):
            fn(); ctx.timing_enable(True); ctx.timing_read()
            for _ in range(4): fn()
            ms, n = ctx.timing_read(); ctx.timing_enable(False)
            res[name] = round(nbytes / (ms / n * 1e-3) / 1e9)
        print(json.dumps({"dtype": str(dt)[6:], "d": d, "split": split, "bpc": bpc, **res, "kernel": ctx.last_kernel()}), flush=True)
    ctx.set_option("split_all", 0); ctx.set_option("split_blocks_per_cu", 0)
    del A, b, table, F
    torch.cuda.empty_cache()
