#!/usr/bin/env python3
"""Grid scan for the table modes of the rows kernel (SAGA init: read A + write table; Finito batch: read A, read+write table)."""
import json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
ctx = Context(0)
ctx.set_option("sweep_blocks_per_cu", 16)
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
    ctx.finito_init(F, g, gam, hg, x0, table, av, z)
    r = N // 2
    bidx = ctx._idx(IndexStream(0).sample_without_replacement(N, r))
    bptr = np.array([0, r], np.int64)
    ctx.set_option("chain_max_batch", 0)
    for grid in (128, 192, 256, 384, 512, 768, 1024, 2048):
        ctx.set_option("sweep_grid", grid)
        res = {}
        for name, fn, nbytes in (("saga_init", lambda: ctx.saga_init(F, g, 0.5, x0, table, av, z), 2 * N * d * es),
                                 ("finito_batch", lambda: ctx.finito_steps(F, g, gam, hg, bptr, bidx, table, av, z), 3 * r * d * es)):
            fn(); ctx.timing_enable(True); ctx.timing_read()
            for _ in range(4): fn()
            ms, n = ctx.timing_read(); ctx.timing_enable(False)
            res[name] = round(nbytes / (ms / n * 1e-3) / 1e9)
        print(json.dumps({"dtype": str(dt)[6:], "d": d, "grid": grid, **res}), flush=True)
    del A, b, table, F
    torch.cuda.empty_cache()
