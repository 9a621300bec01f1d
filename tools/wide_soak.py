#!/usr/bin/env python3
"""Soak of chain_wide_kernel (several workgroups share one chain): long SAGA / SVRG chains run twice -- bitwise equal (the mailbox protocol
has no timing-dependent result) -- and SAGA's invariant av == mean of the table rows afterwards."""
import os, sys, time
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
m = int(os.environ.get("CIAO_M", "500000"))
for d, dt in ((9000, torch.float32), (20000, torch.float64), (131072, torch.float32)):
    N = 1500
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LOGISTIC, A, b, 1.0)
    ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, True, 1, b)
    g = ProxG(L.PROX_L1, lam=1.0 / N)
    x0 = torch.ones(d, dtype=dt, device="cuda")
    idx = ctx._idx(IndexStream(d).rand_indices(N, m))
    outs = []
    for rep in range(2):
        table = torch.empty((N, d), dtype=dt, device="cuda")
        av, z = torch.empty_like(x0), torch.empty_like(x0)
        ctx.saga_init(F, g, 0.2, x0, table, av, z)
        t0 = time.perf_counter(); ctx.saga_steps(F, g, 0.2, False, idx, table, av, z); ctx.synchronize(); t = time.perf_counter() - t0
        outs.append((z.clone(), av.clone(), table))
    same = torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])
    inv = float((outs[0][1].double() - outs[0][2].double().mean(dim=0)).abs().max() / outs[0][1].double().abs().max())
    print(f"SAGA d={d} {str(dt)[6:]} m={m}: {t / m * 1e6:.2f} us/update, two runs bitwise equal: {same}, |av - mean(table)| / |av| = {inv:.1e}  [{ctx.last_kernel().split(' block')[0]}]", flush=True)
    assert same and np.isfinite(inv) and inv < (1e-3 if dt == torch.float32 else 1e-10)
    ws = []
    for rep in range(2):
        av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
        ctx.svrg_init(F, x0, av, z, zf, w)
        ctx.svrg_inner(F, g, 0.05, idx, av, z, zf, w); ctx.synchronize()
        ws.append((w.clone(), z.clone()))
    same = torch.equal(ws[0][0], ws[1][0]) and torch.equal(ws[0][1], ws[1][1])
    print(f"SVRG d={d} {str(dt)[6:]} m={m}: two runs bitwise equal: {same}, finite: {bool(torch.isfinite(ws[0][0]).all())}", flush=True)
    assert same
    del A, table, outs
    torch.cuda.empty_cache()
print("ok")
