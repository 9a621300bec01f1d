#!/usr/bin/env python3
"""Where a chain step's cycles go: runs the SVRG inner cycle of a CIAO_CHAIN_DBG=8 build (s_memtime stamps at five points
of every step, summed per wave) and prints the average core cycles per segment.  Use through tools/chain_stamps.sh."""
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
SEG = ["barrier -> 4 partials read+added", "link fn + update + prox + DMA issue", "next inputs from LDS + dot + in-wave reduce",
       "write partial, LDS idle", "wait in barrier"]
for dt in (torch.float64, torch.float32):
    N, d, m = 200_000, int(os.environ.get("CIAO_D", "1024")), 100_000
    A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    F = PackedF(L.LOSS_LS, A, b, float(N))
    ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=dt, device="cuda")
    av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
    ctx.svrg_init(F, x0, av, z, zf, w)
    idx = ctx._idx(IndexStream(0).rand_indices(N, m))
    dbg = torch.zeros(24, dtype=torch.int64, device="cuda")
    ctx.set_option("chain_dbg_ptr", dbg.data_ptr())
    ctx.svrg_inner(F, g, 1e-7, idx[:2000], av, z, zf, w); ctx.synchronize()
    t0 = time.perf_counter(); ctx.svrg_inner(F, g, 1e-7, idx, av, z, zf, w); ctx.synchronize()
    us = (time.perf_counter() - t0) / m * 1e6
    v = dbg.cpu().numpy().reshape(4, 6)
    print(f"{'f64' if dt == torch.float64 else 'f32'} d={d}: {us:.3f} us/update with stamps; {ctx.last_kernel()}")
    for k in range(5):
        per = v[:, k] / np.maximum(v[:, 5], 1)
        print(f"   {SEG[k]:46s} cycles/step per wave: {np.round(per, 1).tolist()}")
    tot = (v[:, :5].sum(axis=1) / np.maximum(v[:, 5], 1))
    print(f"   total {np.round(tot, 1).tolist()} cycles/step -> {us * 1e3 / tot.mean():.2f} ns per cycle")
