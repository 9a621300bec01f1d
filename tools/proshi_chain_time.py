#!/usr/bin/env python3
"""Times the ProShI small-batch chain (proshi_chain_kernel) at d = CIAO_D (1024) fp64/fp32: random vs sequential agents, r = 1 and 8."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedSepQuad, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
ctx = Context(0)
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
d = int(os.environ.get("CIAO_D", "1024"))
for dt in (torch.float64, torch.float32):
    N = 250_000
    Q = torch.empty((N, d), dtype=dt, device="cuda"); q = torch.empty((N, d), dtype=dt, device="cuda")
    ctx.synth_normal(Q, 0, 7, 1.0); ctx.synth_normal(q, 0, 8, 1.0); Q.abs_()
    f = PackedSepQuad(Q, q, eta=30.0, lo=-2.0, hi=2.0)
    g = ProxG(L.PROX_BOX, lo=-float("inf"), hi=1.0)
    gam = torch.full((N,), 0.999 * N / 40.0, dtype=dt, device="cuda")
    x0 = torch.zeros(d, dtype=dt, device="cuda")
    table = torch.empty((N, d), dtype=dt, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    hgd = torch.empty(1, dtype=dt, device="cuda")
    ctx.proshi_init(f, g, gam, x0, table, av, z, hgd)
    hg = float(hgd.item())
    out = []
    for r in (1, 8):
        nit = 40000 // r
        for name, idx in (("random", IndexStream(1).sample_batches(N, r, nit).reshape(-1).copy()),
                          ("sequential", (np.arange(nit * r, dtype=np.int64) * 1) % N)):
            bidx = ctx._idx(idx)
            bptr = np.arange(nit + 1, dtype=np.int64) * r
            ctx.proshi_steps(f, g, gam, hg, bptr[:3], bidx[:2 * r], table, av, z); ctx.synchronize()
            t0 = time.perf_counter(); ctx.proshi_steps(f, g, gam, hg, bptr, bidx, table, av, z); ctx.synchronize()
            t = time.perf_counter() - t0
            out.append(f"r={r} {name}: {t / (nit * r) * 1e6:.3f} us/visit")
    print(("f64 " if dt == torch.float64 else "f32 ") + " | ".join(out) + f"  [{ctx.last_kernel()}]")
    del Q, q, table
