#!/usr/bin/env python3
"""Would a Finito batch run faster if its rows were already in the memory-side cache?  The same batch repeated (its 48 KB per row
re-read from wherever the previous visit left them) against distinct batches, d = 4096 fp32; us per batch."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG
from ciaoalgorithms_jl_amd.sampling import IndexStream
torch.cuda.set_device(0)
ctx = Context(0)
N, d, dt = 250_000, 4096, torch.float32
A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
F = PackedF(L.LOSS_LS, A, b, float(N))
ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
g = ProxG(L.PROX_L1, lam=1e-3)
gam = torch.full((N,), 0.999 / 1.3, dtype=dt, device="cuda")
hg = ctx.hat_gamma(gam)
x0 = torch.zeros(d, dtype=dt, device="cuda")
table = torch.empty((N, d), dtype=dt, device="cuda")
av, z = torch.empty_like(x0), torch.empty_like(x0)
ctx.finito_init(F, g, gam, hg, x0, table, av, z)
st = IndexStream(0)
for r in (256, 512, 1024, 4096):
    nit = 400
    one = st.sample_without_replacement(N, r)
    for name, batches in (("distinct", [st.sample_without_replacement(N, r) for _ in range(nit)]), ("repeated", [one] * nit)):
        bidx = ctx._idx(np.concatenate(batches))
        bptr = np.arange(nit + 1, dtype=np.int64) * r
        ctx.finito_steps(F, g, gam, hg, bptr[:3], bidx[:2 * r], table, av, z); ctx.synchronize()
        t0 = time.perf_counter(); ctx.finito_steps(F, g, gam, hg, bptr, bidx, table, av, z); ctx.synchronize()
        t = time.perf_counter() - t0
        print(f"r={r} {name}: {t / nit * 1e6:.2f} us/batch", flush=True)
