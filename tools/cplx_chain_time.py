#!/usr/bin/env python3
"""us per update of the complex chains (512 complex fp64 entries per row = 8 KiB, and 512 complex fp32): SVRG inner cycle and SAGA
steps; CIAO_OPTS=chain_no_dma=1 times the register-resident kernel (one row in flight) instead of the LDS-DMA ring."""
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
for kv in os.environ.get("CIAO_OPTS", "").split(","):
    if "=" in kv:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
out = []
for dt in (torch.float64, torch.float32):
    N, n, m = 200_000, 512, 100_000
    A = torch.empty((N, 2 * n), dtype=dt, device="cuda"); b = torch.empty((2 * N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, seed=11, scale=1.0 / np.sqrt(2 * n))
    ctx.synth_normal(b.view(N, 2), 0, seed=12, scale=1.0)
    F = PackedF.least_squares_complex(A, b, float(N))
    g = ProxG(L.PROX_L1_COMPLEX, lam=1e-3)
    x0 = torch.zeros(2 * n, dtype=dt, device="cuda")
    av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
    ctx.svrg_init(F, x0, av, z, zf, w)
    idx = ctx._idx(IndexStream(0).rand_indices(N, m))
    ctx.svrg_inner(F, g, 1e-7, idx[:2000], av, z, zf, w); ctx.synchronize()
    t0 = time.perf_counter(); ctx.svrg_inner(F, g, 1e-7, idx, av, z, zf, w); ctx.synchronize()
    ts = (time.perf_counter() - t0) / m * 1e6
    ks = ctx.last_kernel().split(" ")[0]
    table = torch.empty((N, 2 * n), dtype=dt, device="cuda")
    sav, sz = torch.empty_like(x0), torch.empty_like(x0)
    ctx.saga_init(F, g, 1e-7, x0, table, sav, sz)
    ctx.saga_steps(F, g, 1e-7, False, idx[:2000], table, sav, sz); ctx.synchronize()
    t0 = time.perf_counter(); ctx.saga_steps(F, g, 1e-7, False, idx, table, sav, sz); ctx.synchronize()
    tg = (time.perf_counter() - t0) / m * 1e6
    out.append(f"{'f64' if dt == torch.float64 else 'f32'} n=512: SVRG {ts:.3f} us [{ks}]  SAGA {tg:.3f} us [{ctx.last_kernel().split(' ')[0]}]")
    del A, table
print(" | ".join(out))
