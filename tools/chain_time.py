#!/usr/bin/env python3
"""Times the SVRG inner chain (d = 1024 f64 and f32) on cuda:0; prints us per update."""
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
    ctx.svrg_inner(F, g, 1e-7, idx[:2000], av, z, zf, w); ctx.synchronize()
    t0 = time.perf_counter(); ctx.svrg_inner(F, g, 1e-7, idx, av, z, zf, w); ctx.synchronize()
    out.append(f"{'f64' if dt == torch.float64 else 'f32'} {(time.perf_counter() - t0) / m * 1e6:.3f} us/update")
print(" | ".join(out))
