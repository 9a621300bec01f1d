#!/usr/bin/env python3
"""End-to-end iteration rates through the public solver API (functor call, host batch selection included)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L
from ciaoalgorithms_jl_amd.device import PackedF, default_context
from ciaoalgorithms_jl_amd.operators import NormL1
from ciaoalgorithms_jl_amd.solvers import SAGA, Finito, SVRG
torch.cuda.set_device(0)
ctx = default_context()
N, d, dt = 500_000, 1024, torch.float32
A = torch.empty((N, d), dtype=dt, device="cuda"); b = torch.empty((N,), dtype=dt, device="cuda")
ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
F = PackedF(L.LOSS_LS, A, b, float(N))
ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda") * 0.1, 0.1, False, 1, b)
x0 = np.zeros(d, np.float32)
g = NormL1(1e-3)
Lc = float(N) * 1.3
cases = [("SAGA", SAGA(np.float32, γ=1 / (3 * Lc), maxit=400_000), {}),
         ("Finito r=1 sweeping=1", Finito(np.float32, maxit=400_000), {"L": Lc}),
         ("Finito r=1 sweeping=3", Finito(np.float32, sweeping=3, maxit=400_000), {"L": Lc}),
         ("Finito r=256 sweeping=1", Finito(np.float32, minibatch=(True, 256), maxit=20_000), {"L": Lc}),
         ("Finito r=4096 sweeping=1", Finito(np.float32, minibatch=(True, 4096), maxit=2_000), {"L": Lc}),
         ("Finito r=256 sweeping=2", Finito(np.float32, sweeping=2, minibatch=(True, 256), maxit=20_000), {"L": Lc}),
         ("Finito adaptive", Finito(np.float32, adaptive=True, maxit=200_000), {}),
         ("LFinito r=256", Finito(np.float32, LFinito=True, sweeping=2, minibatch=(True, 256), maxit=4), {"L": Lc})]
for name, solver, kw in cases:
    solver(x0, F=F, g=g, N=N, **kw)      # warm: the first call of a shape pays allocations and the kernels' first load
    ctx.synchronize()
    t0 = time.perf_counter()
    x, it = solver(x0, F=F, g=g, N=N, **kw)
    ctx.synchronize()
    t = time.perf_counter() - t0
    print(f"{name:28s} {it:8d} iterations in {t:7.3f} s  = {t / it * 1e6:9.2f} us per iteration", flush=True)
