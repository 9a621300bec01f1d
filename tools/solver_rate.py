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

# ---- BASELINE config #1 (N = 1000, d = 50, fp64 Lasso; SVRG, 30 epochs of m = N updates) through the functor: a 400 KB problem, one
# workgroup's worth of work -- the GPU path against one host core running the oracle (tests/test_gpu_configs.py compares the results)
del A, b, F
rng = np.random.default_rng(0)
N, d = 1000, 50
A1 = rng.standard_normal((N, d)) / np.sqrt(d)
xt = rng.standard_normal(d) * (rng.random(d) < 0.2)
b1 = A1 @ xt + 0.01 * rng.standard_normal(N)
from ciaoalgorithms_jl_amd.operators import LeastSquares, pack_F
F1 = [LeastSquares(A1[i:i + 1], b1[i:i + 1], float(N)) for i in range(N)]          # test_lasso.jl:52-54
L1 = N * (A1 ** 2).sum(1)
sol = SVRG(np.float64, γ=1 / (7 * L1.max()), maxit=31)
x0 = np.zeros(d)
sol(x0, F=F1, g=NormL1(0.01), N=N); ctx.synchronize()
t0 = time.perf_counter(); x, it = sol(x0, F=F1, g=NormL1(0.01), N=N); ctx.synchronize(); t = time.perf_counter() - t0
print(f"config #1, SVRG 30 epochs, F as {N} operator objects (packed inside the call): {t * 1e3:.2f} ms")
Fp = pack_F(F1, N, d, torch.float64, torch.device("cuda", 0))
t0 = time.perf_counter(); x, it = sol(x0, F=Fp, g=NormL1(0.01), N=N); ctx.synchronize(); t = time.perf_counter() - t0
print(f"config #1, SVRG 30 epochs, F packed beforehand: {t * 1e3:.2f} ms = {t / 30 * 1e6:.0f} us per epoch of {N} updates + a sweep")
# (the oracle's time for the same 30 epochs on one host core is printed by tests/test_gpu_configs.py::test_C1...: tools never import oracle/)
