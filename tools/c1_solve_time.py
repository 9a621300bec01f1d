#!/usr/bin/env python3
"""BASELINE config #1 (N = 1000, d = 50, Float64 lasso built from N one-row operators, as test/test_lasso.jl does) through the public
solver API: wall time per iteration of each solver, host work (packing, index draws, launches) included -- and the single-threaded C
oracle's time for the same iterations beside it."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ciao_loader
ciao_loader.load()
from ciaoalgorithms_jl_amd import solvers as S, operators as ops
from ciaoalgorithms_jl_amd.sampling import IndexStream
from oracle import oracle as O
import problems as P
torch.cuda.set_device(0)
N, d = int(os.environ.get("CIAO_N", "1000")), int(os.environ.get("CIAO_D", "50"))
T = np.float32 if os.environ.get("CIAO_F32") else np.float64
A, b, x0 = P.synthetic("ls", N, d, T, seed=3)
x0 = np.zeros(d, T)
F = [ops.LeastSquares(A[i:i + 1, :], b[i:i + 1], float(N)) for i in range(N)]
g = ops.NormL1(0.01)
L = float(N) * np.sum(A.astype(np.float64) ** 2, axis=1)
op, og = O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=0.01)
gamma = 1.0 / (10 * L.max())
cases = [("SVRG (m = N)", lambda mi: S.SVRG(T, γ=gamma, maxit=mi), 200, N),
         ("SAGA", lambda mi: S.SAGA(T, γ=1 / (3 * L.max()), maxit=mi), 200_000, 1),
         ("Finito r=1 sweeping=1", lambda mi: S.Finito(T, maxit=mi), 200_000, 1),
         ("Finito r=10 sweeping=2", lambda mi: S.Finito(T, sweeping=2, minibatch=(True, 10), maxit=mi), 50_000, 10),
         ("LFinito r=1 sweeping=2", lambda mi: S.Finito(T, LFinito=True, sweeping=2, maxit=mi), 200, 2 * N),
         ("adaptive Finito", lambda mi: S.Finito(T, adaptive=True, maxit=mi), 100_000, 1)]
print(f"N={N} d={d} {np.dtype(T).name}")
for name, mk, maxit, updates in cases:
    kw = dict(F=F, g=g, N=N, L=L, stream=IndexStream(1))
    mk(3)(x0, **kw)                                    # warm-up (packing paths, kernels)
    t0 = time.perf_counter(); x, it = mk(maxit)(x0, **dict(kw, stream=IndexStream(1))); t = time.perf_counter() - t0
    print(f"{name:26s} {it:7d} iterations in {t * 1e3:8.1f} ms = {t / it * 1e6:8.2f} us per iteration, {it * updates / t / 1e6:6.2f} M updates/s (API, packing included)", flush=True)
# the C oracle on one host core: the same kinds of update, steps in bulk (no Python between steps)
st = IndexStream(1)
rav, rz, rzf, rw = O.svrg_init(op, x0)
idx = st.rand_indices(N, 200 * N)
t0 = time.perf_counter(); O.svrg_inner(op, og, T(gamma), idx, rav, rz, rzf, rw); t = time.perf_counter() - t0
print(f"oracle svrg_inner: {len(idx) / t / 1e6:.2f} M updates/s on one core")
rt, rav, rz = O.saga_init(op, og, T(gamma), x0)
t0 = time.perf_counter(); O.saga_steps(op, og, T(gamma), False, idx, rt, rav, rz); t = time.perf_counter() - t0
print(f"oracle saga_steps: {len(idx) / t / 1e6:.2f} M updates/s on one core")
