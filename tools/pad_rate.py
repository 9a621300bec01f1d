#!/usr/bin/env python3
"""What feature padding (solvers.PAD_FEATURES) buys through the solver API on rows that are not whole 16-byte chunks: seconds per SVRG
epoch (m = N) and per 10^5 SAGA iterations, problem packed from N host operator objects as the reference's tests build it."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ciao_loader
ciao_loader.load()
import ciaoalgorithms_jl_amd.solvers as S
import ciaoalgorithms_jl_amd.operators as ops
from ciaoalgorithms_jl_amd.device import Context
from ciaoalgorithms_jl_amd.sampling import IndexStream
ctx = Context(0)
N = int(os.environ.get("CIAO_N", "40000"))
for T, d in ((np.float64, 51), (np.float64, 785), (np.float64, 2049), (np.float32, 50), (np.float32, 1001), (np.float32, 4095)):
    rng = np.random.default_rng(d)
    A = (rng.standard_normal((N, d)) / np.sqrt(d)).astype(T)
    b = (A @ rng.standard_normal(d).astype(T) * 0.1).astype(T)
    F = [ops.LeastSquares(A[i:i + 1], b[i:i + 1], float(N)) for i in range(N)]
    g = ops.NormL1(1e-3)
    x0 = np.zeros(d, dtype=T)
    row = [f"d={d:5d} {'f64' if T == np.float64 else 'f32'} N={N}:"]
    for pad in (True, False):
        S.PAD_FEATURES = pad
        it = iter(S.iterator(S.SVRG(T, γ=1.0 / (7 * 1.3 * N)), x0, F=F, g=g, N=N, ctx=ctx, stream=IndexStream(1)))
        next(it); next(it); ctx.synchronize()
        t0 = time.perf_counter(); next(it); next(it); ctx.synchronize(); ts = (time.perf_counter() - t0) / 2
        ks = ctx.last_kernel().split("<")[0]
        it = iter(S.iterator(S.SAGA(T, γ=1.0 / (3 * 1.3 * N)), x0, F=F, g=g, N=N, ctx=ctx, stream=IndexStream(1)))
        for _ in range(2000): next(it)
        ctx.synchronize()
        t0 = time.perf_counter()
        for _ in range(100000 // 1): 
            st = next(it)
        ctx.synchronize(); tg = time.perf_counter() - t0
        row.append(f"{'padded  ' if pad else 'unpadded'} svrg epoch {ts * 1e3:7.2f} ms = {ts / N * 1e6:5.2f} us/update  saga {tg / 1e5 * 1e6:5.2f} us/iteration (host loop included)")
    S.PAD_FEATURES = True
    print(" | ".join(row), flush=True)
