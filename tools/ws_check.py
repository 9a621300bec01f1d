#!/usr/bin/env python3
"""chain_ws_kernel (wave-specialised chain) against chain_dma_kernel (option chain_no_ws=1) on cuda:0: results must be BITWISE
equal; prints us per update for both.  CIAO_WS_CASES=quick|all, CIAO_WS_STEPS=<m>."""
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
M = int(os.environ.get("CIAO_WS_STEPS", "20000"))
ISS = [int(v) for v in os.environ.get("CIAO_WS_ISSUERS", "2").split(",")]


def problem(dt, N, d, loss):
    A = torch.empty((N, d), dtype=dt, device="cuda")
    b = torch.empty((N,), dtype=dt, device="cuda")
    ctx.synth_normal(A, 0, 1, 1 / np.sqrt(d))
    if loss == "ls":
        F = PackedF(L.LOSS_LS, A, b, float(N))
        ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, False, 1, b)
    else:
        F = PackedF(L.LOSS_LOGISTIC, A, b, 1.0)
        ctx.synth_targets(F, torch.ones(d, dtype=dt, device="cuda"), 0.1, True, 1, b)
    return F


def run(alg, F, g, dt, N, d, m, loss, ws, iss=0):
    ctx.set_option("chain_no_ws", 0 if ws else 1)
    ctx.set_option("chain_ws_issuers", iss)
    x0 = torch.full((d,), 0.01, dtype=dt, device="cuda")
    idx = ctx._idx(IndexStream(3).rand_indices(N, m))
    gamma = 1e-7 if loss == "ls" else 0.5
    if alg in ("svrg", "svrgc"):
        av, z, zf, w = (torch.empty_like(x0) for _ in range(4))
        ctx.svrg_init(F, x0, av, z, zf, w)
        ctx.synchronize()
        t0 = time.perf_counter()
        if alg == "svrg":
            ctx.svrg_inner(F, g, gamma, idx, av, z, zf, w)
        else:
            ctx.svrg_iterate(F, g, gamma, idx, False, av, z, zf, w, reuse_rowdots=True)
        ctx.synchronize()
        t = time.perf_counter() - t0
        return t, [v.clone() for v in (av, z, zf, w)], ctx.last_kernel()
    table = torch.empty((N, d), dtype=dt, device="cuda")
    av, z = torch.empty_like(x0), torch.empty_like(x0)
    ctx.saga_init(F, g, gamma, x0, table, av, z)
    ctx.synchronize()
    t0 = time.perf_counter()
    ctx.saga_steps(F, g, gamma, alg == "sag", idx, table, av, z)
    ctx.synchronize()
    t = time.perf_counter() - t0
    return t, [av.clone(), z.clone(), table], ctx.last_kernel()


cases = []
mode = os.environ.get("CIAO_WS_CASES", "quick")
if mode == "quick":
    cases = [("svrgc", torch.float64, 50_000, 1024, "ls", "l1"), ("svrg", torch.float64, 50_000, 1024, "ls", "l1"),
             ("saga", torch.float32, 50_000, 1024, "lg", "l1"), ("saga", torch.float32, 300, 1024, "lg", "l1")]
else:
    for dt in (torch.float64, torch.float32):
        for alg in ("svrgc", "svrg", "saga", "sag"):
            for d in ((1024, 1000, 600) if dt == torch.float64 else (1024, 2048, 4096, 1000, 3000, 700)):
                for loss in ("ls", "lg"):
                    for gk in ("l1", "box", "zero"):
                        if (loss, gk) in (("lg", "box"),) and d not in (1024,):
                            continue
                        for N in (20_000, 97):
                            cases.append((alg, dt, N, d, loss, gk))
bad = 0
for alg, dt, N, d, loss, gk in cases:
    F = problem(dt, N, d, loss)
    g = {"l1": ProxG(L.PROX_L1, lam=1e-3), "box": ProxG(L.PROX_BOX, lo=-0.02, hi=0.015), "zero": ProxG(L.PROX_ZERO)}[gk]
    m = M if N > 1000 else max(M // 10, 500)
    t_old, r_old, k_old = run(alg, F, g, dt, N, d, m, loss, ws=False)
    line = f"{alg:5s} {'f64' if dt == torch.float64 else 'f32'} N={N} d={d} {loss} {gk}: old {t_old / m * 1e6:.3f} us [{k_old.split('<')[0]}]"
    for iss in ISS:
        t_new, r_new, k_new = run(alg, F, g, dt, N, d, m, loss, ws=True, iss=iss)
        same = all(torch.equal(a, b) for a, b in zip(r_old, r_new))
        maxd = max(float((a - b).abs().max()) for a, b in zip(r_old, r_new))
        bad += 0 if same else 1
        line += f" | ws{iss} {t_new / m * 1e6:.3f} us {'BITWISE' if same else 'DIFF %.3e' % maxd} [{k_new.split('>')[0].split('<')[0]}]"
    print(line, flush=True)
    del F
    torch.cuda.empty_cache()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
