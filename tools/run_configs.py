#!/usr/bin/env python3
"""Runs the five BASELINE.json configs (their single-GPU share) through the PUBLIC solver API on cuda:0 and prints one
JSON object: timings, throughput in the BASELINE metric's units, and -- where the CPU oracle can follow in seconds --
the iterate error against it.

  C1  Lasso SVRG N=1000 d=50 fp64            (the scaled-up test_lasso.jl generator; full parity vs the oracle)
  C2  Lasso SVRG N=1M d=1024 fp64            (1 epoch = m=N inner updates + full-gradient sweep)
  C3  l1-logistic SAGA N=10M d=1024 fp32     (table 40.96 GB in HBM)
  C4  Lasso SVRG N=80M over 8 GPUs           -> this rank's 10M-row shard: the sweep only (bench.py measures it)
  C5  Finito N=10M d=4096 fp32 over 8 GPUs   -> this rank's 1.25M-row shard, batch-parallel batches of 4096 rows
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ciao_loader  # noqa: E402

ciao_loader.load()
from ciaoalgorithms_jl_amd import _lib as L  # noqa: E402
from ciaoalgorithms_jl_amd.device import Context, PackedF, ProxG  # noqa: E402
from ciaoalgorithms_jl_amd.sampling import IndexStream  # noqa: E402
from ciaoalgorithms_jl_amd.solvers import SAGA, SVRG, Finito  # noqa: E402


def synth(ctx, N, d, tdt, logistic, seed):
    dev = torch.device("cuda", 0)
    A = torch.empty((N, d), dtype=tdt, device=dev)
    b = torch.empty((N,), dtype=tdt, device=dev)
    ctx.synth_normal(A, 0, seed=seed, scale=1.0 / np.sqrt(d))
    rng = np.random.default_rng(seed)
    xt = torch.from_numpy(rng.standard_normal(d) * (rng.random(d) < 0.05)).to(dev, tdt)
    F = PackedF(L.LOSS_LOGISTIC if logistic else L.LOSS_LS, A, b, 1.0 if logistic else float(N))
    ctx.synth_targets(F, xt, 0.1 if logistic else 0.01, logistic, seed, b)
    ctx.synchronize()
    return F


def main():
    torch.cuda.set_device(0)
    ctx = Context(0)
    out = {}

    # ---- C1: full parity against the oracle on the reference's generator at N=1000, d=50 --------------------------------
    import problems as P
    from oracle import oracle as O
    from oracle import ref_solvers as RS
    A, b, Lc, lam, x0, x_star, f_star = P.lasso_known_answer(N=1000, n=50, p=5, seed=0)
    N = A.shape[0]
    gamma = 1.0 / (7 * Lc.max())
    F = PackedF.least_squares(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda(), float(N))
    g = ProxG(L.PROX_L1, lam=lam)
    t0 = time.perf_counter()
    x, it = SVRG(np.float64, γ=gamma, maxit=30)(x0, F=F, g=g, N=N, ctx=ctx, stream=IndexStream(0))
    t_gpu = time.perf_counter() - t0
    t0 = time.perf_counter()
    xr, _ = RS.svrg(O.Problem("ls", A, b, float(N)), O.Prox("l1", lam=lam), x0, maxit=30, gamma=gamma, stream=IndexStream(0))
    t_cpu = time.perf_counter() - t0
    out["C1_lasso_svrg_N1000_d50_f64"] = {"epochs": 29, "max_abs_err_vs_oracle": float(np.abs(x - xr).max()),
                                          "rel_err": float(np.abs(x - xr).max() / np.abs(xr).max()),
                                          "cost_gap": P.lasso_cost(A, b, lam, x) - f_star, "gpu_s": t_gpu, "oracle_s": t_cpu}

    # ---- C2: Lasso SVRG N=1M d=1024 fp64, 3 epochs with m = N -----------------------------------------------------------
    N, d = 1_000_000, 1024
    F = synth(ctx, N, d, torch.float64, False, 2)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=torch.float64, device="cuda")
    solver = SVRG(np.float64, γ=1.0 / (7 * 1.3 * N), maxit=4)
    obj0 = ctx.objective(F, g, x0)
    t0 = time.perf_counter()
    x, it = solver(x0, F=F, g=g, N=N, ctx=ctx, stream=IndexStream(0))
    dt = time.perf_counter() - t0
    out["C2_lasso_svrg_N1M_d1024_f64"] = {"epochs": it - 1, "seconds": dt, "epochs_per_s": (it - 1) / dt,
                                          "inner_updates_per_s": (it - 1) * N / dt, "objective_before": obj0,
                                          "objective_after": ctx.objective(F, g, x)}
    del F
    torch.cuda.empty_cache()

    # ---- C3: l1-logistic SAGA N=10M d=1024 fp32, table in HBM ------------------------------------------------------------
    N, d = 10_000_000, 1024
    F = synth(ctx, N, d, torch.float32, True, 3)
    g = ProxG(L.PROX_L1, lam=1.0 / N)
    x0 = torch.ones(d, dtype=torch.float32, device="cuda")
    steps = 2_000_000
    solver = SAGA(np.float32, γ=1.0 / (3 * 0.25 * 1.3), maxit=steps + 1)
    obj0 = ctx.objective(F, g, x0)
    t0 = time.perf_counter()
    x, it = solver(x0, F=F, g=g, N=N, ctx=ctx, stream=IndexStream(0))
    dt = time.perf_counter() - t0
    out["C3_l1logistic_saga_N10M_d1024_f32"] = {"updates": it - 1, "seconds_incl_table_init": dt, "updates_per_s": (it - 1) / dt,
                                                "table_GB": N * d * 4 / 1e9, "objective_before": obj0,
                                                "objective_after": ctx.objective(F, g, x),
                                                "hbm_allocated_GB": torch.cuda.max_memory_allocated() / 1e9}
    del F, solver, x
    torch.cuda.empty_cache()

    # ---- C5 (this rank's share): Finito N=1.25M d=4096 fp32, batches of 4096 rows ------------------------------------------
    N, d = 1_250_000, 4096
    F = synth(ctx, N, d, torch.float32, False, 5)
    g = ProxG(L.PROX_L1, lam=1e-3)
    x0 = torch.zeros(d, dtype=torch.float32, device="cuda")
    its = 200
    solver = Finito(np.float32, maxit=its + 1, sweeping=2, minibatch=(True, 4096))
    obj0 = ctx.objective(F, g, x0)
    t0 = time.perf_counter()
    x, it = solver(x0, F=F, g=g, L=float(1.3 * N), N=N, ctx=ctx, stream=IndexStream(0))
    dt = time.perf_counter() - t0
    out["C5share_finito_N1.25M_d4096_f32_batch4096"] = {"iterations": it - 1, "seconds_incl_table_init": dt,
                                                        "samples_per_s": (it - 1) * 4096 / dt, "objective_before": obj0,
                                                        "objective_after": ctx.objective(F, g, x)}
    del F
    torch.cuda.empty_cache()
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
